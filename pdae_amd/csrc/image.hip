// Input pipeline on the device: decoded uint8 images -> network-ready batch.
//
// Reference ops replaced (ckczzj/PDAE, executed per image on DataLoader worker processes by PIL / torchvision):
//   [CropCelebA64 (F.crop 57,25,128,128)]   dataset/celeba64.py:11-14
//   transforms.Resize((S,S))                dataset/ffhq.py:21,27     PIL BILINEAR with antialiasing: triangle filter stretched by the scale
//   transforms.RandomHorizontalFlip()       dataset/ffhq.py:22
//   ToTensor + Normalize(0.5, 0.5)          dataset/ffhq.py:23-24     x = (v/255 - 0.5) / 0.5
//   gt = x*0.5+0.5 -> *255 + 0.5 -> uint8   dataset/ffhq.py:46        (= the resized uint8 pixel)
// The resize reproduces PIL's 8-bit resampler bit for bit: 22-bit fixed-point coefficients (built on the host exactly like
// Resample.c precompute_coeffs / normalize_coeffs_8bpc), horizontal pass rounded to uint8, then vertical pass rounded to uint8.
#include "common.h"
#include "kernels.h"

#define IMG_PREC 22

struct ImgParams {
  const unsigned char* src; int B, Hs, Ws, C;          // [B][Hs][Ws][C] uint8
  int cy, cx, ch, cw;                                  // crop box (top, left, height, width) inside the stored image
  int S;                                               // output edge
  const int* kx; const int* bx; int ksx;               // horizontal: coefficients [S][ksx], bounds [S][2] = (first input column, count)
  const int* ky; const int* by; int ksy;               // vertical
  unsigned char* tmp;                                  // [B][ch][S][C] horizontally resized rows
  const unsigned char* flip;                           // [B] 1 = mirror horizontally (NULL: never)
  float* x0; long long sn, sc, sh, sw;                 // normalised output, element strides of its (B,C,S,S) view
  unsigned char* gts;                                  // [B][S][S][C] uint8 (NULL: skip)
};

__device__ __forceinline__ int img_clip8(int v) { v >>= IMG_PREC; return v < 0 ? 0 : (v > 255 ? 255 : v); }

__global__ void __launch_bounds__(256) img_resize_h_kernel(const ImgParams P) {
  const long long total = (long long)P.B * P.ch * P.S * P.C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % P.C); long long r = i / P.C;
    const int xo = (int)(r % P.S); r /= P.S;
    const int y = (int)(r % P.ch); const int b = (int)(r / P.ch);
    const int xmin = P.bx[2 * xo], cnt = P.bx[2 * xo + 1];
    const unsigned char* row = P.src + (((size_t)b * P.Hs + P.cy + y) * P.Ws + P.cx + xmin) * P.C + c;
    int ss = 1 << (IMG_PREC - 1);
    for (int k = 0; k < cnt; ++k) ss += (int)row[(size_t)k * P.C] * P.kx[xo * P.ksx + k];
    P.tmp[i] = (unsigned char)img_clip8(ss);
  }
}

__global__ void __launch_bounds__(256) img_resize_v_finish_kernel(const ImgParams P) {
  const long long total = (long long)P.B * P.S * P.S * P.C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % P.C); long long r = i / P.C;
    const int xo = (int)(r % P.S); r /= P.S;
    const int yo = (int)(r % P.S); const int b = (int)(r / P.S);
    const int ymin = P.by[2 * yo], cnt = P.by[2 * yo + 1];
    const unsigned char* col = P.tmp + (((size_t)b * P.ch + ymin) * P.S + xo) * P.C + c;
    int ss = 1 << (IMG_PREC - 1);
    for (int k = 0; k < cnt; ++k) ss += (int)col[(size_t)k * P.S * P.C] * P.ky[yo * P.ksy + k];
    const int v = img_clip8(ss);
    const int xd = (P.flip && P.flip[b]) ? P.S - 1 - xo : xo;
    if (P.gts) P.gts[(((size_t)b * P.S + yo) * P.S + xd) * P.C + c] = (unsigned char)v;
    float f = (float)v / 255.0f;                       // ToTensor
    f = (f - 0.5f) / 0.5f;                             // Normalize(0.5, 0.5)
    P.x0[b * P.sn + c * P.sc + yo * P.sh + xd * P.sw] = f;
  }
}

size_t k_image_workspace_bytes(int B, int crop_h, int S, int C) { return (size_t)B * crop_h * S * C; }

int k_image_prepare(const unsigned char* src, int B, int Hs, int Ws, int C, int cy, int cx, int ch, int cw, int S, const int* kx, const int* bx, int ksx,
                    const int* ky, const int* by, int ksy, const unsigned char* flip, float* x0, const long long* strides, unsigned char* gts,
                    unsigned char* ws, hipStream_t st) {
  ImgParams P{src, B, Hs, Ws, C, cy, cx, ch, cw, S, kx, bx, ksx, ky, by, ksy, ws, flip, x0, strides[0], strides[1], strides[2], strides[3], gts};
  long long n1 = (long long)B * ch * S * C, n2 = (long long)B * S * S * C;
  int g1 = (int)((n1 + 255) / 256), g2 = (int)((n2 + 255) / 256);
  if (g1 > 8192) g1 = 8192; if (g2 > 8192) g2 = 8192;
  hipLaunchKernelGGL(img_resize_h_kernel, dim3(g1), dim3(256), 0, st, P);
  hipLaunchKernelGGL(img_resize_v_finish_kernel, dim3(g2), dim3(256), 0, st, P);
  return pdae_launch_status("image_prepare");
}
