// Parameter blocks of the f32-MFMA implicit-GEMM kernel family (internal; the C ABI is include/pdae_hip.h).
#pragma once
#include <hip/hip_runtime.h>

enum { OP_CONV_KC = 0, OP_DENSE_KC = 1, OP_DENSE_OC = 2, OP_DGRAD_OC = 3, OP_GATHER_OC = 4 };

// Gather geometry over NHWC activations (optionally a virtual channel-concat of two tensors).
struct ConvGeom {
  const float* src0;
  const float* src1;
  int C0, C1, Cin;   // channels of src0 / src1, Cin = C0 + C1
  int Hs, Ws;        // stored spatial size
  int Hl, Wl;        // logical size seen by the taps (2x when `up` or `dil`)
  int Ho, Wo;        // spatial size of the GEMM rows (output pixels)
  int KH, KW, stride, pad;
  int up;            // logical = nearest-upsample x2 of stored
  int dil;           // logical = zero-dilation x2 of stored (dgrad of a stride-2 conv)
};

struct OpParams {
  const float* p;    // dense / dgrad base pointer
  long long ld;      // leading dimension (dense)
  long long so, si;  // outer / inner batch strides (dense)
  ConvGeom g;        // CONV_KC / GATHER_OC
  int dgT, dgCout, dgWCin, dgCiOff;   // DGRAD_OC: taps, Cout, weight Cin, first input channel
  int kpT, kpC;      // chunk-major K order of the vector conv paths: taps, channels (0 = natural order)
};

struct GemmParams {
  int M, N, K;
  int splitk, kchunk;          // split-K over blockIdx.z when splitk > 1 (kchunk multiple of 32)
  long long split_stride;      // workspace stride between splits (= M*N)
  int Bi;                      // inner batch count when splitk == 1 (blockIdx.z = bo*Bi + bi)
  OpParams a, b;
  float* C;
  long long ldc, sCo, sCi;
  const float* bias;           // [N] or null
  const float* res;            // residual or null
  long long ldr;
  int res_mode;                // 0 none, 1 same row index, 2 residual stored at half resolution (nearest-up)
  int rHo, rWo;                // output spatial dims for res_mode 2
  float alpha;
  int accumulate;              // C += ...
};

int igemm_conv_fwd(const GemmParams& P, int tile, int math, hipStream_t s);
int igemm_conv_dgrad(const GemmParams& P, int tile, int math, hipStream_t s);
int igemm_conv_wgrad(const GemmParams& P, int tile, int splits, int math, hipStream_t s);
// db_part / db: optional bias-gradient partial rows [db_rows][C] summed into db[C] (same accumulate flag) by extra blocks of the same launch
int igemm_splitk_reduce(const float* ws, float* out, long long n, int splits, int accumulate, hipStream_t s, const float* db_part = nullptr,
                        int db_rows = 0, int C = 0, float* db = nullptr);
int igemm_dense(int transA, int transB, const GemmParams& P, int zdim, hipStream_t s);

// conv3x3p.hip: 3x3 stride-1 "patch" kernel on the bf16 MFMA pipe (math modes 1..3)
struct PatchSkip { const float* s0; const float* s1; int C0, C1; const unsigned short* wps; const float* bias; };   // fused 1x1 skip conv
// GroupNorm-backward sums from a DATA-GRADIENT launch's epilogue (pdae_conv_gnbwd_arm): x = [x0 | x1] the GroupNorm's raw input (C0 + C1 = the
// launch's output channels), coef = [mu | a | b] of its forward, part = [N][tiles per image][C][2] x (sum dv, sum dv (x - mu)), dv = dA silu'(a (x - mu) + b)
struct PatchGnb { const float* x0; const float* x1; int C0, C1; const float* coef; float* part; };
bool conv3x3p_ok(int math, int KH, int KW, int stride, int pad, int C1, int C, int H, int W, int N, int Nout, bool fill);
size_t conv3x3p_wprep_bytes(int math, int Nout, int C, int H, int W, int N);
int conv3x3p_wprep(int math, const float* w, int Nout, int C, int transposed, unsigned short* wp, hipStream_t s, int H, int W, int N);
int conv3x3p_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const unsigned short* wp, int Nout,
                    float* y, const float* bias, const float* res, int res_mode, int accumulate, hipStream_t s, const float* x1 = nullptr,
                    int C0 = 0, const float* coef = nullptr, int act = 0, const PatchSkip* sk = nullptr, const float* amax = nullptr,
                    float* stat_part = nullptr, const PatchGnb* gb = nullptr);
int conv3x3p_gnb_tiles(int math, int C, int H, int W, int N, int Nout, int C0, int C1);      // tiles per image of PatchGnb::part, 0 = the launch cannot leave the sums
void conv3x3p_arm_stats(float* part);          // one-shot request of pdae_conv_stats_arm (thread-local)
float* conv3x3p_take_stats();                  // ... taken AND cleared by the next forward entry point, first thing, on every return path
size_t conv3x3p_stats_bytes(int math, int C, int H, int W, int N, int Nout, int fused_skip_chunks, int* tpi);
size_t conv3x3p_skip_wprep_bytes(int math, int Nout, int Cs);
int conv3x3p_skip_wprep(int math, const float* w, int Nout, int Cs, int Cmain, unsigned short* wp, hipStream_t s, int H, int W, int N);
bool conv3x3p_skip_ok(int math, int C, int H, int W, int N, int Nout, int up, int Cs0, int Cs1);

// conv3x3w.hip: 3x3 stride-1 weight gradient with transposing LDS reads (math modes 1..3)
bool conv3x3w_ok(int math, int KH, int KW, int stride, int pad, int C1, int C, int H, int W, int N, int Cout);
size_t conv3x3w_workspace_bytes(int N, int H, int W, int C, int Cout);
int conv3x3w_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const float* dy, int Cout, float* dw,
                    int accumulate, float* ws, size_t ws_bytes, hipStream_t s, float** db_part = nullptr, int* db_rows = nullptr,
                    const float* dy_amax = nullptr, float* db = nullptr, const float* x1 = nullptr, int C0 = 0, const float* coef = nullptr, int act = 0);
// conv3x3v.hip: the producer / consumer form of the same weight gradient (one workgroup per CU: four matrix waves fed by four staging waves);
// conv3x3w_launch / conv3x3w_workspace_bytes route to it for the shapes it takes (knob PDAE_W3V)
bool conv3x3v_ok(int math, int C, int H, int W, int N, int Cout);
size_t conv3x3v_workspace_bytes(int N, int H, int W, int C, int Cout);
int conv3x3v_launch(int math, const float* x, int N, int Hs, int Ws, int C, int H, int W, int up, const float* dy, int Cout, float* dw,
                    int accumulate, float* ws, size_t ws_bytes, hipStream_t s, float** db_part, int* db_rows, const float* dy_amax, float* db,
                    const float* x1, int C0, const float* coef, int act);
// the same kernel with GroupNorm + SiLU recomputed on the RAW two-source input [x | x1] while it is staged (coef = [mu | a | b], pdae_gn_coef)
bool conv3x3w_gn_ok(int math, int KH, int KW, int stride, int pad, int C0, int C1, int H, int W, int N, int Cout);

// conv1x1.hip: 1x1 convolution (forward / data gradient) on prepared weights, activations staged through LDS
bool conv1x1_ok(int math, int KH, int KW, int stride, int pad, int up, int C0, int C1, int Nout);
size_t conv1x1_wprep_bytes(int math, int Nrows, int C, long long M);
int conv1x1_wprep(int math, const float* w, int Nrows, int C, int transposed, unsigned short* wp, hipStream_t s);
int conv1x1_launch(int math, const float* x0, int C0, const float* x1, int C1, long long M, const unsigned short* wp, int Nrows, int row_off,
                   int Nout, float* y, const float* bias, const float* res, int res_mode, int H, int W, int accumulate, hipStream_t s,
                   const float* amax = nullptr);

// skinny.hip: M <= 32 linear layers (one wave per output feature)
// conv3x3w.hip: dedicated 1x1 weight-gradient kernel
bool conv1x1w_ok(int math, int KH, int KW, int stride, int pad, int up, int C0, int C1, long long M, int Cout);
size_t conv1x1w_workspace_bytes(long long M, int C, int Cout);
int conv1x1w_launch(int math, const float* x0, int C0, const float* x1, int C1, long long M, const float* dy, int Cout, float* dw, int accumulate,
                    float* ws, size_t ws_bytes, hipStream_t s, float** db_part, int* db_rows, const float* dy_amax, float* db = nullptr);
int linear_bwd_group_launch(const void* items, const int* first, int n_items, int total_blocks, int M, int K, hipStream_t s);
int skinny_group_launch(const void* items, const int* first, int n_items, int total, int M, int K, hipStream_t s);
bool skinny_ok(int transA, int transB, int M, int N, int K, float alpha, long long lda, long long ldb, const float* A, const float* B, int batch);
int skinny_launch(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, const float* bias, int M, int N, int K,
                  int accumulate, hipStream_t s);

// convedge.hip: the 3-channel edges of the U-Nets on the matrix cores (head forward as 1x1 conv to 9 Cout virtual channels + shifted sum)
bool edge_head_ok(int KH, int KW, int stride, int pad, int up, int C1, int C, int Cout);
int edge_head_fwd(const float* x, int N, int H, int W, int C, const float* w, int Cout, const float* bias, float* y, hipStream_t s);
bool edge_head_wgrad_ok(int KH, int KW, int stride, int pad, int up, int C1, int C, int Cout);
int edge_head_wgrad(const float* x, int N, int H, int W, int C, const float* dy, int Cout, float* dw, int accumulate, float* ws, size_t ws_bytes,
                    hipStream_t s);
bool edge_in_ok(int KH, int KW, int stride, int pad, int up, int C1, int Cin, int Nout);
int edge_in_conv(const float* x, int N, int H, int W, int Cin, const float* w, int transposed, int Nout, const float* bias, float* y, int accumulate,
                 hipStream_t s);
// convhead.hip: 3x3 convolutions with Cout <= 4 (image heads) as exact fp32 FMA kernels
bool convhead_ok(int KH, int KW, int stride, int pad, int up, int C1, int C, int Cout);
int convhead_fwd(const float* x, int N, int H, int W, int C, const float* w, int Cout, const float* bias, float* y, hipStream_t s);
size_t convhead_wgrad_workspace_bytes(int N, int H, int W, int C, int Cout);
int convhead_wgrad(const float* x, int N, int H, int W, int C, const float* dy, int Cout, float* dw, int accumulate, float* ws, size_t ws_bytes,
                   hipStream_t s);
