// winograd.hip: F(2x2, 3x3) forward convolution for weight-constant 3x3 / stride-1 / pad-1 layers (f16x3 arithmetic)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

bool wino_ok(int math, int KH, int KW, int stride, int pad, int up, int C0, int C1, int H, int W, int N, int Nout);
size_t wino_wprep_bytes(int Nout, int C);
float wino_wscale(int C);
int wino_wprep(const float* w, int Nout, int C, unsigned short* wp, hipStream_t s);
int wino_fwd(const float* x, int N, int H, int W, int C, const unsigned short* wp, int Nout, const float* bias, float* y, hipStream_t s);
