// internal launcher prototypes (norm.hip / elementwise.hip); the public C ABI is include/pdae_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

size_t k_gn_workspace_floats(int N, int C);
int k_gn_stats(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, float* mean, float* rstd, float* ws, hipStream_t st);
int k_gn_coef_from_conv_stats(int N, int HW, int C0, int C1, int G, float eps, const float* part0, int tpi0, const float* part1, int tpi1,
                              const float* gamma, const float* beta, const float* ss, const float* zss, float* mean, float* rstd, float* coef,
                              hipStream_t st);
int k_gn_stats_quads(const float* x, int N, int HW, int C, int tpi, float* part, hipStream_t st);
int k_gn_stats_coef(const float* x0, int C0, const float* x1, int C1, int N, int HW, int G, float eps, const float* gamma, const float* beta,
                    const float* ss, const float* zss, float* mean, float* rstd, float* coef, float* ws, hipStream_t st, unsigned* ticket = nullptr);
int k_gn_coef(int N, int C, int G, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* ss, const float* zss,
              float* coef, hipStream_t st);
int k_gn_apply(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, const float* coef, int act, int mode, float* y, float* xpool,
               float drop_p, unsigned long long seed, unsigned long long offset, hipStream_t st);
int k_gn_bwd(const float* x0, int C0, const float* x1, int C1, int N, int H, int W, int G, const float* coef, const float* rstd, const float* gamma,
             const float* beta, const float* ss, const float* zss, const float* dA, int act, int mode, float drop_p, unsigned long long seed,
             unsigned long long offset, const float* add, float* dx0, int acc0, float* dx1, int acc1, float* dgamma, float* dbeta, int acc_param,
             float* dss, float* dzss, float* ws, hipStream_t st, float* dx0_amax = nullptr, unsigned* ticket = nullptr,
             const float* parts = nullptr, int parts_tiles = 0);

int k_timestep_embedding(const long long* t, const float* freqs, int N, int dim, float* out, hipStream_t st);
int k_mlp_modln_fwd(const float* u, const float* e, const float* gamma, const float* beta, int R, int C, int norm, int act, float eps, float* y,
                    float* mean, float* rstd, hipStream_t st);
int k_mlp_modln_bwd(const float* u, const float* e, const float* gamma, const float* beta, const float* mean, const float* rstd, const float* dy,
                    int R, int C, int norm, int act, float* du, float* de, float* tg, float* tb, hipStream_t st);
int k_amax(const float* x, size_t n, float* out, hipStream_t st);
int k_silu(const float* x, float* y, size_t n, hipStream_t st);
int k_subsample2(const float* x, int N, int H, int W, int C, float* y, hipStream_t st);
int k_zero_insert2(const float* x, int N, int Ho, int Wo, int C, float* y, hipStream_t st);
int k_silu_bwd(const float* x, const float* dy, float* dx, size_t n, int acc, hipStream_t st);
int k_axpby(const float* x, float* y, size_t n, float alpha, float beta, hipStream_t st);
int k_embedding(const float* table, const long long* idx, int N, int D, float* out, int acc, hipStream_t st);
int k_embedding_bwd(const float* dout, const long long* idx, int N, int D, float* dtable, hipStream_t st);
int k_to_nhwc(const float* x, long long sn, long long sc, long long sh, long long sw, int N, int C, int H, int W, float* y, hipStream_t st);
int k_from_nhwc(const float* x, int N, int C, int H, int W, float* y, long long sn, long long sc, long long sh, long long sw, hipStream_t st);
int k_q_sample(const float* x0, const float* noise, const long long* t, const float* ta, const float* tb, int N, size_t per, float* xt, hipStream_t st);
int k_loss(const float* noise, const float* eps, const float* g, const long long* t, const float* tc, const float* tw, int N, size_t per, int l1,
           float scale, float* loss, float* deps, float* dg, float* ws, hipStream_t st);
int k_ddim_step(const float* x, const float* eps, const float* g, size_t total, float c_shift, float ra, float rm1, float sab, float s1ab, int clamp,
                float* out, hipStream_t st);
int k_ddpm_step(const float* x, const float* eps, const float* g, const float* z, size_t total, float cx, float ce, float cs, float sigma, float* out,
                hipStream_t st);
int k_axpby_rows(const float* a, const float* b, const float* ca, const float* cb, int N, size_t per, float* out, hipStream_t st);
int k_ddim_step_rows(const float* x, const float* eps, const float* g, const float* coef, int N, size_t per, int clamp, float* out, hipStream_t st);
int k_ddpm_step_rows(const float* x, const float* eps, const float* g, const float* noise, const float* lrange, const float* coef, int N, size_t per,
                     float* out, hipStream_t st);
int k_adam_ema(float* p, const float* g, float* m, float* v, float* ema, size_t n, float lr, float b1, float b2, float eps, float wd, int decoupled,
               float step_size, float inv_sqrt_bc2, float grad_scale, float ema_decay, unsigned int* guard, int count_skip, hipStream_t st);
int k_softmax(float* s, long long rows, int T, hipStream_t st);
int k_softmax_bwd(const float* p, float* dp, long long rows, int T, hipStream_t st);
size_t k_colsum_workspace_floats(long long M, int C);
int k_colsum(const float* x, long long M, int C, float* out, int acc, float* ws, hipStream_t st);

// metric.hip / image.hip
size_t k_ssim_mse_workspace_floats(int N, int C, int H, int W);
int k_ssim_mse(const float* a, const long long* as, const float* b, const long long* bs, int N, int C, int H, int W, float mul, float add,
               const float* window, float* ssim, float* mse, float* ws, hipStream_t st);
size_t k_image_workspace_bytes(int B, int crop_h, int S, int C);
int k_image_prepare(const unsigned char* src, int B, int Hs, int Ws, int C, int cy, int cx, int ch, int cw, int S, const int* kx, const int* bx, int ksx,
                    const int* ky, const int* by, int ksy, const unsigned char* flip, float* x0, const long long* strides, unsigned char* gts,
                    unsigned char* ws, hipStream_t st);

// attention.hip
bool attn_fused_ok(int T, int ch, int C, int heads);
int k_attn_fwd(const float* qkv, int N, int T, int C, int heads, int new_order, float* out, float* lse, hipStream_t st);
int k_attn_bwd(const float* qkv, const float* o, const float* lse, const float* d_o, int N, int T, int C, int heads, int new_order, float* dqkv,
               float* dvec, hipStream_t st);

// comm.hip
int k_comm_unique_id(const char* librccl_path, void* id128);
int k_comm_init(const char* librccl_path, const void* id128, int nranks, int rank, void** comm);
int k_allreduce(void* comm, void* buf, size_t count, int dtype, int op, hipStream_t st);
int k_comm_destroy(void* comm);
