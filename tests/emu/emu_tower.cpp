// TEST INFRASTRUCTURE: runs the tower-glue kernels of shogidrl_amd/csrc/ksh_tower_kernels.h on the host, launched the
// way ksh_lib.hip launches them (same row splits, same workspace layout).
#include <hip/hip_runtime.h>

#include "../../shogidrl_amd/csrc/ksh_tower_kernels.h"

namespace {
struct A {
    const void *x, *res, *dy, *y; void *out, *out2; int dtype, relu; KshRowSplit sp; const float *w, *b, *mean, *invstd, *shift;
    float *rm, *rv, *save_mean, *save_invstd, *partial, *coef, *dw, *db; const float *pre_bias; float momentum, eps;
    int HW, C, N, spb; const float *s, *dp; float *p;
} a;
template <typename T> void stats_t() { ksh_bn_stats_kernel<T>((const T *)a.x, a.sp, a.shift, a.pre_bias, a.partial); }
void stats_entry(void *) { a.dtype ? stats_t<ksh_bf16>() : stats_t<float>(); }
void fin_entry(void *) { ksh_bn_finalize_kernel(a.partial, a.sp, a.shift, a.pre_bias, a.w, a.b, a.rm, a.rv, a.momentum, a.eps, a.save_mean, a.save_invstd, a.coef); }
template <typename T> void apply_t() {
    const T *x = (const T *)a.x, *r = (const T *)a.res; T *y = (T *)a.out;
    if (a.relu && r) ksh_bn_apply_kernel<T, 1, 1>(x, r, y, a.sp, a.coef);
    else if (a.relu) ksh_bn_apply_kernel<T, 1, 0>(x, r, y, a.sp, a.coef);
    else if (r) ksh_bn_apply_kernel<T, 0, 1>(x, r, y, a.sp, a.coef);
    else ksh_bn_apply_kernel<T, 0, 0>(x, r, y, a.sp, a.coef);
}
void apply_entry(void *) { a.dtype ? apply_t<ksh_bf16>() : apply_t<float>(); }
template <typename T> void bred_t() {
    if (a.relu) ksh_bn_bwd_reduce_kernel<T, 1>((const T *)a.dy, (const T *)a.x, (const T *)a.y, a.sp, a.mean, a.partial);
    else ksh_bn_bwd_reduce_kernel<T, 0>((const T *)a.dy, (const T *)a.x, (const T *)a.y, a.sp, a.mean, a.partial);
}
void bred_entry(void *) { a.dtype ? bred_t<ksh_bf16>() : bred_t<float>(); }
void bfin_entry(void *) { ksh_bn_bwd_finalize_kernel(a.partial, a.sp, a.w, a.invstd, a.dw, a.db, a.coef); }
template <typename T> void bapp_t() {
    const T *g = (const T *)a.dy, *x = (const T *)a.x, *y = (const T *)a.y; T *dx = (T *)a.out, *dr = (T *)a.out2;
    if (a.relu && dr) ksh_bn_bwd_apply_kernel<T, 1, 1>(g, x, y, dx, dr, a.sp, a.mean, a.coef);
    else if (a.relu) ksh_bn_bwd_apply_kernel<T, 1, 0>(g, x, y, dx, dr, a.sp, a.mean, a.coef);
    else if (dr) ksh_bn_bwd_apply_kernel<T, 0, 1>(g, x, y, dx, dr, a.sp, a.mean, a.coef);
    else ksh_bn_bwd_apply_kernel<T, 0, 0>(g, x, y, dx, dr, a.sp, a.mean, a.coef);
}
void bapp_entry(void *) { a.dtype ? bapp_t<ksh_bf16>() : bapp_t<float>(); }
template <typename T> void pool_t() { ksh_se_pool_kernel<T>((const T *)a.x, a.HW, a.C, a.p); }
void pool_entry(void *) { a.dtype ? pool_t<ksh_bf16>() : pool_t<float>(); }
template <typename T> void seap_t() { ksh_se_apply_kernel<T>((const T *)a.x, a.s, (const T *)a.res, (T *)a.out, a.HW, a.C); }
void seap_entry(void *) { a.dtype ? seap_t<ksh_bf16>() : seap_t<float>(); }
template <typename T> void sered_t() { ksh_se_bwd_reduce_kernel<T>((const T *)a.dy, (const T *)a.y, (const T *)a.x, a.HW, a.C, a.p); }
void sered_entry(void *) { a.dtype ? sered_t<ksh_bf16>() : sered_t<float>(); }
template <typename T> void sebap_t() { ksh_se_bwd_apply_kernel<T>((const T *)a.dy, (const T *)a.y, a.s, a.dp, (T *)a.out, (T *)a.out2, a.HW, a.C); }
void sebap_entry(void *) { a.dtype ? sebap_t<ksh_bf16>() : sebap_t<float>(); }
template <typename T> void appool_t() { ksh_bn_apply_pool_kernel<T>((const T *)a.x, (T *)a.out, a.HW, a.C, a.coef, a.p); }
void appool_entry(void *) { a.dtype ? appool_t<ksh_bf16>() : appool_t<float>(); }
template <typename T> void sebn_t() {
    ksh_se_bwd_apply_bnred_kernel<T>((const T *)a.dy, (const T *)a.y, a.s, a.dp, (const T *)a.x, a.mean, (T *)a.out, (T *)a.out2, a.N, a.HW,
                                     a.C, a.spb, a.partial);
}
void sebn_entry(void *) { a.dtype ? sebn_t<ksh_bf16>() : sebn_t<float>(); }
void run(void (*e)(void *), int blocks) { for (int b = 0; b < blocks; b++) emu::run_block(e, nullptr, b, KSH_TW_THREADS); }
}  // namespace

extern "C" {
long long emu_tower_workspace_floats(int C) { return ((long long)KSH_TW_MAXG * 2 + 3) * C; }
void emu_bn_fwd_train(const void *x, const void *res, void *y, int dtype, long long M, int C, const float *pre_bias, const float *w, const float *b,
                      float *rm, float *rv, float momentum, float eps, int relu, float *save_mean, float *save_invstd,
                      float *ws) {
    a = A{};
    a.x = x; a.res = res; a.out = y; a.dtype = dtype; a.relu = relu; a.w = w; a.b = b; a.rm = rm; a.rv = rv; a.shift = rm; a.pre_bias = pre_bias;
    a.momentum = momentum; a.eps = eps; a.save_mean = save_mean; a.save_invstd = save_invstd;
    a.partial = ws; a.coef = ws + (size_t)KSH_TW_MAXG * 2 * C;
    a.sp = ksh_row_split(M, C, KSH_TW_MAXG);
    run(stats_entry, a.sp.G);
    run(fin_entry, (C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH);
    a.sp = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    run(apply_entry, a.sp.G);
}
void emu_bn_fwd_eval(const void *x, const void *res, void *y, int dtype, long long M, int C, float *coef, int relu) {
    a = A{};
    a.x = x; a.res = res; a.out = y; a.dtype = dtype; a.relu = relu; a.coef = coef;
    a.sp = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    run(apply_entry, a.sp.G);
}
void emu_bn_bwd(const void *dy, const void *x, const void *y, void *dx, void *dres, int dtype, long long M, int C, const float *w,
                const float *mean, const float *invstd, int relu, float *dw, float *db, float *ws) {
    a = A{};
    a.dy = dy; a.x = x; a.y = y; a.out = dx; a.out2 = dres; a.dtype = dtype; a.relu = relu; a.w = w; a.mean = mean;
    a.invstd = invstd; a.dw = dw; a.db = db; a.partial = ws; a.coef = ws + (size_t)KSH_TW_MAXG * 2 * C;
    a.sp = ksh_row_split(M, C, KSH_TW_MAXG);
    run(bred_entry, a.sp.G);
    run(bfin_entry, (C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH);
    a.sp = ksh_row_split(M, C, 4 * KSH_TW_MAXG);
    run(bapp_entry, a.sp.G);
}
void emu_se_pool(const void *x, int dtype, int N, int HW, int C, float *p) {
    a = A{}; a.x = x; a.dtype = dtype; a.HW = HW; a.C = C; a.p = p; run(pool_entry, N);
}
void emu_se_apply(const void *x, const float *s, const void *r, void *z, int dtype, int N, int HW, int C) {
    a = A{}; a.x = x; a.s = s; a.res = r; a.out = z; a.dtype = dtype; a.HW = HW; a.C = C; run(seap_entry, N);
}
void emu_se_bwd_reduce(const void *dy, const void *z, const void *x, int dtype, int N, int HW, int C, float *ds) {
    a = A{}; a.dy = dy; a.y = z; a.x = x; a.dtype = dtype; a.HW = HW; a.C = C; a.p = ds; run(sered_entry, N);
}
void emu_se_bwd_apply(const void *dy, const void *z, const float *s, const float *dp, void *dx, void *dr, int dtype, int N,
                      int HW, int C) {
    a = A{}; a.dy = dy; a.y = z; a.s = s; a.dp = dp; a.out = dx; a.out2 = dr; a.dtype = dtype; a.HW = HW; a.C = C;
    run(sebap_entry, N);
}
void emu_bn_pool_fwd_train(const void *x, void *y, int dtype, int N, int HW, int C, const float *pre_bias, const float *w,
                           const float *b, float *rm, float *rv, float momentum, float eps, float *save_mean,
                           float *save_invstd, float *pooled, float *ws) {
    a = A{};
    a.x = x; a.out = y; a.dtype = dtype; a.w = w; a.b = b; a.rm = rm; a.rv = rv; a.shift = rm; a.pre_bias = pre_bias;
    a.momentum = momentum; a.eps = eps; a.save_mean = save_mean; a.save_invstd = save_invstd; a.HW = HW; a.C = C; a.p = pooled;
    a.partial = ws; a.coef = ws + (size_t)KSH_TW_MAXG * 2 * C;
    a.sp = ksh_row_split((long long)N * HW, C, KSH_TW_MAXG);
    run(stats_entry, a.sp.G);
    run(fin_entry, (C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH);
    run(appool_entry, N);
}
void emu_se_bn_bwd(const void *dz, const void *z, const float *s, const float *dp, const void *x, void *dy2, void *dx, void *dr,
                   int dtype, int N, int HW, int C, const float *w, const float *mean, const float *invstd, float *dw, float *db,
                   float *ws) {
    a = A{};
    a.dy = dz; a.y = z; a.s = s; a.dp = dp; a.x = x; a.out = dy2; a.out2 = dr; a.dtype = dtype; a.N = N; a.HW = HW; a.C = C;
    a.mean = mean; a.invstd = invstd; a.w = w; a.dw = dw; a.db = db; a.partial = ws; a.coef = ws + (size_t)KSH_TW_MAXG * 2 * C;
    a.spb = (N + KSH_TW_MAXG - 1) / KSH_TW_MAXG;
    a.sp.M = (long long)N * HW; a.sp.R = (long long)a.spb * HW; a.sp.C = C; a.sp.G = (N + a.spb - 1) / a.spb;
    run(sebn_entry, a.sp.G);
    run(bfin_entry, (C + KSH_TW_FIN_CH - 1) / KSH_TW_FIN_CH);
    // batch-norm backward apply on the gradient just produced
    a.dy = dy2; a.y = nullptr; a.out = dx; a.out2 = nullptr; a.relu = 0;
    a.sp = ksh_row_split((long long)N * HW, C, 4 * KSH_TW_MAXG);
    run(bapp_entry, a.sp.G);
}
}
