// ksh_tower_kernels.h -- the HBM-bound glue between the convolutions of the actor-critic tower
// (keisei/training/models/resnet_tower.py:28-46, 82-83: `relu(bn(conv(x)))`, `relu(se(bn(conv)) + x)`), fused for
// gfx950.  The convolutions and linear layers stay in PyTorch-ROCm (MIOpen / hipBLASLt); what is here is the
// normalisation / activation / residual work around them, which on channels-last bf16 activations costs more
// than the convolutions when left to the framework's generic kernels (profiles/README.md):
//
//   ksh_bn_stats_kernel        per-channel partial (sum, sum of squares) of x[M][C]          reads x once
//   ksh_bn_finalize_kernel     Chan-combines the partials in double -> mean, 1/std, running-stat update
//                              (nn.BatchNorm2d training semantics: biased variance for the normalisation,
//                              unbiased for running_var, momentum blend), per-channel coefficients
//   ksh_bn_apply_kernel        y = [relu]((x - mean) * scale + bias [+ residual])             reads x (+res), writes y
//   ksh_bn_bwd_reduce_kernel   partial sums of dz and dz * (x - mean), dz = dy * (y > 0) when ReLU is fused
//   ksh_bn_bwd_finalize_kernel d_weight, d_bias, per-channel coefficients of dx
//   ksh_bn_bwd_apply_kernel    dx = (dz - mean(dz) - (x - mean) * invstd^2 * mean(dz * (x - mean))) * w * invstd
//                              [, d_residual = dz]
//   ksh_se_pool_kernel / ksh_se_apply_kernel / ksh_se_bwd_reduce_kernel / ksh_se_bwd_apply_kernel
//                              squeeze-excitation tail: p = mean_hw(x); z = relu(x * s + r) and its backward
//
// Layout: activations are channels-last, i.e. a dense [M][C] matrix with M = N*H*W rows of C channels
// (C % 8 == 0, 256 % (C/8) == 0), bf16 or float32.  A thread owns 8 consecutive channels (one 16-byte load for
// bf16) of every (256 / (C/8))-th row of its block's contiguous row chunk, so a wave reads whole 512-byte rows;
// per-channel reductions finish in LDS and leave one partial row per block -- no atomics, deterministic sums.
#pragma once
#include "ksh_device_util.h"

#define KSH_TW_THREADS 256
#define KSH_TW_MAXG 1024   // upper bound of the partial-row count (workspace sizing)

template <typename T> struct KshVec8;
template <> struct KshVec8<ksh_bf16> {
    static __device__ __forceinline__ void load(const ksh_bf16 *p, float v[8]) {
        const uint4 u = *reinterpret_cast<const uint4 *>(p);
        v[0] = ksh_u2f(u.x << 16); v[1] = ksh_u2f(u.x & 0xFFFF0000u);
        v[2] = ksh_u2f(u.y << 16); v[3] = ksh_u2f(u.y & 0xFFFF0000u);
        v[4] = ksh_u2f(u.z << 16); v[5] = ksh_u2f(u.z & 0xFFFF0000u);
        v[6] = ksh_u2f(u.w << 16); v[7] = ksh_u2f(u.w & 0xFFFF0000u);
    }
    static __device__ __forceinline__ void store(ksh_bf16 *p, const float v[8]) {
        uint4 u;
        u.x = (uint32_t)ksh_bf16_bits(v[0]) | ((uint32_t)ksh_bf16_bits(v[1]) << 16);
        u.y = (uint32_t)ksh_bf16_bits(v[2]) | ((uint32_t)ksh_bf16_bits(v[3]) << 16);
        u.z = (uint32_t)ksh_bf16_bits(v[4]) | ((uint32_t)ksh_bf16_bits(v[5]) << 16);
        u.w = (uint32_t)ksh_bf16_bits(v[6]) | ((uint32_t)ksh_bf16_bits(v[7]) << 16);
        *reinterpret_cast<uint4 *>(p) = u;
    }
    // the values as a store + load round trip would return them
    static __device__ __forceinline__ void round(float v[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = ksh_u2f((uint32_t)ksh_bf16_bits(v[k]) << 16);
    }
};
template <> struct KshVec8<float> {
    static __device__ __forceinline__ void round(float *) {}
    static __device__ __forceinline__ void load(const float *p, float v[8]) {
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float *p, const float v[8]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// Row chunk [r0, r1) of block `blk` when M rows are split into G chunks of R rows (R a multiple of the rows a
// block covers per pass, G * R >= M).
struct KshRowSplit {
    long long M, R;
    int C, G;
};

// rows split into at most `max_g` chunks, each a multiple of the rows one block pass covers (host side)
static inline KshRowSplit ksh_row_split(long long M, int C, int max_g) {
    const int rpp = KSH_TW_THREADS / (C >> 3);
    long long R = (M + max_g - 1) / max_g;
    const long long min_rows = (long long)rpp * 4;         // at least four passes per block
    if (R < min_rows) R = min_rows;
    R = (R + rpp - 1) / rpp * rpp;
    KshRowSplit sp;
    sp.M = M;
    sp.R = R;
    sp.C = C;
    sp.G = (int)((M + R - 1) / R);
    return sp;
}

// Sum `acc[2][8]` over the threads of a block that share a channel group; result valid in the threads of row
// lane 0 (threadIdx.x < C/8).  `red` is 256 * 16 floats of LDS.
__device__ __forceinline__ void ksh_tw_reduce16(float *red, float a[8], float b[8], int cg, int tcg, int tr) {
    // red[k][thread]: conflict-free both for the writes and for the row-lane-0 threads' reads
#pragma unroll
    for (int k = 0; k < 8; k++) {
        red[k * KSH_TW_THREADS + (int)threadIdx.x] = a[k];
        red[(8 + k) * KSH_TW_THREADS + (int)threadIdx.x] = b[k];
    }
    __syncthreads();
    if (tr == 0) {
        const int rpp = KSH_TW_THREADS / cg;
        for (int j = 1; j < rpp; j++) {
            const float *o = red + j * cg + tcg;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                a[k] += o[k * KSH_TW_THREADS];
                b[k] += o[(8 + k) * KSH_TW_THREADS];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ BN forward
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_bn_stats_kernel(const T *__restrict__ x, KshRowSplit sp,
                                                                      const float *__restrict__ shift /*[C] or null*/,
                                                                      const float *__restrict__ pre_bias /*[C] or null*/,
                                                                      float *__restrict__ partial /*[G][2][C]*/) {
    __shared__ float red[KSH_TW_THREADS * 16];
    const int cg = sp.C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const long long r0 = (long long)blockIdx.x * sp.R;
    const long long r1 = r0 + sp.R < sp.M ? r0 + sp.R : sp.M;
    // sums of (x - K) and (x - K)^2 with K = the running mean (an estimate of the batch mean that removes the
    // cancellation of E[x^2] - E[x]^2 when |mean| >> std; any K gives the same exact result)
    float s[8], q[8], kk[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        s[k] = q[k] = 0.0f;
        kk[k] = shift != nullptr ? shift[tcg * 8 + k] - (pre_bias != nullptr ? pre_bias[tcg * 8 + k] : 0.0f) : 0.0f;
    }
#pragma unroll 4
    for (long long r = r0 + tr; r < r1; r += rpp) {
        float v[8];
        KshVec8<T>::load(x + r * sp.C + tcg * 8, v);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float d = v[k] - kk[k];
            s[k] += d;
            q[k] += d * d;
        }
    }
    ksh_tw_reduce16(red, s, q, cg, tcg, tr);
    if (tr == 0) {
        float *o = partial + (size_t)blockIdx.x * 2 * sp.C + tcg * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            o[k] = s[k];
            o[sp.C + k] = q[k];
        }
    }
}

// Per-channel sums of the G partial rows in double: a 256-thread block covers 16 channels x 16 partial-row lanes
// (launch with ceil(C / 16) blocks); the loads of a lane are independent, so they are issued eight at a time.
// Returns true in the threads (lane 0) that hold the totals.
#define KSH_TW_FIN_CH 16
__device__ __forceinline__ bool ksh_tw_sum_partials(const float *__restrict__ partial, const KshRowSplit &sp, int &c,
                                                    double &s, double &q) {
    __shared__ double acc[2][KSH_TW_THREADS];
    constexpr int LANES = KSH_TW_THREADS / KSH_TW_FIN_CH;
    const int cl = (int)threadIdx.x % KSH_TW_FIN_CH, j = (int)threadIdx.x / KSH_TW_FIN_CH;
    c = (int)blockIdx.x * KSH_TW_FIN_CH + cl;
    s = q = 0.0;
    const int geff = (int)((sp.M + sp.R - 1) / sp.R);
    if (c < sp.C) {
        int g = j;
        for (; g + 7 * LANES < geff; g += 8 * LANES) {
            float a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                a[u] = partial[(size_t)(g + u * LANES) * 2 * sp.C + c];
                b[u] = partial[(size_t)(g + u * LANES) * 2 * sp.C + sp.C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                s += (double)a[u];
                q += (double)b[u];
            }
        }
        for (; g < geff; g += LANES) {
            s += (double)partial[(size_t)g * 2 * sp.C + c];
            q += (double)partial[(size_t)g * 2 * sp.C + sp.C + c];
        }
    }
    acc[0][threadIdx.x] = s;
    acc[1][threadIdx.x] = q;
    __syncthreads();
    if (j != 0 || c >= sp.C) return false;
    for (int i = 1; i < LANES; i++) {
        s += acc[0][i * KSH_TW_FIN_CH + cl];
        q += acc[1][i * KSH_TW_FIN_CH + cl];
    }
    return true;
}

// launch with ceil(C / 16) blocks of 256 threads.  coef: [0] scale = w * invstd, [1] mean, [2] bias;
// running_* may be null; `shift` / `pre_bias`: the stats kernel subtracted K = shift - pre_bias (null = 0).
// pre_bias is the bias of the convolution that produced x, folded in here instead of being added to x: BN(x + pb)
// equals BN(x) with the batch mean moved by pb, so only the running mean sees it.
extern "C" __global__ void __launch_bounds__(KSH_TW_THREADS)
ksh_bn_finalize_kernel(const float *__restrict__ partial, KshRowSplit sp, const float *shift /* may alias running_mean */,
                       const float *pre_bias, const float *__restrict__ w, const float *__restrict__ b, float *running_mean, float *running_var,
                       float momentum, float eps, float *__restrict__ save_mean, float *__restrict__ save_invstd,
                       float *__restrict__ coef /*[3][C]*/) {
    int c;
    double s, q;
    if (!ksh_tw_sum_partials(partial, sp, c, s, q)) return;
    const double n = (double)sp.M;
    const double dm = s / n;                          // mean of (x - K)
    double m2 = q - s * dm;                           // sum of squared deviations from the mean
    if (m2 < 0.0) m2 = 0.0;
    const float pb = pre_bias != nullptr ? pre_bias[c] : 0.0f;
    const double mean = dm + (shift != nullptr ? (double)(shift[c] - pb) : 0.0);     // batch mean of x itself
    const double var = m2 / n;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    save_mean[c] = fmean;
    save_invstd[c] = invstd;
    if (running_mean != nullptr) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (fmean + pb);
    if (running_var != nullptr) {
        const float unbiased = (float)(n > 1.0 ? m2 / (n - 1.0) : var);
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
    coef[c] = (w != nullptr ? w[c] : 1.0f) * invstd;
    coef[sp.C + c] = fmean;
    coef[2 * sp.C + c] = b != nullptr ? b[c] : 0.0f;
}

template <typename T, int RELU, int HAS_RES>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_bn_apply_kernel(const T *__restrict__ x, const T *__restrict__ res,
                                                                      T *__restrict__ y, KshRowSplit sp,
                                                                      const float *__restrict__ coef /*[3][C]*/) {
    const int cg = sp.C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const long long r0 = (long long)blockIdx.x * sp.R;
    const long long r1 = r0 + sp.R < sp.M ? r0 + sp.R : sp.M;
    float sc[8], mu[8], bi[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        sc[k] = coef[tcg * 8 + k];
        mu[k] = coef[sp.C + tcg * 8 + k];
        bi[k] = coef[2 * sp.C + tcg * 8 + k];
    }
#pragma unroll 4
    for (long long r = r0 + tr; r < r1; r += rpp) {
        const size_t off = (size_t)r * sp.C + tcg * 8;
        float v[8], o[8];
        KshVec8<T>::load(x + off, v);
        if (HAS_RES) KshVec8<T>::load(res + off, o);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float t = (v[k] - mu[k]) * sc[k] + bi[k];   // centred first: no cancellation when |mean| >> std
            if (HAS_RES) t += o[k];
            if (RELU) t = t > 0.0f ? t : 0.0f;
            v[k] = t;
        }
        KshVec8<T>::store(y + off, v);
    }
}

// ------------------------------------------------------------------------------------------ BN backward
template <typename T, int RELU>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_bn_bwd_reduce_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                                           const T *__restrict__ y, KshRowSplit sp,
                                                                           const float *__restrict__ mean,
                                                                           float *__restrict__ partial /*[G][2][C]*/) {
    __shared__ float red[KSH_TW_THREADS * 16];
    const int cg = sp.C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const long long r0 = (long long)blockIdx.x * sp.R;
    const long long r1 = r0 + sp.R < sp.M ? r0 + sp.R : sp.M;
    float mu[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        mu[k] = mean[tcg * 8 + k];
        s1[k] = s2[k] = 0.0f;
    }
#pragma unroll 2
    for (long long r = r0 + tr; r < r1; r += rpp) {
        const size_t off = (size_t)r * sp.C + tcg * 8;
        float g[8], v[8], o[8];
        KshVec8<T>::load(dy + off, g);
        KshVec8<T>::load(x + off, v);
        if (RELU) KshVec8<T>::load(y + off, o);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float dz = (RELU && !(o[k] > 0.0f)) ? 0.0f : g[k];
            s1[k] += dz;
            s2[k] += dz * (v[k] - mu[k]);
        }
    }
    ksh_tw_reduce16(red, s1, s2, cg, tcg, tr);
    if (tr == 0) {
        float *o = partial + (size_t)blockIdx.x * 2 * sp.C + tcg * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            o[k] = s1[k];
            o[sp.C + k] = s2[k];
        }
    }
}

// launch with ceil(C / 16) blocks of 256 threads.  coef: [0] a = w * invstd, [1] k1 = sum(dz) / M,
// [2] k2 = invstd^2 * sum(dz * (x - mean)) / M;  dw / db may be null
extern "C" __global__ void __launch_bounds__(KSH_TW_THREADS)
ksh_bn_bwd_finalize_kernel(const float *__restrict__ partial, KshRowSplit sp, const float *__restrict__ w,
                           const float *__restrict__ invstd, float *__restrict__ dw, float *__restrict__ db,
                           float *__restrict__ coef /*[3][C]*/) {
    int c;
    double s1, s2;
    if (!ksh_tw_sum_partials(partial, sp, c, s1, s2)) return;
    const double is = (double)invstd[c];
    if (dw != nullptr) dw[c] = (float)(s2 * is);
    if (db != nullptr) db[c] = (float)s1;
    coef[c] = (w != nullptr ? w[c] : 1.0f) * invstd[c];
    coef[sp.C + c] = (float)(s1 / (double)sp.M);
    coef[2 * sp.C + c] = (float)(s2 * is * is / (double)sp.M);
}

template <typename T, int RELU, int HAS_RES>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_bn_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ x,
                                                                          const T *__restrict__ y, T *__restrict__ dx,
                                                                          T *__restrict__ dres, KshRowSplit sp,
                                                                          const float *__restrict__ mean,
                                                                          const float *__restrict__ coef /*[3][C]*/) {
    const int cg = sp.C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const long long r0 = (long long)blockIdx.x * sp.R;
    const long long r1 = r0 + sp.R < sp.M ? r0 + sp.R : sp.M;
    float mu[8], a[8], k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        mu[k] = mean[tcg * 8 + k];
        a[k] = coef[tcg * 8 + k];
        k1[k] = coef[sp.C + tcg * 8 + k];
        k2[k] = coef[2 * sp.C + tcg * 8 + k];
    }
#pragma unroll 2
    for (long long r = r0 + tr; r < r1; r += rpp) {
        const size_t off = (size_t)r * sp.C + tcg * 8;
        float g[8], v[8], o[8];
        KshVec8<T>::load(dy + off, g);
        KshVec8<T>::load(x + off, v);
        if (RELU) KshVec8<T>::load(y + off, o);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float dz = (RELU && !(o[k] > 0.0f)) ? 0.0f : g[k];
            g[k] = dz;
            v[k] = (dz - k1[k] - (v[k] - mu[k]) * k2[k]) * a[k];
        }
        KshVec8<T>::store(dx + off, v);
        if (HAS_RES) KshVec8<T>::store(dres + off, g);
    }
}

// ------------------------------------------------------------------------------------------ squeeze-excitation tail
// x [N][HW][C] (channels-last), one block per sample: p[n][c] = mean over the HW positions (SqueezeExcitation's
// adaptive_avg_pool2d, resnet_tower.py:22)
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_se_pool_kernel(const T *__restrict__ x, int HW, int C,
                                                                     float *__restrict__ p /*[N][C]*/) {
    __shared__ float red[KSH_TW_THREADS * 16];
    const int cg = C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const T *xs = x + (size_t)blockIdx.x * HW * C;
    float s[8], unused[8];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = unused[k] = 0.0f;
#pragma unroll 4
    for (int r = tr; r < HW; r += rpp) {
        float v[8];
        KshVec8<T>::load(xs + (size_t)r * C + tcg * 8, v);
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] += v[k];
    }
    ksh_tw_reduce16(red, s, unused, cg, tcg, tr);
    if (tr == 0) {
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int k = 0; k < 8; k++) p[(size_t)blockIdx.x * C + tcg * 8 + k] = s[k] * inv;
    }
}

// z = relu(x * s[n][c] + r)   (x * s: resnet_tower.py:25; out += x; relu: :44-45)
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_se_apply_kernel(const T *__restrict__ x, const float *__restrict__ s,
                                                                      const T *__restrict__ r, T *__restrict__ z, int HW,
                                                                      int C) {
    const int cg = C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const size_t base = (size_t)blockIdx.x * HW * C;
    float sc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) sc[k] = s[(size_t)blockIdx.x * C + tcg * 8 + k];
#pragma unroll 4
    for (int row = tr; row < HW; row += rpp) {
        const size_t off = base + (size_t)row * C + tcg * 8;
        float v[8], o[8];
        KshVec8<T>::load(x + off, v);
        KshVec8<T>::load(r + off, o);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float t = v[k] * sc[k] + o[k];
            v[k] = t > 0.0f ? t : 0.0f;
        }
        KshVec8<T>::store(z + off, v);
    }
}

// ds[n][c] = sum over HW of dz * x, dz = dy * (z > 0)
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_se_bwd_reduce_kernel(const T *__restrict__ dy, const T *__restrict__ z,
                                                                           const T *__restrict__ x, int HW, int C,
                                                                           float *__restrict__ ds /*[N][C]*/) {
    __shared__ float red[KSH_TW_THREADS * 16];
    const int cg = C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const size_t base = (size_t)blockIdx.x * HW * C;
    float acc[8], unused[8];
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = unused[k] = 0.0f;
#pragma unroll 2
    for (int row = tr; row < HW; row += rpp) {
        const size_t off = base + (size_t)row * C + tcg * 8;
        float g[8], o[8], v[8];
        KshVec8<T>::load(dy + off, g);
        KshVec8<T>::load(z + off, o);
        KshVec8<T>::load(x + off, v);
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += (o[k] > 0.0f) ? g[k] * v[k] : 0.0f;
    }
    ksh_tw_reduce16(red, acc, unused, cg, tcg, tr);
    if (tr == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) ds[(size_t)blockIdx.x * C + tcg * 8 + k] = acc[k];
    }
}

// dx = dz * s + dp / HW (dp = gradient reaching the pooled mean), dr = dz
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_se_bwd_apply_kernel(const T *__restrict__ dy, const T *__restrict__ z,
                                                                          const float *__restrict__ s,
                                                                          const float *__restrict__ dp, T *__restrict__ dx,
                                                                          T *__restrict__ dr, int HW, int C) {
    const int cg = C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const size_t base = (size_t)blockIdx.x * HW * C;
    float sc[8], add[8];
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        sc[k] = s[(size_t)blockIdx.x * C + tcg * 8 + k];
        add[k] = dp[(size_t)blockIdx.x * C + tcg * 8 + k] * inv;
    }
#pragma unroll 2
    for (int row = tr; row < HW; row += rpp) {
        const size_t off = base + (size_t)row * C + tcg * 8;
        float g[8], o[8];
        KshVec8<T>::load(dy + off, g);
        KshVec8<T>::load(z + off, o);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float dz = (o[k] > 0.0f) ? g[k] : 0.0f;
            g[k] = dz;
            o[k] = dz * sc[k] + add[k];
        }
        KshVec8<T>::store(dx + off, o);
        KshVec8<T>::store(dr + off, g);
    }
}

// ------------------------------------------------------------------------------------------ bn2 + SE tail, merged passes
// In a residual block with squeeze-excitation the second batch norm feeds the SE tail directly, so two passes can ride
// on their neighbours: the SE pooling on the batch-norm apply pass (forward), and the batch-norm backward reduction on
// the SE backward apply pass (which produces the very tensor that reduction would read).

// y = (x - mean) * scale + bias (no ReLU, no residual) AND pooled[n][c] = mean over HW of the stored y; one block per sample
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS) ksh_bn_apply_pool_kernel(const T *__restrict__ x, T *__restrict__ y, int HW,
                                                                           int C, const float *__restrict__ coef /*[3][C]*/,
                                                                           float *__restrict__ pooled /*[N][C]*/) {
    __shared__ float red[KSH_TW_THREADS * 16];
    const int cg = C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const size_t base = (size_t)blockIdx.x * HW * C;
    float sc[8], mu[8], bi[8], acc[8], unused[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        sc[k] = coef[tcg * 8 + k];
        mu[k] = coef[C + tcg * 8 + k];
        bi[k] = coef[2 * C + tcg * 8 + k];
        acc[k] = unused[k] = 0.0f;
    }
#pragma unroll 4
    for (int row = tr; row < HW; row += rpp) {
        const size_t off = base + (size_t)row * C + tcg * 8;
        float v[8];
        KshVec8<T>::load(x + off, v);
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (v[k] - mu[k]) * sc[k] + bi[k];
        KshVec8<T>::round(v);                  // the pooled mean is taken over the values as stored (bf16-rounded)
        KshVec8<T>::store(y + off, v);
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += v[k];
    }
    ksh_tw_reduce16(red, acc, unused, cg, tcg, tr);
    if (tr == 0) {
        const float inv = 1.0f / (float)HW;
#pragma unroll
        for (int k = 0; k < 8; k++) pooled[(size_t)blockIdx.x * C + tcg * 8 + k] = acc[k] * inv;
    }
}

// SE backward apply (dy2 = dz * (z > 0) * s + dp / HW -> the gradient of the batch norm's output; dr = dz * (z > 0)) that
// also leaves the batch-norm backward partial sums of dy2 and dy2 * (xc - mean), xc = the batch norm's input.  One block
// per `spb` samples; partial row b matches KshRowSplit{M = N * HW, R = spb * HW}.
template <typename T>
__global__ void __launch_bounds__(KSH_TW_THREADS)
ksh_se_bwd_apply_bnred_kernel(const T *__restrict__ dz, const T *__restrict__ z, const float *__restrict__ s,
                              const float *__restrict__ dp, const T *__restrict__ xc, const float *__restrict__ mean,
                              T *__restrict__ dy2, T *__restrict__ dr, int N, int HW, int C, int spb,
                              float *__restrict__ partial /*[G][2][C]*/) {
    __shared__ float red[KSH_TW_THREADS * 16];
    const int cg = C >> 3, tcg = (int)threadIdx.x % cg, tr = (int)threadIdx.x / cg, rpp = KSH_TW_THREADS / cg;
    const float inv = 1.0f / (float)HW;
    float mu[8], a1[8], a2[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        mu[k] = mean[tcg * 8 + k];
        a1[k] = a2[k] = 0.0f;
    }
    const int n0 = (int)blockIdx.x * spb, n1 = n0 + spb < N ? n0 + spb : N;
    for (int n = n0; n < n1; n++) {
        const size_t base = (size_t)n * HW * C;
        float sc[8], add[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            sc[k] = s[(size_t)n * C + tcg * 8 + k];
            add[k] = dp[(size_t)n * C + tcg * 8 + k] * inv;
        }
#pragma unroll 2
        for (int row = tr; row < HW; row += rpp) {
            const size_t off = base + (size_t)row * C + tcg * 8;
            float g[8], o[8], v[8];
            KshVec8<T>::load(dz + off, g);
            KshVec8<T>::load(z + off, o);
            KshVec8<T>::load(xc + off, v);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float dpre = (o[k] > 0.0f) ? g[k] : 0.0f;
                g[k] = dpre;
                o[k] = dpre * sc[k] + add[k];
            }
            KshVec8<T>::round(o);              // sums over the gradient as stored, like a separate reduction pass would see it
            KshVec8<T>::store(dr + off, g);
            KshVec8<T>::store(dy2 + off, o);
#pragma unroll
            for (int k = 0; k < 8; k++) {
                a1[k] += o[k];
                a2[k] += o[k] * (v[k] - mu[k]);
            }
        }
    }
    ksh_tw_reduce16(red, a1, a2, cg, tcg, tr);
    if (tr == 0) {
        float *o = partial + (size_t)blockIdx.x * 2 * C + tcg * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            o[k] = a1[k];
            o[C + k] = a2[k];
        }
    }
}
