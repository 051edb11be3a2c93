// ksh_ppo_kernels.h -- gfx950 kernels for the learner side of the Keisei hot path:
//   ksh_gae_kernel            ExperienceBuffer.compute_advantages_and_returns (core/experience_buffer.py:99-145)
//   ksh_masked_sample_kernel  BaseActorCriticModel.get_action_and_value tail (core/base_actor_critic.py:64-116)
//   ksh_ppo_loss_kernel (+ ksh_ppo_stats_kernel)
//                             evaluate_actions tail (core/base_actor_critic.py:147-184) + the loss lines of
//                             PPOAgent.learn (core/ppo_agent.py:340-383), forward AND backward
//
// Row kernels use one 256-thread block per sample row of 13 527 logits.  The legal mask (1 byte per
// action, ~0.4 % dense in real positions) is read ONCE with aligned 16-byte loads and compressed to a
// 1 692-byte bit-set in LDS; the logits are then visited in 16-byte groups and a group is only
// fetched when its legality bits are non-zero, so illegal logits never leave HBM.  The same code is
// correct for dense masks (the reference's tests use all-ones masks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ksh_device_util.h"

#define KSH_PA 13527
#define KSH_ROW_THREADS 256
#define KSH_BITW 432
#define KSH_PBITROW 424   // dwords per row of a packed mask (matches KSH_BITROW of the env kernel)


// ------------------------------------------------------------------------------------------ GAE
// One thread per stream n, sequential in t, coalesced across n.  Every binary op is individually
// rounded to float32 (no FMA contraction) in the reference's order, so results are bit-exact.
extern "C" __global__ void ksh_gae_kernel(const float *__restrict__ r, const float *__restrict__ v,
                                          const uint8_t *__restrict__ d, const float *__restrict__ last_v, float g,
                                          float gl, int T, int N, float *__restrict__ adv, float *__restrict__ ret) {
    // hipcc contracts a*b+c into one FMA by default; the reference rounds after every operation.
#pragma clang fp contract(off)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float gae = 0.0f;
    float next_v = last_v[n];
    // The recursion is sequential in t, but its INPUTS are not: the loads of KSH_GAE_CHUNK steps are issued together
    // (one memory round trip per chunk instead of one per step: 32 steps took 18 us as a chain of dependent loads).
    // The arithmetic and its order are untouched.
    constexpr int CH = 16;
    for (int t0 = T - 1; t0 >= 0; t0 -= CH) {
        float rv[CH], vv[CH];
        uint8_t dv[CH];
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int t = t0 - k;
            const size_t i = (size_t)(t >= 0 ? t : 0) * (size_t)N + (size_t)n;
            rv[k] = r[i];
            vv[k] = v[i];
            dv[k] = d[i];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) {
            const int t = t0 - k;
            if (t < 0) continue;
            const size_t i = (size_t)t * (size_t)N + (size_t)n;
            const float vi = vv[k];
            const float m = dv[k] ? 0.0f : 1.0f;                                 // 1.0 - done
            // plain operators under the pragma: the __f*_rn wrappers are inlined with their own
            // (contractable) flags and would be fused back into v_fmac_f32
            const float a1 = g * next_v;
            const float a2 = a1 * m;
            const float a3 = rv[k] + a2;
            const float delta = a3 - vi;
            const float b1 = gl * m;
            const float b2 = b1 * gae;
            gae = delta + b2;
            adv[i] = gae;
            ret[i] = gae + vi;
            next_v = vi;
        }
    }
}

struct KshRowLds {
    uint32_t bits[KSH_BITW];
    float rf[KSH_ROW_THREADS / 64][4];
    int ri[KSH_ROW_THREADS / 64][2];
    float bc[8];
    int bci[4];
};

// Compress the mask row (1 byte per action) into R.bits; returns nothing, ends with a barrier.
__device__ __forceinline__ void ksh_build_bits(KshRowLds &R, const uint8_t *__restrict__ mrow, int tid) {
    for (int i = tid; i < KSH_BITW; i += KSH_ROW_THREADS) R.bits[i] = 0;
    __syncthreads();
    const size_t addr = reinterpret_cast<size_t>(mrow);
    const int head = (int)((16 - (addr & 15)) & 15);
    const int nfull = (KSH_PA - head) >> 4;
    const int tail0 = head + 16 * nfull;
    if (tid < head && mrow[tid]) atomicOr(&R.bits[0], 1u << tid);
    for (int c = tid; c < nfull; c += KSH_ROW_THREADS) {
        const int b0 = head + 16 * c;
        const uint4 v = *reinterpret_cast<const uint4 *>(mrow + b0);
        // byte != 0 -> 1 bit (torch.bool bytes are 0/1; be safe for any non-zero byte)
        uint32_t x = 0;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t nz = w[k] | (w[k] >> 4);
            nz |= nz >> 2;
            nz |= nz >> 1;
            nz &= 0x01010101u;                                   // bit0 of each byte = (byte != 0)
            x |= (((nz * 0x01020408u) >> 24) & 0xFu) << (4 * k); // gather the 4 bits
        }
        if (x) {
            const int wd = b0 >> 5, sh = b0 & 31;
            atomicOr(&R.bits[wd], x << sh);
            if (sh > 16) atomicOr(&R.bits[wd + 1], x >> (32 - sh));
        }
    }
    if (tid < KSH_PA - tail0) {
        const int b = tail0 + tid;
        if (mrow[b]) atomicOr(&R.bits[b >> 5], 1u << (b & 31));
    }
    __syncthreads();
}

// Packed-mask variant: the row already is the bit-set (424 dwords, 16-byte aligned rows).
__device__ __forceinline__ void ksh_load_bits(KshRowLds &R, const uint32_t *__restrict__ brow, int tid) {
    for (int i = tid; i < KSH_BITW; i += KSH_ROW_THREADS) R.bits[i] = i < KSH_PBITROW ? brow[i] : 0u;
    __syncthreads();
}

template <typename T> struct KshVec;
template <> struct KshVec<float> { static constexpr int N = 4; };
template <> struct KshVec<ksh_bf16> { static constexpr int N = 8; };

__device__ __forceinline__ uint32_t ksh_bits_at(const uint32_t *bits, int e0) {
    return __funnelshift_r(bits[e0 >> 5], bits[(e0 >> 5) + 1], e0 & 31);
}

// Visit every legal element j of the row with its logit z (as float): f(j, z).  Groups of 16 bytes whose
// legality bits are all zero are skipped without touching global memory.
template <typename T, typename F>
__device__ __forceinline__ void ksh_for_each_legal(const T *__restrict__ row, const uint32_t *bits, int tid, F f) {
    constexpr int V = KshVec<T>::N;
    const size_t addr = reinterpret_cast<size_t>(row);
    int head = (int)(((16 - (addr & 15)) & 15) / sizeof(T));
    const int nvec = (KSH_PA - head) / V;
    const int tail0 = head + nvec * V;
    if (tid < head && ((bits[0] >> tid) & 1u)) f(tid, ksh_ld(row + tid));
    for (int v = tid; v < nvec; v += KSH_ROW_THREADS) {
        const int e0 = head + v * V;
        uint32_t x = ksh_bits_at(bits, e0) & ((1u << V) - 1u);
        if (x) {
            const uint4 q = *reinterpret_cast<const uint4 *>(row + e0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < V; k++) {
                if ((x >> k) & 1u) {
                    float z;
                    if (V == 4) z = ksh_u2f(w[k]);
                    else z = ksh_u2f((k & 1) ? (w[k >> 1] & 0xFFFF0000u) : (w[k >> 1] << 16));
                    f(e0 + k, z);
                }
            }
        }
    }
    if (tid < KSH_PA - tail0) {
        const int j = tail0 + tid;
        if ((bits[j >> 5] >> (j & 31)) & 1u) f(j, ksh_ld(row + j));
    }
}

// Deterministic block reductions (fixed butterfly order inside a wave, fixed order across the 4 waves).
__device__ __forceinline__ float ksh_block_sum(KshRowLds &R, float v, int tid) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) R.rf[tid >> 6][0] = v;
    __syncthreads();
    const float s = ((R.rf[0][0] + R.rf[1][0]) + R.rf[2][0]) + R.rf[3][0];
    __syncthreads();
    return s;
}
// (max, sum-of-exp) pairs of a streaming soft-max
__device__ __forceinline__ void ksh_ms_merge(float &m, float &s, float m2, float s2) {
    const float M = fmaxf(m, m2);
    if (M == -INFINITY) { m = M; s = 0.0f; return; }
    s = s * expf(m - M) + s2 * expf(m2 - M);
    m = M;
}
__device__ __forceinline__ void ksh_block_ms(KshRowLds &R, float &m, float &s, int tid) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        ksh_ms_merge(m, s, m2, s2);
    }
    if ((tid & 63) == 0) { R.rf[tid >> 6][0] = m; R.rf[tid >> 6][1] = s; }
    __syncthreads();
    m = R.rf[0][0]; s = R.rf[0][1];
    ksh_ms_merge(m, s, R.rf[1][0], R.rf[1][1]);
    ksh_ms_merge(m, s, R.rf[2][0], R.rf[2][1]);
    ksh_ms_merge(m, s, R.rf[3][0], R.rf[3][1]);
    __syncthreads();
}
// arg-max of (key, lower index wins ties), carrying the logit of the winner
__device__ __forceinline__ void ksh_block_argmax(KshRowLds &R, float &key, int &idx, float &z, int tid) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float k2 = __shfl_xor(key, o);
        const int i2 = __shfl_xor(idx, o);
        const float z2 = __shfl_xor(z, o);
        if (i2 >= 0 && (idx < 0 || k2 > key || (k2 == key && i2 < idx))) { key = k2; idx = i2; z = z2; }
    }
    if ((tid & 63) == 0) { R.rf[tid >> 6][2] = key; R.rf[tid >> 6][3] = z; R.ri[tid >> 6][0] = idx; }
    __syncthreads();
    key = R.rf[0][2]; z = R.rf[0][3]; idx = R.ri[0][0];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        const float k2 = R.rf[w][2], z2 = R.rf[w][3];
        const int i2 = R.ri[w][0];
        if (i2 >= 0 && (idx < 0 || k2 > key || (k2 == key && i2 < idx))) { key = k2; idx = i2; z = z2; }
    }
    __syncthreads();
}

// Philox4x32-10 (Salmon et al. 2011), counter-based: one independent 128-bit block per (row, action).
__device__ __forceinline__ uint32_t ksh_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}

// uniform in the OPEN interval (0,1) from 32 random bits
__device__ __forceinline__ float ksh_uniform_open(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// Categorical's eps for float32 (torch.distributions.utils.clamp_probs: finfo(float32).eps)
#define KSH_EPS 1.1920928955078125e-07f

// ------------------------------------------------------------------------------------------ sampler
// Exact categorical sampling by the Gumbel-max trick: argmax_j (z_j + G_j) over legal j with
// G_j = -log(-log u_j), u_j from Philox keyed by (seed; action, row, step) -- order independent, so the
// result is a pure function of (seed, step, row, logits, mask).  log_prob follows
// Categorical(probs=softmax(masked logits)).log_prob: log(clamp(p, eps, 1-eps)).
//
// One WAVE per row (4 rows per block, no block barrier): each lane streams 14 aligned 16-byte chunks of the
// mask row (all loads in flight), and only for non-zero bytes gathers the logit -- a real position has ~50
// legal actions out of 13 527, so the logits of illegal actions are never read.
#define KSH_SAMPLE_WPB 4
#define KSH_SAMPLE_CAP 1024

struct KshSampleAcc {
    float m, s, key, zbest;
    int idx;
};

template <typename T>
__device__ __forceinline__ void ksh_sample_visit(KshSampleAcc &A, const T *row, int j, int b, bool deterministic,
                                                 uint32_t s0, uint32_t s1, uint32_t k0, uint32_t k1) {
    const float z = ksh_ld(row + j);
    ksh_ms_merge(A.m, A.s, z, 1.0f);
    float k = z;
    if (!deterministic) {
        const uint32_t x = ksh_philox((uint32_t)j, (uint32_t)b, s0, s1, k0, k1);
        // 23 random bits + 0.5: every value k + 0.5, k < 2^23, is exact in float32, so u lies in [2^-24, 1 - 2^-24] --
        // strictly inside (0,1).  (24 bits + 0.5 would round 16777215.5 up to 2^24, i.e. u == 1, a key of +inf.)
        const float u = ksh_uniform_open(x);
        k = z - logf(-logf(u));
    }
    if (A.idx < 0 || k > A.key || (k == A.key && j < A.idx)) { A.key = k; A.idx = j; A.zbest = z; }
}

template <typename T>
__global__ void __launch_bounds__(64 * KSH_SAMPLE_WPB)
ksh_masked_sample_kernel(const T *__restrict__ logits, const uint8_t *__restrict__ mask,
                         const uint32_t *__restrict__ mask_bits, int B, unsigned long long seed,
                         unsigned long long step, const unsigned long long *__restrict__ step_dev, int deterministic,
                         long long *__restrict__ action, float *__restrict__ log_prob) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * KSH_SAMPLE_WPB + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (b >= B) return;  // wave-uniform
    const uint8_t *mrow = mask + (mask_bits != nullptr ? (size_t)0 : (size_t)b * KSH_PA);
    const T *row = logits + (size_t)b * KSH_PA;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    // the Philox counter of this call: an immediate, plus (graph-captured rollouts) a counter that lives on the device
    if (step_dev != nullptr) step += step_dev[0];
    const uint32_t s0 = (uint32_t)step, s1 = (uint32_t)(step >> 32);
    const bool det = deterministic != 0;
    KshSampleAcc A = {-INFINITY, 0.0f, -INFINITY, 0.0f, -1};
    // Legal ids are first collected in a per-wave LDS list and then dealt round-robin to the lanes: drop ids
    // of one square are 7 consecutive actions, so without this a single lane would gather up to 16 logits one
    // after the other.  Ids beyond the list capacity (dense masks) are visited directly by the lane that found them.
    __shared__ unsigned short s_list[KSH_SAMPLE_WPB][KSH_SAMPLE_CAP];
    __shared__ unsigned int s_cnt[KSH_SAMPLE_WPB];
    const int wv = (int)(threadIdx.x >> 6);
    if (lane == 0) s_cnt[wv] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto push = [&](int j) {
        const unsigned slot = atomicAdd(&s_cnt[wv], 1u);
        if (slot < KSH_SAMPLE_CAP) s_list[wv][slot] = (unsigned short)j;
        else ksh_sample_visit<T>(A, row, j, b, det, s0, s1, k0, k1);
    };

    if (mask_bits != nullptr) {
        // packed rows: 424 dwords per row, 7 per lane; every set bit is a legal id
        const uint32_t *brow = mask_bits + (size_t)b * KSH_PBITROW;
        uint32_t wd[7];
#pragma unroll
        for (int it = 0; it < 7; it++) {
            const int i = lane + 64 * it;
            wd[it] = i < KSH_PBITROW ? brow[i] : 0u;
        }
#pragma unroll
        for (int it = 0; it < 7; it++) {
            uint32_t v = wd[it];
            const int j0 = 32 * (lane + 64 * it);
            while (v) {
                const int bit = __ffs((int)v) - 1;
                v &= v - 1u;
                if (j0 + bit < KSH_PA) push(j0 + bit);
            }
        }
    } else {
        const size_t addr = reinterpret_cast<size_t>(mrow);
        const int head = (int)((16 - (addr & 15)) & 15);
        const int nfull = (KSH_PA - head) >> 4;               // 844 or 845 aligned chunks
        const int tail0 = head + 16 * nfull;
        if (lane < head && mrow[lane]) push(lane);
        if (lane < KSH_PA - tail0 && mrow[tail0 + lane]) push(tail0 + lane);
        for (int half = 0; half < 2; half++) {                 // 2 x 7 x 64 = 896 >= 845 chunks, 7 loads in flight per lane
            uint4 ch[7];
    #pragma unroll
            for (int it = 0; it < 7; it++) {
                const int c = lane + 64 * (7 * half + it);
                ch[it] = uint4{0u, 0u, 0u, 0u};
                if (c < nfull) ch[it] = *reinterpret_cast<const uint4 *>(mrow + head + 16 * c);
            }
    #pragma unroll
            for (int it = 0; it < 7; it++) {
                const uint32_t any = ch[it].x | ch[it].y | ch[it].z | ch[it].w;
                if (any != 0u) {
                    const int j0 = head + 16 * (lane + 64 * (7 * half + it));
    #pragma unroll
                    for (int q = 0; q < 4; q++) {
                        uint32_t v = q == 0 ? ch[it].x : (q == 1 ? ch[it].y : (q == 2 ? ch[it].z : ch[it].w));
                        while (v) {                             // visit the non-zero bytes of this dword
                            const int byte = (__ffs((int)v) - 1) >> 3;
                            v &= ~(0xFFu << (8 * byte));
                            push(j0 + 4 * q + byte);
                        }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const unsigned total = s_cnt[wv];
        const int n = (int)(total < KSH_SAMPLE_CAP ? total : KSH_SAMPLE_CAP);
        for (int e = lane; e < n; e += 64) ksh_sample_visit<T>(A, row, (int)s_list[wv][e], b, det, s0, s1, k0, k1);
    }
    // wave reductions (xor butterfly): soft-max statistics and the arg-max with its logit
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(A.m, o), s2 = __shfl_xor(A.s, o);
        ksh_ms_merge(A.m, A.s, m2, s2);
        const float k2 = __shfl_xor(A.key, o), z2 = __shfl_xor(A.zbest, o);
        const int i2 = __shfl_xor(A.idx, o);
        if (i2 >= 0 && (A.idx < 0 || k2 > A.key || (k2 == A.key && i2 < A.idx))) { A.key = k2; A.idx = i2; A.zbest = z2; }
    }
    if (lane == 0) {
        long long a;
        float lp;
        if (A.idx < 0) {
            // no legal action: the reference falls back to a uniform distribution over ALL actions
            // (base_actor_critic.py:93-101)
            const uint32_t x = ksh_philox(0xFFFFFFFFu, (uint32_t)b, s0, s1, k0, k1);
            a = det ? 0 : (long long)(((unsigned long long)x * (unsigned long long)KSH_PA) >> 32);
            lp = logf(1.0f / (float)KSH_PA);
        } else {
            a = A.idx;
            float p = expf(A.zbest - A.m) / A.s;
            p = fminf(fmaxf(p, KSH_EPS), 1.0f - KSH_EPS);
            lp = logf(p);
        }
        action[b] = a;
        log_prob[b] = lp;
    }
}

// ------------------------------------------------------------------------------------------ PPO loss
struct KshLossArgs {
    const uint8_t *mask;         // one byte per action ...
    const uint32_t *mask_bits;   // ... or packed rows (exactly one of the two is non-null)
    const long long *actions;
    const float *old_logp;
    const float *adv;
    const float *ret;
    const float *values;
    const float *old_values;
    int B;
    float clip_eps, vf_coef, ent_coef;
    int value_clip;
    float *dvalues;
    float *stats;
    float *partial;  // [B][8]: surrogate, (V-R)^2, (Vc-R)^2, H, old-new logp, clipped flag, overflow flag, -
    int only_overflow;  // dense kernel: process only the rows the sparse kernel flagged (partial[b][6] != 0)
    float *row_logp;     // [B] or null: log-prob of actions[b] under the masked soft-max (evaluate_actions' first output)
    float *row_entropy;  // [B] or null: entropy of the row's distribution (evaluate_actions' second output)
    int overflow_is_error;  // stats kernel: a row still flagged in partial[b][6] (dense row, dense path switched off) poisons the outputs with NaN
};

template <typename T>
__global__ void __launch_bounds__(KSH_ROW_THREADS)
ksh_ppo_loss_kernel(KshLossArgs a, const T *__restrict__ logits, T *__restrict__ dlogits) {
    __shared__ KshRowLds R;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    if (a.only_overflow && a.partial[(size_t)b * 8 + 6] == 0.0f) return;   // block-uniform: the sparse kernel did this row
    if (a.mask_bits != nullptr) ksh_load_bits(R, a.mask_bits + (size_t)b * KSH_PBITROW, tid);
    else ksh_build_bits(R, a.mask + (size_t)b * KSH_PA, tid);
    const T *row = logits + (size_t)b * KSH_PA;
    T *drow = dlogits + (size_t)b * KSH_PA;
    const int act = (int)a.actions[b];
    const float invB = 1.0f / (float)a.B;

    // pass 1: soft-max statistics over the legal logits
    float m = -INFINITY, s = 0.0f;
    ksh_for_each_legal<T>(row, R.bits, tid, [&](int, float z) { ksh_ms_merge(m, s, z, 1.0f); });
    ksh_block_ms(R, m, s, tid);
    const bool none = !(s > 0.0f);  // no legal action in this row

    // pass 2: E1 = sum y*l, E2 = sum y*u, and the taken action's probability
    float e1 = 0.0f, e2 = 0.0f, ya = 0.0f;
    const float inv_s = none ? 0.0f : 1.0f / s;
    ksh_for_each_legal<T>(row, R.bits, tid, [&](int j, float z) {
        const float y = expf(z - m) * inv_s;
        const bool u = (y >= KSH_EPS) && (y <= 1.0f - KSH_EPS);
        const float l = logf(fminf(fmaxf(y, KSH_EPS), 1.0f - KSH_EPS));
        e1 += y * l;
        e2 += u ? y : 0.0f;
        if (j == act) ya = y;
    });
    e1 = ksh_block_sum(R, e1, tid);
    e2 = ksh_block_sum(R, e2, tid);
    ya = ksh_block_sum(R, ya, tid);  // exactly one thread holds a non-zero value

    // per-row scalars (every thread computes the same values)
    float new_logp, H;
    bool ua;
    if (none) {
        // NaN row -> uniform over all actions, detached from the logits (base_actor_critic.py:168-174)
        new_logp = logf(1.0f / (float)KSH_PA);
        H = -new_logp;
        ua = false;
        ya = 1.0f;
    } else {
        ua = (ya >= KSH_EPS) && (ya <= 1.0f - KSH_EPS);
        new_logp = logf(fminf(fmaxf(ya, KSH_EPS), 1.0f - KSH_EPS));
        H = -e1;
    }
    const float old_lp = a.old_logp[b], A = a.adv[b];
    const float ratio = expf(new_logp - old_lp);
    const float surr1 = ratio * A;
    const float surr2 = fminf(fmaxf(ratio, 1.0f - a.clip_eps), 1.0f + a.clip_eps) * A;
    const float gr = (surr1 <= surr2) ? A : 0.0f;                 // d min(surr1,surr2) / d ratio
    const float pol = (ua && !none) ? (-gr * ratio * invB) : 0.0f; // dL/dnew_logp when the clamp passes gradient
    const float ce = a.ent_coef * invB;
    const float c = ce * (e1 + e2) + pol;                         // sum_k y_k G_k

    if (tid == 0) {
        const float V = a.values[b], Rt = a.ret[b];
        float *p = a.partial + (size_t)b * 8;
        p[0] = fminf(surr1, surr2);
        p[1] = (V - Rt) * (V - Rt);
        float vc = 0.0f;
        if (a.value_clip) {
            const float ov = a.old_values[b];
            const float Vc = ov + fminf(fmaxf(V - ov, -a.clip_eps), a.clip_eps);
            vc = (Vc - Rt) * (Vc - Rt);
        }
        p[2] = vc;
        p[3] = H;
        p[4] = old_lp - new_logp;
        p[5] = (fabsf(ratio - 1.0f) > a.clip_eps) ? 1.0f : 0.0f;
        if (!a.only_overflow) p[6] = 0.0f;   // a flagged row stays flagged: the fused head's backward GEMMs gate on it
        p[7] = 0.0f;
        if (a.row_logp != nullptr) a.row_logp[b] = new_logp;
        if (a.row_entropy != nullptr) a.row_entropy[b] = H;
    }
    if (dlogits == nullptr) return;   // evaluate-only call (ksh_evaluate_actions): no gradient wanted

    // pass 3: dz_j = y_j (G_j - c) on legal j, 0 elsewhere; 16-byte stores over the aligned body
    constexpr int V = KshVec<T>::N;
    const size_t addr = reinterpret_cast<size_t>(drow);
    const int head = (int)(((16 - (addr & 15)) & 15) / sizeof(T));
    const int nvec = (KSH_PA - head) / V;
    const int tail0 = head + nvec * V;
    auto grad = [&](int j, float z) -> float {
        const float y = expf(z - m) * inv_s;
        const bool u = (y >= KSH_EPS) && (y <= 1.0f - KSH_EPS);
        const float l = logf(fminf(fmaxf(y, KSH_EPS), 1.0f - KSH_EPS));
        // (the taken action's policy term is added as it is, not as y * (pol / y): a probability that underflowed to 0 has
        // pol = 0 and must give 0, not 0 / 0)
        const float G = ce * (l + (u ? 1.0f : 0.0f));
        return fmaf(y, G - c, j == act ? pol : 0.0f);
    };
    if (tid < head) {
        const bool on = !none && ((R.bits[0] >> tid) & 1u);
        ksh_st(drow + tid, on ? grad(tid, ksh_ld(row + tid)) : 0.0f);
    }
    // the logits row and the dlogits row have the same alignment phase (same dtype, same row stride) only if
    // both base pointers agree mod 16; read logits element-wise inside a group to stay independent of that.
    for (int v = tid; v < nvec; v += KSH_ROW_THREADS) {
        const int e0 = head + v * V;
        const uint32_t x = none ? 0u : (ksh_bits_at(R.bits, e0) & ((1u << V) - 1u));
        uint4 q = {0u, 0u, 0u, 0u};
        if (x) {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < V; k++) {
                if ((x >> k) & 1u) {
                    const float gz = grad(e0 + k, ksh_ld(row + e0 + k));
                    if (V == 4) {
                        union { float f; uint32_t u; } cv;
                        cv.f = gz;
                        w[k] = cv.u;
                    } else {
                        w[k >> 1] |= (uint32_t)ksh_bf16_bits(gz) << ((k & 1) ? 16 : 0);
                    }
                }
            }
            q.x = w[0]; q.y = w[1]; q.z = w[2]; q.w = w[3];
        }
        *reinterpret_cast<uint4 *>(drow + e0) = q;
    }
    if (tid < KSH_PA - tail0) {
        const int j = tail0 + tid;
        const bool on = !none && ((R.bits[j >> 5] >> (j & 31)) & 1u);
        ksh_st(drow + j, on ? grad(j, ksh_ld(row + j)) : 0.0f);
    }
}

// ---- sparse fast path: one WAVE per row (4 rows per block, no block barrier).  A real position has ~50 legal
// actions: their ids go to a per-wave LDS list, their logits are gathered once, the three passes run over the list,
// the gradient row is zero-filled with 16-byte stores and the few non-zero entries are scattered afterwards (behind
// an s_waitcnt vmcnt(0), so the scatter stores are ordered after the acknowledged zero stores).  Rows with more
// than KSH_LOSS_CAP legal actions (dense masks in tests) are only flagged here and done by the block kernel above.
#define KSH_LOSS_WPB 4
#define KSH_LOSS_CAP 1024

template <typename T>
__global__ void __launch_bounds__(64 * KSH_LOSS_WPB)
ksh_ppo_loss_sparse_kernel(KshLossArgs a, const T *__restrict__ logits, T *__restrict__ dlogits) {
    __shared__ unsigned short s_j[KSH_LOSS_WPB][KSH_LOSS_CAP];
    __shared__ float s_z[KSH_LOSS_WPB][KSH_LOSS_CAP];
    __shared__ unsigned int s_n[KSH_LOSS_WPB];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * KSH_LOSS_WPB + wv;
    if (b >= a.B) return;  // wave-uniform
    const T *row = logits + (size_t)b * KSH_PA;
    T *drow = dlogits + (size_t)b * KSH_PA;
    if (lane == 0) s_n[wv] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    auto push = [&](int j) {
        const unsigned slot = atomicAdd(&s_n[wv], 1u);
        if (slot < KSH_LOSS_CAP) s_j[wv][slot] = (unsigned short)j;
    };
    if (a.mask_bits != nullptr) {
        const uint32_t *brow = a.mask_bits + (size_t)b * KSH_PBITROW;
        uint32_t wd[7];
#pragma unroll
        for (int it = 0; it < 7; it++) {
            const int i = lane + 64 * it;
            wd[it] = i < KSH_PBITROW ? brow[i] : 0u;
        }
#pragma unroll
        for (int it = 0; it < 7; it++) {
            uint32_t v = wd[it];
            const int j0 = 32 * (lane + 64 * it);
            while (v) {
                const int bit = __ffs((int)v) - 1;
                v &= v - 1u;
                if (j0 + bit < KSH_PA) push(j0 + bit);
            }
        }
    } else {
        const uint8_t *mrow = a.mask + (size_t)b * KSH_PA;
        const size_t maddr = reinterpret_cast<size_t>(mrow);
        const int mhead = (int)((16 - (maddr & 15)) & 15);
        const int mfull = (KSH_PA - mhead) >> 4;
        const int mtail0 = mhead + 16 * mfull;
        if (lane < mhead && mrow[lane]) push(lane);
        if (lane < KSH_PA - mtail0 && mrow[mtail0 + lane]) push(mtail0 + lane);
        for (int half = 0; half < 2; half++) {
            uint4 ch[7];
#pragma unroll
            for (int it = 0; it < 7; it++) {
                const int c = lane + 64 * (7 * half + it);
                ch[it] = uint4{0u, 0u, 0u, 0u};
                if (c < mfull) ch[it] = *reinterpret_cast<const uint4 *>(mrow + mhead + 16 * c);
            }
#pragma unroll
            for (int it = 0; it < 7; it++) {
                if ((ch[it].x | ch[it].y | ch[it].z | ch[it].w) != 0u) {
                    const int j0 = mhead + 16 * (lane + 64 * (7 * half + it));
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        uint32_t v = q == 0 ? ch[it].x : (q == 1 ? ch[it].y : (q == 2 ? ch[it].z : ch[it].w));
                        while (v) {
                            const int byte = (__ffs((int)v) - 1) >> 3;
                            v &= ~(0xFFu << (8 * byte));
                            push(j0 + 4 * q + byte);
                        }
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n = (int)s_n[wv];
    float *part = a.partial + (size_t)b * 8;
    if (n > KSH_LOSS_CAP) {          // wave-uniform: leave the row to the dense kernel
        if (lane == 0) part[6] = 1.0f;
        return;
    }
    // gather the legal logits once; pass 1: soft-max statistics
    float m = -INFINITY, s = 0.0f;
    for (int e = lane; e < n; e += 64) {
        const float z = ksh_ld(row + s_j[wv][e]);
        s_z[wv][e] = z;
        ksh_ms_merge(m, s, z, 1.0f);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        ksh_ms_merge(m, s, m2, s2);
    }
    const bool none = n == 0;
    const float inv_s = none ? 0.0f : 1.0f / s;
    const int act = (int)a.actions[b];
    const float invB = 1.0f / (float)a.B;
    // pass 2
    float e1 = 0.0f, e2 = 0.0f, ya = 0.0f;
    for (int e = lane; e < n; e += 64) {
        const float y = expf(s_z[wv][e] - m) * inv_s;
        const bool u = (y >= KSH_EPS) && (y <= 1.0f - KSH_EPS);
        const float l = logf(fminf(fmaxf(y, KSH_EPS), 1.0f - KSH_EPS));
        e1 += y * l;
        e2 += u ? y : 0.0f;
        if ((int)s_j[wv][e] == act) ya = y;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        e1 += __shfl_xor(e1, o);
        e2 += __shfl_xor(e2, o);
        ya += __shfl_xor(ya, o);
    }
    float new_logp, H;
    bool ua;
    if (none) {
        new_logp = logf(1.0f / (float)KSH_PA);
        H = -new_logp;
        ua = false;
        ya = 1.0f;
    } else {
        ua = (ya >= KSH_EPS) && (ya <= 1.0f - KSH_EPS);
        new_logp = logf(fminf(fmaxf(ya, KSH_EPS), 1.0f - KSH_EPS));
        H = -e1;
    }
    const float old_lp = a.old_logp[b], A = a.adv[b];
    const float ratio = expf(new_logp - old_lp);
    const float surr1 = ratio * A;
    const float surr2 = fminf(fmaxf(ratio, 1.0f - a.clip_eps), 1.0f + a.clip_eps) * A;
    const float gr = (surr1 <= surr2) ? A : 0.0f;
    const float pol = (ua && !none) ? (-gr * ratio * invB) : 0.0f;
    const float ce = a.ent_coef * invB;
    const float c = ce * (e1 + e2) + pol;
    if (lane == 0) {
        const float V = a.values[b], Rt = a.ret[b];
        part[0] = fminf(surr1, surr2);
        part[1] = (V - Rt) * (V - Rt);
        float vc = 0.0f;
        if (a.value_clip) {
            const float ov = a.old_values[b];
            const float Vc = ov + fminf(fmaxf(V - ov, -a.clip_eps), a.clip_eps);
            vc = (Vc - Rt) * (Vc - Rt);
        }
        part[2] = vc;
        part[3] = H;
        part[4] = old_lp - new_logp;
        part[5] = (fabsf(ratio - 1.0f) > a.clip_eps) ? 1.0f : 0.0f;
        part[6] = 0.0f;
        part[7] = 0.0f;
        if (a.row_logp != nullptr) a.row_logp[b] = new_logp;
        if (a.row_entropy != nullptr) a.row_entropy[b] = H;
    }
    if (dlogits == nullptr) return;   // evaluate-only call: no gradient wanted (wave-uniform)
    // pass 3: gradients of the legal entries (kept in LDS), zero row, scatter
    for (int e = lane; e < n; e += 64) {
        const float y = expf(s_z[wv][e] - m) * inv_s;
        const bool u = (y >= KSH_EPS) && (y <= 1.0f - KSH_EPS);
        const float l = logf(fminf(fmaxf(y, KSH_EPS), 1.0f - KSH_EPS));
        const float G = ce * (l + (u ? 1.0f : 0.0f));
        s_z[wv][e] = fmaf(y, G - c, (int)s_j[wv][e] == act ? pol : 0.0f);     // (not y * (pol / y): see the dense kernel)
    }
    constexpr int V = KshVec<T>::N;
    const size_t addr = reinterpret_cast<size_t>(drow);
    const int head = (int)(((16 - (addr & 15)) & 15) / sizeof(T));
    const int nvec = (KSH_PA - head) / V;
    const int tail0 = head + nvec * V;
    if (lane < head) ksh_st(drow + lane, 0.0f);
    if (lane < KSH_PA - tail0) ksh_st(drow + tail0 + lane, 0.0f);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (int v = lane; v < nvec; v += 64) *reinterpret_cast<uint4 *>(drow + head + v * V) = zero;
    if (n > 0) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every zero store of this wave has been acknowledged
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < n; e += 64) ksh_st(drow + s_j[wv][e], s_z[wv][e]);
    }
}

// Batch means, the value-loss branch (max of the two MEANS, ppo_agent.py:357-373) and d(loss)/d(values).
// One block of KSH_STATS_THREADS threads (any power of two up to that works); fixed-order tree reduction.
#define KSH_STATS_THREADS 1024
struct KshStatsLds {
    float red[KSH_STATS_THREADS][6];
    float tot[6];
    int bad;
};
// the body, for one whole block of blockDim.x (a power of two <= KSH_STATS_THREADS) threads: ksh_ppo_stats_kernel below, and
// one block of the fused head's bit-matrix transpose launch (ksh_head_kernels.h), which saves a dependent launch
__device__ __forceinline__ void ksh_ppo_stats_body(const KshLossArgs &a, KshStatsLds &S) {
    float (*red)[6] = S.red;
    float *tot = S.tot;
    int &bad = S.bad;
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    int flagged = 0;
    for (int i = tid; i < a.B; i += nt) {
        const float4 *p = reinterpret_cast<const float4 *>(a.partial + (size_t)i * 8);   // rows are 32 bytes
        const float4 u = p[0], v = p[1];
        acc[0] += u.x; acc[1] += u.y; acc[2] += u.z; acc[3] += u.w; acc[4] += v.x; acc[5] += v.y;
        flagged |= v.z != 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) red[tid][k] = acc[k];
    if (tid == 0) bad = 0;
    __syncthreads();
    if (flagged) bad = 1;
    for (int o = nt >> 1; o >= 1; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int k = 0; k < 6; k++) red[tid][k] += red[tid + o][k];
        }
        __syncthreads();
    }
    if (tid < 6) tot[tid] = red[0][tid] / (float)a.B;
    __syncthreads();
    if (a.overflow_is_error && bad) {
        // the caller promised sparse rows (<= 1024 legal actions) and skipped the dense path: a flagged row was never
        // evaluated, so nothing computed from this minibatch may look valid
        if (tid < 8) a.stats[tid] = NAN;
        for (int i = tid; i < a.B; i += nt) a.dvalues[i] = NAN;
        return;
    }
    const float policy_loss = -tot[0];
    const float vu = tot[1], vc = tot[2];
    float wu = 1.0f, wc = 0.0f;  // weights of the unclipped / clipped branch in d value_loss
    float value_loss = vu;
    if (a.value_clip) {
        value_loss = fmaxf(vu, vc);
        if (vu > vc) { wu = 1.0f; wc = 0.0f; }
        else if (vc > vu) { wu = 0.0f; wc = 1.0f; }
        else { wu = 0.5f; wc = 0.5f; }  // torch.max(a, b) splits the gradient on ties
    }
    const float entropy_loss = -tot[3];
    if (tid == 0) {
        a.stats[0] = policy_loss + a.vf_coef * value_loss + a.ent_coef * entropy_loss;
        a.stats[1] = policy_loss;
        a.stats[2] = value_loss;
        a.stats[3] = entropy_loss;
        a.stats[4] = tot[4];
        a.stats[5] = tot[5];
        a.stats[6] = 0.0f;
        a.stats[7] = 0.0f;
    }
    const float k2 = a.vf_coef * 2.0f / (float)a.B;
    for (int i = tid; i < a.B; i += nt) {
        const float V = a.values[i], Rt = a.ret[i];
        float g = wu * k2 * (V - Rt);
        if (a.value_clip && wc != 0.0f) {
            const float ov = a.old_values[i];
            const float dv = V - ov;
            const float Vc = ov + fminf(fmaxf(dv, -a.clip_eps), a.clip_eps);
            const float pass = (dv >= -a.clip_eps && dv <= a.clip_eps) ? 1.0f : 0.0f;
            g += wc * k2 * (Vc - Rt) * pass;
        }
        a.dvalues[i] = g;
    }
}

extern "C" __global__ void __launch_bounds__(KSH_STATS_THREADS) ksh_ppo_stats_kernel(KshLossArgs a) {
    __shared__ KshStatsLds S;
    ksh_ppo_stats_body(a, S);
}
