// ksh_head_kernels.h -- the policy head of the actor-critic FUSED with the masked soft-max tail, for gfx950.
//
// What it replaces (reference = /root/reference/keisei/):
//   training/models/resnet_tower.py:63-69   policy_head's nn.Linear(2*9*9, 13527)  (core/neural_network.py:17 for the tiny CNN)
//   core/base_actor_critic.py:64-116        get_action_and_value tail: masked soft-max -> sample / arg-max -> log-prob
//   core/base_actor_critic.py:139-184       evaluate_actions tail: log-prob, entropy
//   core/ppo_agent.py:340-383               clipped-surrogate loss lines; and the autograd backward of all of it
//
// The reference materialises logits[B,13527] (hipBLASLt: 221 MB of float32 per 4096-row minibatch), masks them, and
// autograd later reads a dense dlogits of the same size back through two more GEMMs.  A real position has ~50 legal
// actions out of 13 527 (never more than 593), so 99.6 % of that contraction multiplies into -inf.  Here:
//
//   sparse rows (the production path, n_legal <= 1024): one WAVE per sample.  The packed legal bit-set becomes an id
//       list in LDS, and only those rows of W are gathered: z_j = <feat_b, W_j> + bias_j with 8 lanes per action
//       (16 actions in flight per wave, 16-byte loads, 3 DPP-sized reduction steps).  Soft-max statistics, sampling or the
//       PPO loss and its gradient dz stay in LDS; dfeat_b = sum_j dz_j W_j re-gathers the same (cache-resident) rows;
//       dz is scattered into a transposed scratch dzT[a][b] + a transposed bit matrix, from which the weight-gradient
//       kernels build dW_a = sum_b dzT[a][b] feat_b and dbias_a in ascending b: no float atomics, bit-reproducible.
//       Logits and dlogits never exist in HBM.
//   dense rows (n_legal > 1024: only hand-made masks, e.g. the all-ones masks of the reference's unit tests): the
//       contraction is genuinely dense there, so it runs on the matrix cores: an LDS-tiled float32 MFMA GEMM
//       (v_mfma_f32_32x32x2_f32: exact float32, no xf32 on gfx950) produces the logits of the flagged rows, the
//       block-per-row loss kernel of ksh_ppo_kernels.h does the soft-max / loss / dlogits, and two more MFMA GEMMs
//       produce their dfeat rows and add their dW / dbias contribution.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ksh_ppo_kernels.h"

#ifndef KSH_HEAD_WPB
#define KSH_HEAD_WPB 4      // waves (= rows) per block of the row kernels
#endif
#define KSH_HEAD_CAP 1024   // legal ids per row the sparse path holds (Shogi's maximum is 593)

struct KshHeadArgs {
    const float *feat;          // [B][K] policy features (input of the Linear layer)
    const float *W;             // [A][K] nn.Linear.weight
    const float *bias;          // [A] or null
    int K;                      // even, <= KMAX of the instantiation
    const uint32_t *mask_bits;  // [B][424] packed legal masks
    int B;
    // ---- sampling (ksh_head_sample_kernel)
    unsigned long long seed, step;
    const unsigned long long *step_dev;
    int deterministic;
    long long *action;
    float *log_prob;
    // ---- loss (ksh_head_loss_kernel); `L` carries the per-row inputs, partial sums and optional per-row outputs
    KshLossArgs L;
    int want_grad;              // 0: evaluate only (log-prob / entropy / loss terms), nothing below is touched
    float *dfeat;               // [B][K]
    float *dzT;                 // [A][B] scratch: d loss / d logit, transposed, only legal entries are ever written
    uint32_t *bitsT;            // [A][BW] transposed legal bit matrix (BW = ceil(B/32)): written by ksh_head_bits_transpose_kernel
    int BW;
    unsigned *n_over;           // counter of rows left to the dense path
};

__device__ __forceinline__ void ksh_head_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Legal ids of one packed row -> list[] in ASCENDING order, without atomics: lane l owns dwords 7l .. 7l+6 of the row,
// an inclusive scan of the per-lane popcounts gives every lane its output offset.  Returns the number of legal ids;
// ids beyond `cap` are not stored.
__device__ __forceinline__ int ksh_head_collect(const uint32_t *__restrict__ brow, int lane, unsigned short *list, int cap) {
    uint32_t wd[7];
    int c = 0;
#pragma unroll
    for (int it = 0; it < 7; it++) {
        const int i = 7 * lane + it;
        uint32_t v = brow[i < KSH_PBITROW ? i : KSH_PBITROW - 1];     // unconditional load (clamped), masked below
        v = i < 422 ? v : (i == 422 ? (v & 0x007FFFFFu) : 0u);        // bits >= 13527 never count
        wd[it] = v;
        c += __popc(v);
    }
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl(incl, lane >= o ? lane - o : 0);
        if (lane >= o) incl += t;
    }
    const int n = __shfl(incl, 63);
    int pos = incl - c;
#pragma unroll
    for (int it = 0; it < 7; it++) {
        uint32_t v = wd[it];
        const int j0 = 32 * (7 * lane + it);
        while (v) {
            const int bit = __ffs((int)v) - 1;
            v &= v - 1u;
            if (pos < cap) list[pos] = (unsigned short)(j0 + bit);
            pos++;
        }
    }
    ksh_head_wave_sync();
    return n;
}

// Rows of W and of feat start 8-byte aligned (K is even, not a multiple of 4 in general: 162): a row is read as
// "quads" of 16 bytes -- gfx9 global loads only need dword alignment -- plus one trailing pair when K % 4 == 2.
struct __attribute__((packed, aligned(8))) ksh_f4a8 {
    float x, y, z, w;
};
// BRANCH-FREE on purpose: one unconditional 16-byte load per call and selects afterwards.  (A version with an `if` per
// case compiled into exec-mask branches with an s_waitcnt vmcnt(0) behind EVERY load -- the gathers of a wave then ran
// one after the other, which is what the first ISA dump of the row kernel showed.)  The trailing pair is fetched as
// the upper half of the quad that ENDS at the row's end (elements K-4 .. K-1), so no load ever leaves the row; lanes
// beyond the row read its first quad and discard it.  Needs K >= 4.
__device__ __forceinline__ float4 ksh_ld_quad(const float *row, int q, int nq, bool tail) {
    const bool full = q < nq;
    const bool last = q == nq && tail;
    const int off = full ? 4 * q : (last ? 4 * nq - 2 : 0);
    const ksh_f4a8 v = *reinterpret_cast<const ksh_f4a8 *>(row + off);
    // bit masks, not selects: every component of the load is consumed unconditionally, so the compiler cannot split the
    // load and sink parts of it into branches again
    const uint32_t mf = full ? 0xFFFFFFFFu : 0u, ml = last ? 0xFFFFFFFFu : 0u;
    const uint32_t x = __float_as_uint(v.x), y = __float_as_uint(v.y), z = __float_as_uint(v.z), w = __float_as_uint(v.w);
    return make_float4(ksh_u2f((x & mf) | (z & ml)), ksh_u2f((y & mf) | (w & ml)), ksh_u2f(z & mf), ksh_u2f(w & mf));
}

// z_e = <feat, W[list[e]]> + bias for the first n list entries.  8 lanes per action: lane s of a sub-group owns the
// quads q = s, s+8, ... of a row (8 lanes x 16 B = one 128-byte line per load instruction and sub-group); every
// sub-group works on TWO actions at once (e and e+8), CH quads each, so 2*CH independent 16-byte loads are in flight
// per lane before the first multiply.  feat: this wave's feature row in LDS (16-byte aligned, zero-padded to KMAX).
#ifndef KSH_HEAD_NA
#define KSH_HEAD_NA 2      // actions per sub-group and round of ksh_head_logits (1 and 3 measured: profiles/r02_head_kernels_pmc.txt)
#endif
template <int KMAX>
__device__ __forceinline__ void ksh_head_logits(const KshHeadArgs &a, const float *feat, const unsigned short *list, int n,
                                                int lane, float *z) {
    constexpr int CH = (KMAX / 4 + 7) / 8 < 6 ? (KMAX / 4 + 7) / 8 : 6;   // quads per lane and chunk
    constexpr int NA = KSH_HEAD_NA;
    const int s = lane & 7, g = lane >> 3;
    const int nq = a.K >> 2;
    const bool tail = (a.K & 2) != 0;
    const int nqt = nq + (tail ? 1 : 0);
    const float4 *f4 = reinterpret_cast<const float4 *>(feat);
    for (int e0 = 0; e0 < n; e0 += 8 * NA) {       // wave-uniform trip count
        int e[NA], j[NA];
        const float *r[NA];
        float acc[NA];
#pragma unroll
        for (int u = 0; u < NA; u++) {
            e[u] = e0 + 8 * u + g;
            j[u] = (int)list[e[u] < n ? e[u] : 0];
            r[u] = a.W + (size_t)j[u] * (size_t)a.K;
            acc[u] = 0.0f;
        }
        for (int c0 = 0; c0 < nqt; c0 += 8 * CH) {  // one round for K = 162
            float4 w[NA][CH];
#pragma unroll
            for (int c = 0; c < CH; c++) {
#pragma unroll
                for (int u = 0; u < NA; u++) w[u][c] = ksh_ld_quad(r[u], c0 + s + 8 * c, nq, tail);
            }
#pragma unroll
            for (int c = 0; c < CH; c++) {
                const int q = c0 + s + 8 * c;
                const float4 f = q < nqt ? f4[q] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int u = 0; u < NA; u++) {
                    acc[u] = fmaf(w[u][c].x, f.x, acc[u]); acc[u] = fmaf(w[u][c].y, f.y, acc[u]);
                    acc[u] = fmaf(w[u][c].z, f.z, acc[u]); acc[u] = fmaf(w[u][c].w, f.w, acc[u]);
                }
            }
        }
#pragma unroll
        for (int o = 1; o <= 4; o <<= 1) {
#pragma unroll
            for (int u = 0; u < NA; u++) acc[u] += __shfl_xor(acc[u], o);
        }
#pragma unroll
        for (int u = 0; u < NA; u++)
            if (s == 0 && e[u] < n) z[e[u]] = acc[u] + (a.bias != nullptr ? a.bias[j[u]] : 0.0f);
    }
    ksh_head_wave_sync();
}

// one lane's own dot product (ids beyond the list capacity in the sampler: hand-made dense masks only)
__device__ __forceinline__ float ksh_head_logit_serial(const KshHeadArgs &a, const float *feat, int j) {
    const float *wrow = a.W + (size_t)j * (size_t)a.K;
    float acc = 0.0f;
    for (int k = 0; k < a.K; k++) acc = fmaf(wrow[k], feat[k], acc);
    return acc + (a.bias != nullptr ? a.bias[j] : 0.0f);
}

template <int KMAX>
struct KshHeadLds {
    unsigned short list[KSH_HEAD_WPB][KSH_HEAD_CAP];
    float z[KSH_HEAD_WPB][KSH_HEAD_CAP];
    __attribute__((aligned(16))) float feat[KSH_HEAD_WPB][KMAX];
};

template <int KMAX>
__device__ __forceinline__ void ksh_head_stage_feat(const KshHeadArgs &a, int b, int lane, float *dst) {
    const float2 *src = reinterpret_cast<const float2 *>(a.feat + (size_t)b * (size_t)a.K);
    float2 *d2 = reinterpret_cast<float2 *>(dst);
    const int np = a.K >> 1;
    for (int p = lane; p < ((np + 1) & ~1); p += 64) d2[p] = p < np ? src[p] : make_float2(0.0f, 0.0f);   // padded to a quad
    ksh_head_wave_sync();
}
// K <= 192 (at most 96 pairs: two per lane): the feature row is FETCHED at the top of the kernel, next to the mask row, and
// only written to LDS after the id list has been built -- one global round trip on the wave's critical path instead of two
struct KshFeatRegs { float2 r0, r1; };
__device__ __forceinline__ KshFeatRegs ksh_head_fetch_feat(const KshHeadArgs &a, int b, int lane) {
    const float2 *src = reinterpret_cast<const float2 *>(a.feat + (size_t)b * (size_t)a.K);
    const int np = a.K >> 1;
    KshFeatRegs f;
    const float2 x0 = src[lane < np ? lane : 0], x1 = src[lane + 64 < np ? lane + 64 : 0];      // unconditional (clamped) loads
    f.r0 = lane < np ? x0 : make_float2(0.0f, 0.0f);
    f.r1 = lane + 64 < np ? x1 : make_float2(0.0f, 0.0f);
    return f;
}
__device__ __forceinline__ void ksh_head_put_feat(const KshHeadArgs &a, const KshFeatRegs &f, int lane, float *dst) {
    float2 *d2 = reinterpret_cast<float2 *>(dst);
    const int np2 = ((a.K >> 1) + 1) & ~1;                    // padded to a quad
    if (lane < np2) d2[lane] = f.r0;
    if (lane + 64 < np2) d2[lane + 64] = f.r1;
    ksh_head_wave_sync();
}

// ------------------------------------------------------------------------------------------ fused head + sampler
// get_action_and_value's tail without logits in memory: gather-dot of the legal rows of W, Gumbel-max sample (or
// arg-max), Categorical log-prob.  Same arithmetic after the logits as ksh_masked_sample_kernel.
template <int KMAX>
__global__ void __launch_bounds__(64 * KSH_HEAD_WPB) ksh_head_sample_kernel(KshHeadArgs a) {
    __shared__ KshHeadLds<KMAX> S;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * KSH_HEAD_WPB + wv;
    if (b >= a.B) return;  // wave-uniform
    unsigned long long step = a.step;
    if (a.step_dev != nullptr) step += a.step_dev[0];
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32), s0 = (uint32_t)step, s1 = (uint32_t)(step >> 32);
    const bool det = a.deterministic != 0;
    const uint32_t *brow = a.mask_bits + (size_t)b * KSH_PBITROW;
    KshFeatRegs fr;
    if constexpr (KMAX <= 192) fr = ksh_head_fetch_feat(a, b, lane);
    const int total = ksh_head_collect(brow, lane, S.list[wv], KSH_HEAD_CAP);
    const int n = total < KSH_HEAD_CAP ? total : KSH_HEAD_CAP;
    if constexpr (KMAX <= 192) ksh_head_put_feat(a, fr, lane, S.feat[wv]);
    else ksh_head_stage_feat<KMAX>(a, b, lane, S.feat[wv]);
    ksh_head_logits<KMAX>(a, S.feat[wv], S.list[wv], n, lane, S.z[wv]);
    KshSampleAcc A = {-INFINITY, 0.0f, -INFINITY, 0.0f, -1};
    auto visit = [&](int j, float z) {
        ksh_ms_merge(A.m, A.s, z, 1.0f);
        float k = z;
        if (!det) k = z - logf(-logf(ksh_uniform_open(ksh_philox((uint32_t)j, (uint32_t)b, s0, s1, k0, k1))));
        if (A.idx < 0 || k > A.key || (k == A.key && j < A.idx)) { A.key = k; A.idx = j; A.zbest = z; }
    };
    for (int e = lane; e < n; e += 64) visit((int)S.list[wv][e], S.z[wv][e]);
    if (total > KSH_HEAD_CAP) {
        // dense hand-made mask: the ids that did not fit are the ones after the first CAP set bits; every lane re-walks
        // its own dwords and computes those logits by itself (correct for any density; production rows never get here)
        int seen = 0;
        {   // number of set bits in the dwords of lower lanes = this lane's starting rank
            int c = 0;
            for (int it = 0; it < 7; it++) {
                const int i = 7 * lane + it;
                uint32_t v = i < KSH_PBITROW ? brow[i] : 0u;
                if (i == 422) v &= 0x007FFFFFu;
                if (i >= 423) v = 0u;
                c += __popc(v);
            }
            int incl = c;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl(incl, lane >= o ? lane - o : 0);
                if (lane >= o) incl += t;
            }
            seen = incl - c;
        }
        for (int it = 0; it < 7; it++) {
            const int i = 7 * lane + it;
            uint32_t v = i < KSH_PBITROW ? brow[i] : 0u;
            if (i == 422) v &= 0x007FFFFFu;
            if (i >= 423) v = 0u;
            while (v) {
                const int bit = __ffs((int)v) - 1;
                v &= v - 1u;
                if (seen >= KSH_HEAD_CAP) visit(32 * i + bit, ksh_head_logit_serial(a, S.feat[wv], 32 * i + bit));
                seen++;
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(A.m, o), s2 = __shfl_xor(A.s, o);
        ksh_ms_merge(A.m, A.s, m2, s2);
        const float k2 = __shfl_xor(A.key, o), z2 = __shfl_xor(A.zbest, o);
        const int i2 = __shfl_xor(A.idx, o);
        if (i2 >= 0 && (A.idx < 0 || k2 > A.key || (k2 == A.key && i2 < A.idx))) { A.key = k2; A.idx = i2; A.zbest = z2; }
    }
    if (lane == 0) {
        long long act;
        float lp;
        if (A.idx < 0) {   // no legal action: uniform over ALL actions (base_actor_critic.py:93-101)
            const uint32_t x = ksh_philox(0xFFFFFFFFu, (uint32_t)b, s0, s1, k0, k1);
            act = det ? 0 : (long long)(((unsigned long long)x * (unsigned long long)KSH_PA) >> 32);
            lp = logf(1.0f / (float)KSH_PA);
        } else {
            act = A.idx;
            float p = expf(A.zbest - A.m) / A.s;
            p = fminf(fmaxf(p, KSH_EPS), 1.0f - KSH_EPS);
            lp = logf(p);
        }
        a.action[b] = act;
        a.log_prob[b] = lp;
    }
}

// ------------------------------------------------------------------------------------------ fused head + PPO loss
// One wave per sample: logits of the legal actions (never stored), evaluate_actions' log-prob / entropy, the loss terms
// (-> L.partial), and -- when want_grad -- dfeat[b] and the transposed sparse dz for the weight-gradient kernel.
template <int KMAX>
__global__ void __launch_bounds__(64 * KSH_HEAD_WPB) ksh_head_loss_kernel(KshHeadArgs a) {
    __shared__ KshHeadLds<KMAX> S;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * KSH_HEAD_WPB + wv;
    if (b >= a.B) return;  // wave-uniform
    const KshLossArgs &L = a.L;
    float *part = L.partial + (size_t)b * 8;
    // every global input of this row is requested up front (the loads would otherwise sit, one round trip each, behind the
    // fences of the phases that need them)
    KshFeatRegs fr;
    if constexpr (KMAX <= 192) fr = ksh_head_fetch_feat(a, b, lane);
    const int act = (int)L.actions[b];
    // (the evaluate-only call has no loss inputs: null pointers read as zeros)
    const float old_lp = L.old_logp != nullptr ? L.old_logp[b] : 0.0f, Adv = L.adv != nullptr ? L.adv[b] : 0.0f;
    const float Vin = L.values != nullptr ? L.values[b] : 0.0f, Rt = L.ret != nullptr ? L.ret[b] : 0.0f;
    const float ov_in = L.value_clip ? L.old_values[b] : 0.0f;
    const int n = ksh_head_collect(a.mask_bits + (size_t)b * KSH_PBITROW, lane, S.list[wv], KSH_HEAD_CAP);
    if (n > KSH_HEAD_CAP) {            // wave-uniform: a dense row, left to the MFMA path
        if (lane == 0) {
            part[6] = 1.0f;
            atomicAdd(a.n_over, 1u);
            if (L.overflow_is_error) {  // ... which the caller said would not be needed: the row's outputs must not look valid
                if (L.row_logp != nullptr) L.row_logp[b] = NAN;
                if (L.row_entropy != nullptr) L.row_entropy[b] = NAN;
            }
        }
        return;
    }
    unsigned short *list = S.list[wv];
    float *z = S.z[wv];
    if constexpr (KMAX <= 192) ksh_head_put_feat(a, fr, lane, S.feat[wv]);
    else ksh_head_stage_feat<KMAX>(a, b, lane, S.feat[wv]);
    ksh_head_logits<KMAX>(a, S.feat[wv], list, n, lane, z);
    // pass 1: soft-max statistics
    float m = -INFINITY, s = 0.0f;
    for (int e = lane; e < n; e += 64) ksh_ms_merge(m, s, z[e], 1.0f);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        ksh_ms_merge(m, s, m2, s2);
    }
    const bool none = n == 0;
    const float inv_s = none ? 0.0f : 1.0f / s;
    const float invB = 1.0f / (float)L.B;
    // pass 2: entropy terms and the taken action's probability
    float e1 = 0.0f, e2 = 0.0f, ya = 0.0f;
    for (int e = lane; e < n; e += 64) {
        const float y = expf(z[e] - m) * inv_s;
        const bool u = (y >= KSH_EPS) && (y <= 1.0f - KSH_EPS);
        const float l = logf(fminf(fmaxf(y, KSH_EPS), 1.0f - KSH_EPS));
        e1 += y * l;
        e2 += u ? y : 0.0f;
        if ((int)list[e] == act) ya = y;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        e1 += __shfl_xor(e1, o);
        e2 += __shfl_xor(e2, o);
        ya += __shfl_xor(ya, o);
    }
    float new_logp, H;
    bool ua;
    if (none) {   // all-illegal row -> uniform over all actions, detached (base_actor_critic.py:168-174)
        new_logp = logf(1.0f / (float)KSH_PA);
        H = -new_logp;
        ua = false;
        ya = 1.0f;
    } else {
        ua = (ya >= KSH_EPS) && (ya <= 1.0f - KSH_EPS);
        new_logp = logf(fminf(fmaxf(ya, KSH_EPS), 1.0f - KSH_EPS));
        H = -e1;
    }
    const float ratio = expf(new_logp - old_lp);
    const float surr1 = ratio * Adv;
    const float surr2 = fminf(fmaxf(ratio, 1.0f - L.clip_eps), 1.0f + L.clip_eps) * Adv;
    const float gr = (surr1 <= surr2) ? Adv : 0.0f;
    const float pol = (ua && !none) ? (-gr * ratio * invB) : 0.0f;
    const float ce = L.ent_coef * invB;
    const float c = ce * (e1 + e2) + pol;
    if (lane == 0) {
        const float V = Vin;
        part[0] = fminf(surr1, surr2);
        part[1] = (V - Rt) * (V - Rt);
        float vc = 0.0f;
        if (L.value_clip) {
            const float ov = ov_in;
            const float Vc = ov + fminf(fmaxf(V - ov, -L.clip_eps), L.clip_eps);
            vc = (Vc - Rt) * (Vc - Rt);
        }
        part[2] = vc;
        part[3] = H;
        part[4] = old_lp - new_logp;
        part[5] = (fabsf(ratio - 1.0f) > L.clip_eps) ? 1.0f : 0.0f;
        part[6] = 0.0f;
        part[7] = 0.0f;
        if (L.row_logp != nullptr) L.row_logp[b] = new_logp;
        if (L.row_entropy != nullptr) L.row_entropy[b] = H;
    }
    if (!a.want_grad) return;   // wave-uniform
    // pass 3: dz_e = y_e (G_e - c) in place of the logits
    for (int e = lane; e < n; e += 64) {
        const float y = expf(z[e] - m) * inv_s;
        const bool u = (y >= KSH_EPS) && (y <= 1.0f - KSH_EPS);
        const float l = logf(fminf(fmaxf(y, KSH_EPS), 1.0f - KSH_EPS));
        const float G = ce * (l + (u ? 1.0f : 0.0f));
        z[e] = fmaf(y, G - c, (int)list[e] == act ? pol : 0.0f);     // (not y * (pol / y): 0 / 0 when y underflowed to 0)
    }
    ksh_head_wave_sync();
    // dfeat[b] = sum_e dz_e W[list[e]].  K <= 192: four 16-lane groups take every fourth entry (lane s of a group owns quads
    // s, s+16, s+32 of the row), 16 rows in flight per round -- a row with 48 legal actions is three rounds, where one
    // entry per load instruction (lane = quad, U in flight) was twelve; the groups' sums are added in a fixed order.
    if constexpr (KMAX <= 192) {
        constexpr int U = 4;
        const int gs = lane & 15, gg = lane >> 4;
        const int nq = a.K >> 2;
        const bool tail = (a.K & 2) != 0;
        float4 acc[3];
#pragma unroll
        for (int i = 0; i < 3; i++) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int e0 = 0; e0 < n; e0 += 4 * U) {      // wave-uniform trip count
            float4 w[U][3];
            float dz[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = e0 + 4 * u + gg;
                dz[u] = e < n ? z[e] : 0.0f;
                const float *wrow = a.W + (size_t)list[e < n ? e : 0] * (size_t)a.K;
#pragma unroll
                for (int i = 0; i < 3; i++) w[u][i] = ksh_ld_quad(wrow, gs + 16 * i, nq, tail);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    acc[i].x = fmaf(dz[u], w[u][i].x, acc[i].x); acc[i].y = fmaf(dz[u], w[u][i].y, acc[i].y);
                    acc[i].z = fmaf(dz[u], w[u][i].z, acc[i].z); acc[i].w = fmaf(dz[u], w[u][i].w, acc[i].w);
                }
            }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                acc[i].x += __shfl_xor(acc[i].x, o); acc[i].y += __shfl_xor(acc[i].y, o);
                acc[i].z += __shfl_xor(acc[i].z, o); acc[i].w += __shfl_xor(acc[i].w, o);
            }
        }
        float2 *drow = reinterpret_cast<float2 *>(a.dfeat + (size_t)b * (size_t)a.K);
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int q = gs + 16 * i;
            if (gg == i) {
                if (q < nq || (q == nq && tail)) drow[2 * q] = make_float2(acc[i].x, acc[i].y);
                if (q < nq) drow[2 * q + 1] = make_float2(acc[i].z, acc[i].w);
            }
        }
    } else {
        // wider rows: lanes own the quads q = lane + 64 i of the row, U entries' loads in flight at once
        constexpr int QQ = (KMAX / 4 + 63) / 64;
        constexpr int U = 2;
        const int nq = a.K >> 2;
        const bool tail = (a.K & 2) != 0;
        float4 acc[QQ];
#pragma unroll
        for (int i = 0; i < QQ; i++) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int e0 = 0; e0 < n; e0 += U) {
            float4 w[U][QQ];
            float dz[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = e0 + u;
                dz[u] = e < n ? z[e] : 0.0f;
                const float *wrow = a.W + (size_t)list[e < n ? e : 0] * (size_t)a.K;
#pragma unroll
                for (int i = 0; i < QQ; i++) w[u][i] = ksh_ld_quad(wrow, lane + 64 * i, nq, tail);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
                for (int i = 0; i < QQ; i++) {
                    acc[i].x = fmaf(dz[u], w[u][i].x, acc[i].x); acc[i].y = fmaf(dz[u], w[u][i].y, acc[i].y);
                    acc[i].z = fmaf(dz[u], w[u][i].z, acc[i].z); acc[i].w = fmaf(dz[u], w[u][i].w, acc[i].w);
                }
            }
        }
        float2 *drow = reinterpret_cast<float2 *>(a.dfeat + (size_t)b * (size_t)a.K);
#pragma unroll
        for (int i = 0; i < QQ; i++) {
            const int q = lane + 64 * i;
            if (q < nq || (q == nq && tail)) drow[2 * q] = make_float2(acc[i].x, acc[i].y);
            if (q < nq) drow[2 * q + 1] = make_float2(acc[i].z, acc[i].w);
        }
    }
    // transposed scatter for the weight gradient: dzT[a][b].  (The transposed BIT matrix, which tells the weight-gradient
    // kernels where to look, is built by ksh_head_bits_transpose_kernel from the packed masks: setting bit b of row a from
    // here took one device-scope atomicOr per legal pair, 152 k of them per minibatch = 19 of this kernel's 44 us.)
    for (int e = lane; e < n; e += 64) a.dzT[(size_t)list[e] * (size_t)a.B + (size_t)b] = z[e];
}

// ------------------------------------------------------------------------------------------ transposed bit matrix
// bitsT[a][w] bit i = action a is legal in sample 32 w + i = bit a of the packed mask row of that sample, for the rows the
// sparse path handled (rows flagged dense are left out: the MFMA path owns them).  A bit-matrix transpose through LDS:
// a block takes 512 samples x 512 actions, reads its 64-byte row segments of the packed masks, ORs the (few: 0.3 % are
// set) bits into a 512 x 16-word LDS tile with LDS atomics and writes the tile's rows as 64-byte segments -- zeros
// included, so the matrix needs no memset.  Grid: (ceil(BW / 16), ceil(424 / 16) + 1).
#define KSH_BT_TS 16       // sample words (of 32 samples) per tile
#define KSH_BT_TA 16       // action words (of 32 actions) per tile
#define KSH_BT_THREADS 1024
// The launch carries one extra row of blocks (blockIdx.y == gridDim.y - 1): its first block zeroes the work-list counters
// of the weight-gradient kernels and runs the batch statistics (ksh_ppo_stats_body: the loss terms and dvalues only depend
// on the row kernels before this launch), which saves a memset and a dependent launch of ~5 us each.
static_assert(KSH_BT_THREADS == KSH_STATS_THREADS, "the statistics block runs inside the transpose launch");
extern "C" __global__ void __launch_bounds__(KSH_BT_THREADS) ksh_head_bits_transpose_kernel(
    const uint32_t *__restrict__ mask_bits, const float *__restrict__ flags, int B, int BW, uint32_t *__restrict__ bitsT,
    KshLossArgs L, unsigned *__restrict__ n_items) {
    __shared__ uint32_t T[32 * KSH_BT_TA][KSH_BT_TS + 1];
    __shared__ float s_flag[32 * KSH_BT_TS];
    __shared__ KshStatsLds s_stats;
    const int t = (int)threadIdx.x;
    if (blockIdx.y == gridDim.y - 1) {     // block-uniform
        if (blockIdx.x != 0) return;
        if (t < 3) n_items[t] = 0u;        // light-list length, "corrupt" flag, heavy-list length (the scan kernel is the next launch)
        ksh_ppo_stats_body(L, s_stats);
        return;
    }
    const int w0 = (int)blockIdx.x * KSH_BT_TS, c0 = (int)blockIdx.y * KSH_BT_TA;
    constexpr int NW = 32 * KSH_BT_TS * KSH_BT_TA;       // words of the input tile: 512 samples x 16
    constexpr int PER = NW / KSH_BT_THREADS;             // 8 independent loads per thread
    uint32_t v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int u = t + KSH_BT_THREADS * i;
        const int sl = u >> 4, c = c0 + (u & 15);
        const int b = 32 * w0 + sl;
        const bool in = b < B && c < KSH_PBITROW;
        uint32_t x = mask_bits[(size_t)(in ? b : 0) * KSH_PBITROW + (size_t)(in ? c : 0)];
        x = c < 422 ? x : (c == 422 ? (x & 0x007FFFFFu) : 0u);      // bits >= 13527 never count
        v[i] = in ? x : 0u;
    }
    if (t < 32 * KSH_BT_TS) s_flag[t] = 32 * w0 + t < B ? flags[(size_t)(32 * w0 + t) * 8] : 0.0f;
    for (int i = t; i < 32 * KSH_BT_TA * (KSH_BT_TS + 1); i += KSH_BT_THREADS) (&T[0][0])[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++) {
        uint32_t x = v[i];
        const int u = t + KSH_BT_THREADS * i;
        const int sl = u >> 4, cw = u & 15;
        if (x == 0u || s_flag[sl] != 0.0f) continue;               // nothing legal here / a dense row: not ours
        const uint32_t bit = 1u << (sl & 31);
        while (x) {
            const int k = __ffs((int)x) - 1;
            x &= x - 1u;
            atomicOr(&T[32 * cw + k][sl >> 5], bit);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const int u = t + KSH_BT_THREADS * i;
        const int jl = u >> 4, k = u & 15;
        const int j = 32 * c0 + jl, w = w0 + k;
        if (j < KSH_PA && w < BW) bitsT[(size_t)j * (size_t)BW + (size_t)w] = T[jl][k];
    }
}

// ------------------------------------------------------------------------------------------ weight gradient
// dW[a][:] = sum over the samples b in which action a was legal (ascending b) of dzT[a][b] * feat[b][:], dbias[a] = the
// sum of those dzT -- from the transposed bit matrix, without float atomics, in two kernels so that neither the ~10 000
// actions that were legal nowhere nor the few that were legal almost everywhere set the pace:
//   scan   one wave per four actions counts their legal samples.  Light actions (<= light_max samples: almost all of
//          them) go to the LIGHT list, heavy ones (a move that is legal in most positions) to the HEAVY list together with
//          the mask of their non-empty sample segments (at most 16); one integer atomic per block and list picks the
//          slots (the ORDER of a list is arbitrary, its content is not).  Actions that were legal nowhere get their zero
//          row -- the zero-fill of the gradient -- here.
//   work   blocks of 16 waves.  A heavy action is taken by a whole block: wave w accumulates segment w, the partial rows
//          meet in LDS and are added in segment order.  Light actions are taken by single waves and written directly.
//          Within a wave four 16-lane groups take every fourth sample of a 256-sample window (16 feature rows in flight).
// Every sum has a fixed order, so the result is bit-reproducible and independent of scheduling.
struct KshHeadDwArgs {
    const float *feat;       // [B][K]
    const float *dzT;        // [A][B]
    const uint32_t *bitsT;   // [A][BW]
    float *dW;               // [A][K]
    float *dbias;            // [A] or null
    int B, BW, K, A;
    int SEGW, NSEG;          // dwords per segment, segments per bit row (<= 16; <= 4 for K > 192)
    int light_max;           // actions with at most this many legal samples are taken by one wave
    unsigned *n_items;       // [0] light-list length, [1] "lists corrupt" flag, [2] heavy-list length (zeroed by the caller)
    uint32_t *items;         // [A] light list: action ids
    uint32_t *heavy;         // [A] heavy list: segment mask << 16 | action id
};

#define KSH_DW_WPB 16          // waves per block of the work kernel (= the most segments a heavy action can have)
#define KSH_DW_SCAN_WPB 16
#define KSH_DW_SCAN_APW 4      // actions per wave of the scan kernel (their bit-row loads are issued together)

extern "C" __global__ void __launch_bounds__(64 * KSH_DW_SCAN_WPB) ksh_head_dw_scan_kernel(KshHeadDwArgs a) {
    __shared__ int s_cl[KSH_DW_SCAN_WPB], s_ch[KSH_DW_SCAN_WPB];     // light / heavy actions of each wave
    __shared__ int s_bl, s_bh;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int j0 = (blockIdx.x * KSH_DW_SCAN_WPB + wv) * KSH_DW_SCAN_APW;
    uint32_t m[KSH_DW_SCAN_APW];
    int total[KSH_DW_SCAN_APW];
#pragma unroll
    for (int q = 0; q < KSH_DW_SCAN_APW; q++) { m[q] = 0u; total[q] = 0; }
    for (int d = lane; d < a.BW; d += 64) {              // wave-uniform trip count; 4 independent loads per round
        uint32_t w[KSH_DW_SCAN_APW];
#pragma unroll
        for (int q = 0; q < KSH_DW_SCAN_APW; q++) w[q] = j0 + q < a.A ? a.bitsT[(size_t)(j0 + q) * (size_t)a.BW + (size_t)d] : 0u;
#pragma unroll
        for (int q = 0; q < KSH_DW_SCAN_APW; q++)
            if (w[q] != 0u) { m[q] |= 1u << (d / a.SEGW); total[q] += __popc(w[q]); }
    }
#pragma unroll
    for (int q = 0; q < KSH_DW_SCAN_APW; q++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            m[q] |= (uint32_t)__shfl_xor((int)m[q], o);
            total[q] += __shfl_xor(total[q], o);
        }
    }
    int nl = 0, nh = 0;
#pragma unroll
    for (int q = 0; q < KSH_DW_SCAN_APW; q++) {
        nl += (total[q] > 0 && total[q] <= a.light_max) ? 1 : 0;
        nh += total[q] > a.light_max ? 1 : 0;
    }
    if (lane == 0) { s_cl[wv] = nl; s_ch[wv] = nh; }
    __syncthreads();
    if (threadIdx.x < 2) {                               // ONE atomic per 64 actions and list
        int sum = 0;
        for (int w = 0; w < KSH_DW_SCAN_WPB; w++) sum += threadIdx.x == 0 ? s_cl[w] : s_ch[w];
        const int base = sum ? (int)atomicAdd(a.n_items + (threadIdx.x == 0 ? 0 : 2), (unsigned)sum) : 0;
        if (threadIdx.x == 0) s_bl = base; else s_bh = base;
    }
    __syncthreads();
    int bl = s_bl, bh = s_bh;
    for (int w = 0; w < wv; w++) { bl += s_cl[w]; bh += s_ch[w]; }
    const int nq = a.K >> 2;
    const bool tail = (a.K & 2) != 0;
#pragma unroll
    for (int q = 0; q < KSH_DW_SCAN_APW; q++) {
        const int j = j0 + q;
        if (j >= a.A) break;                              // wave-uniform
        if (total[q] == 0) {
            // legal nowhere in this minibatch: its zero row -- the zero-fill of the gradient -- is written here
            float2 *drow = reinterpret_cast<float2 *>(a.dW + (size_t)j * (size_t)a.K);
            for (int p = lane; p < 2 * nq + (tail ? 1 : 0); p += 64) drow[p] = make_float2(0.0f, 0.0f);
            if (lane == 0 && a.dbias != nullptr) a.dbias[j] = 0.0f;
        } else if (total[q] <= a.light_max) {
            if (lane == 0) a.items[bl] = (uint32_t)j;
            bl++;
        } else {
            if (lane == 0) a.heavy[bh] = (m[q] << 16) | (uint32_t)j;
            bh++;
        }
    }
}

#define KSH_DW_WIN 256   // samples per LDS window of the work kernel

// One wave accumulates dzT[j][b] * feat[b][:] over the samples b of the dwords [d0, d1) of action j's bit row, ascending,
// and leaves the partial row at `out` (global or LDS; K floats) and the sum of the dz in *out_dz (lane 0 writes it).
template <int KMAX>
__device__ __forceinline__ void ksh_dw_accumulate(const KshHeadDwArgs &a, int j, int d0, int d1, int lane, int *rows, float *dzl,
                                                  float *out, float *out_dz) {
    constexpr int QQ = (KMAX / 4 + 63) / 64;
    constexpr int T = 2;                    // rows in flight per group and round (4 / 8 measured slower: profiles/r02_head_kernels_pmc.txt)
    constexpr bool GROUPED = KMAX <= 192;   // 16 lanes x 3 quads cover a feature row: four samples side by side
    const int gs = lane & 15, gg = lane >> 4;
    const int nq = a.K >> 2;
    const bool tail = (a.K & 2) != 0;
    const uint32_t *brow = a.bitsT + (size_t)j * (size_t)a.BW;
    const float *zrow = a.dzT + (size_t)j * (size_t)a.B;
    float4 acc[QQ];
#pragma unroll
    for (int i = 0; i < QQ; i++) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 ag[3];              // GROUPED: this lane's quads gs, gs + 16, gs + 32 of its group's partial row
#pragma unroll
    for (int i = 0; i < 3; i++) ag[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float dzsum = 0.0f;
    for (int c0 = d0; c0 < d1; c0 += 64) {
        // lane = dword of the bit row; an inclusive scan of the popcounts ranks every legal sample of the chunk
        const int d = c0 + lane;
        const uint32_t word = d < d1 ? brow[d] : 0u;
        const int c = __popc(word);
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl(incl, lane >= o ? lane - o : 0);
            if (lane >= o) incl += t;
        }
        const int total = __shfl(incl, 63);
        for (int w0 = 0; w0 < total; w0 += KSH_DW_WIN) {     // windows of 256 samples (one, as a rule), ascending
            {
                int pos = incl - c;
                uint32_t v = word;
                while (v) {
                    const int bit = __ffs((int)v) - 1;
                    v &= v - 1u;
                    if (pos >= w0 && pos < w0 + KSH_DW_WIN) rows[pos - w0] = 32 * d + bit;
                    pos++;
                }
            }
            ksh_head_wave_sync();
            const int cw = total - w0 < KSH_DW_WIN ? total - w0 : KSH_DW_WIN;
            for (int i = lane; i < cw; i += 64) dzl[i] = zrow[rows[i]];   // all of the window's dz loads in flight at once
            ksh_head_wave_sync();
            if constexpr (GROUPED) {
                // four 16-lane groups take every fourth sample of the window: 4 T rows in flight per round, a quarter
                // of the rounds (a wave's time is its chain of rounds)
                for (int e0 = 0; e0 < cw; e0 += 4 * T) {
                    float4 f[T][3];
                    float dz[T];
                    bool on[T];
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        const int e = e0 + 4 * t + gg;
                        on[t] = e < cw;
                        dz[t] = on[t] ? dzl[e] : 0.0f;
                        const float *frow = a.feat + (size_t)rows[on[t] ? e : e0] * (size_t)a.K;
#pragma unroll
                        for (int i = 0; i < 3; i++) f[t][i] = ksh_ld_quad(frow, gs + 16 * i, nq, tail);
                    }
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        if (on[t]) {                             // (no "+ 0 * x" terms in the sums)
                            dzsum += dz[t];
#pragma unroll
                            for (int i = 0; i < 3; i++) {
                                ag[i].x = fmaf(dz[t], f[t][i].x, ag[i].x); ag[i].y = fmaf(dz[t], f[t][i].y, ag[i].y);
                                ag[i].z = fmaf(dz[t], f[t][i].z, ag[i].z); ag[i].w = fmaf(dz[t], f[t][i].w, ag[i].w);
                            }
                        }
                    }
                }
            } else {
                for (int e0 = 0; e0 < cw; e0 += T) {              // wide rows: lanes own quads lane + 64 i, T rows in flight
                    float4 f[T][QQ];
                    float dz[T];
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        const bool on = e0 + t < cw;
                        dz[t] = on ? dzl[e0 + t] : 0.0f;
                        const float *frow = a.feat + (size_t)(on ? rows[e0 + t] : rows[e0]) * (size_t)a.K;
#pragma unroll
                        for (int i = 0; i < QQ; i++) f[t][i] = ksh_ld_quad(frow, lane + 64 * i, nq, tail);
                    }
#pragma unroll
                    for (int t = 0; t < T; t++) {
                        if (e0 + t < cw) {                          // wave-uniform: keeps the sums free of "+ 0 * x" terms
                            dzsum += dz[t];
#pragma unroll
                            for (int i = 0; i < QQ; i++) {
                                acc[i].x = fmaf(dz[t], f[t][i].x, acc[i].x); acc[i].y = fmaf(dz[t], f[t][i].y, acc[i].y);
                                acc[i].z = fmaf(dz[t], f[t][i].z, acc[i].z); acc[i].w = fmaf(dz[t], f[t][i].w, acc[i].w);
                            }
                        }
                    }
                }
            }
            ksh_head_wave_sync();
        }
    }
    float2 *prow = reinterpret_cast<float2 *>(out);
    if constexpr (GROUPED) {
        // the four groups' sums ((g0 + g1) + (g2 + g3): a fixed order); group i then stores quad gs + 16 i
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
                ag[i].x += __shfl_xor(ag[i].x, o); ag[i].y += __shfl_xor(ag[i].y, o);
                ag[i].z += __shfl_xor(ag[i].z, o); ag[i].w += __shfl_xor(ag[i].w, o);
            }
            dzsum += __shfl_xor(dzsum, o);
        }
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int q = gs + 16 * i;
            if (gg == i) {
                if (q < nq || (q == nq && tail)) prow[2 * q] = make_float2(ag[i].x, ag[i].y);
                if (q < nq) prow[2 * q + 1] = make_float2(ag[i].z, ag[i].w);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < QQ; i++) {
            const int q = lane + 64 * i;
            if (q < nq || (q == nq && tail)) prow[2 * q] = make_float2(acc[i].x, acc[i].y);
            if (q < nq) prow[2 * q + 1] = make_float2(acc[i].z, acc[i].w);
        }
    }
    if (lane == 0) *out_dz = dzsum;
}

template <int KMAX>
__global__ void __launch_bounds__(64 * KSH_DW_WPB) ksh_head_dw_work_kernel(KshHeadDwArgs a) {
    constexpr int NRED = KMAX <= 192 ? 16 : 4;            // segments of a heavy action (head_ws: maxseg)
    __shared__ int s_rows[KSH_DW_WPB][KSH_DW_WIN];
    __shared__ float s_dz[KSH_DW_WPB][KSH_DW_WIN];
    __shared__ __attribute__((aligned(16))) float s_red[NRED][KMAX];
    __shared__ float s_rdz[NRED];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int n_light = (int)a.n_items[0], n_heavy = (int)a.n_items[2];
    // Lists that cannot be right (a counter beyond the number of actions: only possible if the scratch was not cleared or
    // was overwritten) must neither send this kernel outside its buffers nor pass silently: nothing is processed and row 0
    // of the gradient -- which no other block touches then -- is poisoned with NaN.
    if (n_light < 0 || n_light > a.A || n_heavy < 0 || n_heavy > a.A || a.NSEG > NRED) {
        if (blockIdx.x == 0) {
            for (int k = (int)threadIdx.x; k < a.K; k += (int)blockDim.x) a.dW[k] = NAN;
            if (threadIdx.x == 0) { a.n_items[1] = 1u; if (a.dbias != nullptr) a.dbias[0] = NAN; }
        }
        return;
    }
    bool bad_item = false;
    // heavy actions: one per block, wave w = segment w, partial rows added in segment order
    for (int h = (int)blockIdx.x; h < n_heavy; h += (int)gridDim.x) {       // block-uniform
        const uint32_t hv = a.heavy[h];
        const int j = (int)(hv & 0xFFFFu);
        const uint32_t m = hv >> 16;
        if (j >= a.A || (m >> a.NSEG) != 0u) { bad_item = true; continue; }     // (block-uniform)
        if (wv < a.NSEG && ((m >> wv) & 1u)) {
            const int d0 = wv * a.SEGW, d1 = (d0 + a.SEGW < a.BW) ? d0 + a.SEGW : a.BW;
            ksh_dw_accumulate<KMAX>(a, j, d0, d1, lane, s_rows[wv], s_dz[wv], s_red[wv], &s_rdz[wv]);
        }
        __syncthreads();
        for (int k = (int)threadIdx.x; k < a.K; k += (int)blockDim.x) {
            float v = 0.0f;
            for (int sg = 0; sg < a.NSEG; sg++)
                if ((m >> sg) & 1u) v += s_red[sg][k];
            a.dW[(size_t)j * (size_t)a.K + (size_t)k] = v;
        }
        if (threadIdx.x == 0 && a.dbias != nullptr) {
            float v = 0.0f;
            for (int sg = 0; sg < a.NSEG; sg++)
                if ((m >> sg) & 1u) v += s_rdz[sg];
            a.dbias[j] = v;
        }
        __syncthreads();
    }
    // light actions: one per wave, the whole bit row, straight into the gradient
    float dummy;
    // (dealt out from the LAST block backwards: the heavy actions went to the first blocks)
    for (int it = ((int)gridDim.x - 1 - (int)blockIdx.x) * KSH_DW_WPB + wv; it < n_light; it += (int)gridDim.x * KSH_DW_WPB) {     // wave-uniform
        const int j = (int)a.items[it];
        if (j < 0 || j >= a.A) { bad_item = true; continue; }
        ksh_dw_accumulate<KMAX>(a, j, 0, a.BW, lane, s_rows[wv], s_dz[wv], a.dW + (size_t)j * (size_t)a.K,
                                a.dbias != nullptr ? a.dbias + j : &dummy);
    }
    if (bad_item) {     // an id beyond the table (memory corruption): say so as well as a kernel without a last word can
        if (lane == 0) a.n_items[1] = 1u;
        for (int k = lane; k < a.K; k += 64) a.dW[k] = NAN;
        if (lane == 0 && a.dbias != nullptr) a.dbias[0] = NAN;
    }
}

// ------------------------------------------------------------------------------------------ byte mask -> packed bits
// PolicyOutputMapper.get_legal_mask's bool[13527] rows (utils/utils.py:310-336) as the 424-dword rows the head kernels read.
extern "C" __global__ void __launch_bounds__(KSH_ROW_THREADS) ksh_pack_mask_kernel(const uint8_t *__restrict__ mask, int B,
                                                                                   uint32_t *__restrict__ bits) {
    __shared__ KshRowLds R;
    const int b = blockIdx.x;
    ksh_build_bits(R, mask + (size_t)b * KSH_PA, (int)threadIdx.x);
    for (int i = (int)threadIdx.x; i < KSH_PBITROW; i += KSH_ROW_THREADS) {
        uint32_t v = R.bits[i];
        if (i == 422) v &= 0x007FFFFFu;
        if (i == 423) v = 0u;
        bits[(size_t)b * KSH_PBITROW + i] = v;
    }
}

// ------------------------------------------------------------------------------------------ dense rows: MFMA GEMM
// C[M][N] (=, +=) A[M][Kd] * B[Kd][N] on v_mfma_f32_32x32x2_f32 (float32 in, float32 accumulate, k ascending).  One body
// (ksh_head_gemm_body, below) instantiated per product with its own block tile; K-step 16 staged through LDS as As[k][m] /
// Bs[k][n] (a lane reads A[m = lane & 31][k = lane >> 5] and B[k][n = lane & 31]: consecutive lanes, consecutive LDS words).
// Operands are addressed through (row, column) strides; the three products of the dense path:
//   mode 0  logits[r][a]  = sum_k feat[r][k] W[a][k] + bias[a]      rows gated: only tiles / rows flagged as dense
//   mode 1  dfeat[r][k]   = sum_a dl[r][a] W[a][k]                  rows gated likewise
//   mode 2  dW[a][k]     += sum_r dl[r][a] feat[r][k], dbias[a] += sum_r dl[r][a]   (reduction over the flagged rows only;
//           column N-1 of the B operand is the constant 1 that produces dbias)
// `flags[r * 8]` != 0 marks a dense row (the loss kernels' partial[b][6]); *n_over == 0 ends every block at once.
typedef float ksh_f32x16 __attribute__((vector_size(64)));

struct KshGemmArgs {
    const float *A;
    long long sAm, sAk;
    const float *Bm;
    long long sBk, sBn;
    float *C;
    long long sCm;
    int M, N, Kd;
    const float *bias;
    const float *flags;       // partial + 6 (stride 8 floats per row)
    const unsigned *n_over;
    int mode;
    float *C2;                // mode 2: dbias
    // split-K (mode 1 only: dfeat = dl W has a 13 527-long reduction but only B x K outputs, i.e. a few dozen tiles):
    // blockIdx.z takes the reduction steps [z * kchunk, (z + 1) * kchunk) and writes its partial tile to
    // Cpart[z][M][N]; ksh_head_splitk_reduce_kernel adds the slices in order.  ksplit <= 1: no split.
    int ksplit, kchunk;
    float *Cpart;
};

#define KSH_GEMM_BK 16        // granularity of a split-K slice (host side); the logits product's K-step
#define KSH_GEMM_BK_LONG 16   // K-step of the two products with long reductions (32 measured: dfeat 67 -> 73 us, dW 78 -> 102 us: twice the quads in flight spill)
// block tiles of the three products (rows x columns): see ksh_head_gemm_body
#ifndef KSH_GEMM_M0
#define KSH_GEMM_M0 64    // mode 0 (logits):  B x 13527
#endif
#ifndef KSH_GEMM_N0
#define KSH_GEMM_N0 64    // (tile grid measured: 64x64 69.5 us, 128x64 72.9, 64x128 73-76, 128x128 84, 64x256 87)
#endif
#ifndef KSH_GEMM_M1
#define KSH_GEMM_M1 128   // mode 1 (dfeat):   B x K, reduction over the 13 527 actions split KSH_HEAD_KSPLIT ways
#endif
#ifndef KSH_GEMM_N1
#define KSH_GEMM_N1 192
#endif
#ifndef KSH_GEMM_M2
#define KSH_GEMM_M2 64    // mode 2 (dW):      13527 x (K + 1), reduction over the samples
#endif
#ifndef KSH_GEMM_N2
#define KSH_GEMM_N2 192
#endif

struct __attribute__((packed, aligned(4))) ksh_f4a4 {   // four floats at a dword-aligned address (gfx9: global_load_dwordx4 needs no more)
    float x, y, z, w;
};
// Floats [o0, o0 + 4) of the tensor at `base` (offsets in floats), of which only those below `run_end` (the end of the row /
// reduction range they belong to) count: the others read as zero, and `ok == false` gives four zeros.  ONE unconditional
// 16-byte load and bit masks, no branch (the first version had an `if` per case: the compiler turned the tile loader into 297
// scalar loads, each behind its own `s_waitcnt vmcnt(0)`).  The load never leaves the tensor (`tensor_end` floats, >= 4): a
// quad that would is fetched as the tensor's last quad and shifted into place.
__device__ __forceinline__ ksh_f4a4 ksh_gemm_ld4_raw(const float *base, long long o0, long long tensor_end) {
    const long long oc = o0 < tensor_end - 4 ? o0 : tensor_end - 4;
    return *reinterpret_cast<const ksh_f4a4 *>(base + oc);
}
// (the masking half, applied when the quad is written to LDS -- AFTER the MFMAs of the step the load overlapped)
__device__ __forceinline__ float4 ksh_gemm_ld4_fix(ksh_f4a4 v, long long o0, long long run_end, long long tensor_end, bool ok) {
    const long long oc = o0 < tensor_end - 4 ? o0 : tensor_end - 4;
    const long long dl = o0 - oc;
    const int d = dl > 4 ? 4 : (int)dl;                     // 0: in place; 1..3: shifted; 4: nothing of the quad is wanted
    const uint32_t x = __float_as_uint(v.x), y = __float_as_uint(v.y), z = __float_as_uint(v.z), w = __float_as_uint(v.w);
    const uint32_t m0 = d == 0 ? 0xFFFFFFFFu : 0u, m1 = d == 1 ? 0xFFFFFFFFu : 0u, m2 = d == 2 ? 0xFFFFFFFFu : 0u,
                   m3 = d == 3 ? 0xFFFFFFFFu : 0u;
    const long long left = ok ? run_end - o0 : 0;           // how many of the four floats count
    const uint32_t k0 = left > 0 ? 0xFFFFFFFFu : 0u, k1 = left > 1 ? 0xFFFFFFFFu : 0u, k2 = left > 2 ? 0xFFFFFFFFu : 0u,
                   k3 = left > 3 ? 0xFFFFFFFFu : 0u;
    return make_float4(ksh_u2f(((x & m0) | (y & m1) | (z & m2) | (w & m3)) & k0), ksh_u2f(((y & m0) | (z & m1) | (w & m2)) & k1),
                       ksh_u2f(((z & m0) | (w & m1)) & k2), ksh_u2f((w & m0) & k3));
}

// Round 3: the tile kernel rewritten for the matrix cores' sake -- BM x BN block tile (128x128 / 128x192 / 64x192), four waves
// as 2 x 2 with (BM/64) x (BN/64) accumulator tiles of 32x32 each, K-step 16, DOUBLE-BUFFERED LDS (the next step's global loads
// are in flight during this step's MFMAs, one barrier per step), 16-byte global loads along whichever index of an operand is
// contiguous (AKF / BKF: the reduction index k is; else the row / column index is).  LDS holds As[k][m] / Bs[k][n] with a
// 4-float pad: an MFMA operand read is 32 consecutive words per k (conflict-free), an m- or n-contiguous quad is one 16-byte
// write, a k-contiguous quad four scalar writes.
template <int MODE, int BM, int BN, int BK, int AKF, int BKF>
__device__ __forceinline__ void ksh_head_gemm_body(const KshGemmArgs &g) {
    // row pads: a k-contiguous operand is written to LDS as scalars, lane = (row i = c / 4, k-quad c % 4): a row stride of
    // 2 (mod 32) puts the 32 lanes of a half-wave (8 rows x 4 quads) on 32 different banks (4 (mod 32): 2-way conflicts,
    // 1.8 M conflict cycles per logits launch); an m- / n-contiguous operand is written as 16-byte quads and keeps the pad of 4
    constexpr int LDA = BM + (AKF ? 2 : 4), LDB = BN + (BKF ? 2 : 4);
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int CA = BM * BK / 4 / 256, CB = BN * BK / 4 / 256;     // 16-byte chunks per thread and K-step
    static_assert(BM % 64 == 0 && BN % 64 == 0 && CA >= 1 && CB >= 1, "tile shape");
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    __shared__ int s_any;
    __shared__ int s_nact, s_wtot[4];
    __shared__ unsigned short s_steps[512];                 // mode 2: the K-steps that hold a flagged sample, ascending
    const int t = (int)threadIdx.x, lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1, r = lane & 31, h = lane >> 5;
    const int m0 = (int)blockIdx.x * BM, n0 = (int)blockIdx.y * BN;
    if (MODE != 2) {                                     // any flagged row in this row tile?
        if (t == 0) s_any = 0;
        __syncthreads();
        if (t < BM && m0 + t < g.M && g.flags[(size_t)(m0 + t) * 8] != 0.0f) s_any = 1;
        __syncthreads();
        if (!s_any) return;
    }
    const bool split = g.ksplit > 1;
    const int kbeg = split ? (int)blockIdx.z * g.kchunk : 0;
    const int kend = split ? ((kbeg + g.kchunk < g.Kd) ? kbeg + g.kchunk : g.Kd) : g.Kd;
    const int total_steps = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    ksh_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.0f;
    // ---- global -> registers for the K-step that starts at k0 (rows / steps beyond the operand are clamped to a valid row and
    // masked to zero, so every lane issues the same unconditional loads)
    const long long endA = AKF ? (long long)(g.M - 1) * g.sAm + g.Kd : (long long)(g.Kd - 1) * g.sAk + g.M;
    const int ncolB = MODE == 2 ? g.N - 1 : g.N;           // columns B really has (mode 2: column N-1 is the constant 1)
    const long long endB = BKF ? (long long)(g.N - 1) * g.sBn + g.Kd : (long long)(g.Kd - 1) * g.sBk + ncolB;
    // operand quad q of this thread: where it lives (float offset), where its run ends, whether it counts at all
    auto a_quad = [&](int q, int k0, long long &o0, long long &run_end, bool &ok) {
        const int c = t + 256 * q;
        if (AKF) {                                         // rows of A are contiguous in k: chunk = (row i, four k)
            const int i = c / (BK / 4), kq = (c - i * (BK / 4)) * 4, m = m0 + i;
            const long long row = (long long)(m < g.M ? m : g.M - 1) * g.sAm;
            o0 = row + k0 + kq; run_end = row + kend; ok = m < g.M;
        } else {                                           // A is contiguous in m: chunk = (one k, four rows)
            const int kk = c / (BM / 4), mq = (c - kk * (BM / 4)) * 4, k = k0 + kk;
            const int kc = k < kend ? k : kend - 1;
            const long long row = (long long)kc * g.sAk;
            o0 = row + m0 + mq; run_end = row + g.M; ok = k < kend;
        }
    };
    auto b_quad = [&](int q, int k0, long long &o0, long long &run_end, bool &ok) {
        const int c = t + 256 * q;
        if (BKF) {                                         // B[k][n] with k contiguous (W rows in mode 0)
            const int i = c / (BK / 4), kq = (c - i * (BK / 4)) * 4, n = n0 + i;
            const long long row = (long long)(n < g.N ? n : g.N - 1) * g.sBn;
            o0 = row + k0 + kq; run_end = row + kend; ok = n < g.N;
        } else {                                           // contiguous in n
            const int kk = c / (BN / 4), nq = (c - kk * (BN / 4)) * 4, k = k0 + kk;
            const int kc = k < kend ? k : kend - 1;
            const long long row = (long long)kc * g.sBk;
            o0 = row + n0 + nq; run_end = row + ncolB; ok = k < kend;
        }
    };
    // two register sets for operand quads in flight: a step's loads are issued TWO steps before its MFMAs (one step did not
    // cover the latency of memory: with one wave per SIMD the dW product spent 2.6 k of 4.1 k cycles per step waiting)
    ksh_f4a4 ra0[CA], rb0[CB], ra1[CA], rb1[CB], rat[CA], rbt[CB];   // (t: the range's last step, loaded first, used last)
    float fa0[CA], fa1[CA], fat[CA];                                // mode 2: the flag of the sample row an A quad belongs to
    // What does not change from one K-step to the next is computed ONCE per thread: the address of each of its quads at
    // k = kbeg (rows clamped to the operand), and four keep-bits per quad (row / column validity).  An INTERIOR step (all of
    // its k below kend - the block-uniform case for every step but the last) is then: pointer + k0 * stride, one 16-byte
    // load, four ANDs at staging time.  Only the last step of a reduction range goes through the general quad logic above
    // (k tail, end-of-tensor clamp).  (PMC, first tiled version: 300 VALU instructions per K-step and wave for this
    // bookkeeping against 32-48 MFMAs -- the matrix pipe was 30-40 % busy.)
    const float *pa[CA], *pb[CB];
    uint32_t keepa[CA], keepb[CB];
#pragma unroll
    for (int q = 0; q < CA; q++) {
        long long o0, re; bool ok;
        a_quad(q, kbeg, o0, re, ok);
        pa[q] = g.A + o0;
        if (AKF) keepa[q] = ok ? 15u : 0u;
        else {
            const long long left = re - o0;                // valid rows m in this quad
            keepa[q] = left >= 4 ? 15u : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
            // a quad that lies entirely beyond the operand's columns (tile wider than the matrix) is masked to zero anyway:
            // it is loaded from the START of its k-row instead, or the interior steps near the operand's last rows would
            // read past the end of the tensor (found by the emulator's AddressSanitizer build, K = 10: up to 764 bytes)
            if (left <= 0) pa[q] = g.A + (o0 - (m0 + (t + 256 * q - ((t + 256 * q) / (BM / 4)) * (BM / 4)) * 4));
        }
    }
#pragma unroll
    for (int q = 0; q < CB; q++) {
        long long o0, re; bool ok;
        b_quad(q, kbeg, o0, re, ok);
        pb[q] = g.Bm + o0;
        if (BKF) keepb[q] = ok ? 15u : 0u;
        else {
            const long long left = re - o0;
            keepb[q] = left >= 4 ? 15u : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
            if (left <= 0) pb[q] = g.Bm + (o0 - (n0 + (t + 256 * q - ((t + 256 * q) / (BN / 4)) * (BN / 4)) * 4));   // (as above)
        }
    }
    const long long ska = AKF ? 1 : g.sAk, skb = BKF ? 1 : g.sBk;   // float stride of one k
    auto keep4 = [](ksh_f4a4 v, uint32_t keep) {
        const uint32_t m0 = 0u - (keep & 1u), m1 = 0u - ((keep >> 1) & 1u), m2 = 0u - ((keep >> 2) & 1u), m3 = 0u - ((keep >> 3) & 1u);
        return make_float4(ksh_u2f(__float_as_uint(v.x) & m0), ksh_u2f(__float_as_uint(v.y) & m1), ksh_u2f(__float_as_uint(v.z) & m2),
                           ksh_u2f(__float_as_uint(v.w) & m3));
    };
    // ---- interior K-steps: global -> registers (pointer + k0 * stride, one 16-byte load per quad) ...
    auto load_fast = [&](ksh_f4a4 (&ra)[CA], ksh_f4a4 (&rb)[CB], float (&fa)[CA], int k0) {
        const long long dk = (long long)(k0 - kbeg);
#pragma unroll
        for (int q = 0; q < CA; q++) {
            ra[q] = *reinterpret_cast<const ksh_f4a4 *>(pa[q] + dk * ska);
            if (MODE == 2) fa[q] = g.flags[(size_t)(k0 + (t + 256 * q) / (BM / 4)) * 8];
        }
#pragma unroll
        for (int q = 0; q < CB; q++) rb[q] = *reinterpret_cast<const ksh_f4a4 *>(pb[q] + dk * skb);
    };
    // ... and registers -> LDS buffer `buf` (four ANDs per quad; first touched AFTER the MFMAs the loads overlapped)
    auto stage_fast = [&](ksh_f4a4 (&ra)[CA], ksh_f4a4 (&rb)[CB], float (&fa)[CA], int buf) {
#pragma unroll
        for (int q = 0; q < CA; q++) {
            const int c = t + 256 * q;
            const float4 v = keep4(ra[q], (MODE == 2 && fa[q] == 0.0f) ? 0u : keepa[q]);
            if (AKF) {
                const int i = c / (BK / 4), kq = (c - i * (BK / 4)) * 4;
                As[buf][kq][i] = v.x; As[buf][kq + 1][i] = v.y; As[buf][kq + 2][i] = v.z; As[buf][kq + 3][i] = v.w;
            } else {
                const int kk = c / (BM / 4), mq = (c - kk * (BM / 4)) * 4;
                *reinterpret_cast<float4 *>(&As[buf][kk][mq]) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < CB; q++) {
            const int c = t + 256 * q;
            float4 v = keep4(rb[q], keepb[q]);
            if (BKF) {
                const int i = c / (BK / 4), kq = (c - i * (BK / 4)) * 4;
                Bs[buf][kq][i] = v.x; Bs[buf][kq + 1][i] = v.y; Bs[buf][kq + 2][i] = v.z; Bs[buf][kq + 3][i] = v.w;
            } else {
                const int kk = c / (BN / 4), nq = (c - kk * (BN / 4)) * 4, n = n0 + nq;
                if (MODE == 2) {                           // column N-1 is the constant 1 (-> dbias)
                    v.x = n == g.N - 1 ? 1.0f : v.x; v.y = n + 1 == g.N - 1 ? 1.0f : v.y;
                    v.z = n + 2 == g.N - 1 ? 1.0f : v.z; v.w = n + 3 == g.N - 1 ? 1.0f : v.w;
                }
                *reinterpret_cast<float4 *>(&Bs[buf][kk][nq]) = v;
            }
        }
    };
    // ---- the LAST step of a reduction range (k tail, end-of-tensor clamp): the general quad logic, once per block
    auto load_gen = [&](ksh_f4a4 (&ra)[CA], ksh_f4a4 (&rb)[CB], float (&fa)[CA], int k0) {
#pragma unroll
        for (int q = 0; q < CA; q++) {
            long long o0, re; bool ok;
            a_quad(q, k0, o0, re, ok);
            ra[q] = ksh_gemm_ld4_raw(g.A, o0, endA);
            if (MODE == 2) {
                const int k = k0 + (t + 256 * q) / (BM / 4);
                fa[q] = g.flags[(size_t)(k < kend ? k : kend - 1) * 8];
            }
        }
#pragma unroll
        for (int q = 0; q < CB; q++) {
            long long o0, re; bool ok;
            b_quad(q, k0, o0, re, ok);
            rb[q] = ksh_gemm_ld4_raw(g.Bm, o0, endB);
        }
    };
    auto stage_gen = [&](ksh_f4a4 (&ra)[CA], ksh_f4a4 (&rb)[CB], float (&fa)[CA], int buf, int k0) {
#pragma unroll
        for (int q = 0; q < CA; q++) {
            const int c = t + 256 * q;
            long long o0, re; bool ok;
            a_quad(q, k0, o0, re, ok);
            if (MODE == 2) ok = ok && fa[q] != 0.0f;
            const float4 v = ksh_gemm_ld4_fix(ra[q], o0, re, endA, ok);
            if (AKF) {
                const int i = c / (BK / 4), kq = (c - i * (BK / 4)) * 4;
                As[buf][kq][i] = v.x; As[buf][kq + 1][i] = v.y; As[buf][kq + 2][i] = v.z; As[buf][kq + 3][i] = v.w;
            } else {
                const int kk = c / (BM / 4), mq = (c - kk * (BM / 4)) * 4;
                *reinterpret_cast<float4 *>(&As[buf][kk][mq]) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < CB; q++) {
            const int c = t + 256 * q;
            long long o0, re; bool ok;
            b_quad(q, k0, o0, re, ok);
            float4 v = ksh_gemm_ld4_fix(rb[q], o0, re, endB, ok);
            if (BKF) {
                const int i = c / (BK / 4), kq = (c - i * (BK / 4)) * 4;
                Bs[buf][kq][i] = v.x; Bs[buf][kq + 1][i] = v.y; Bs[buf][kq + 2][i] = v.z; Bs[buf][kq + 3][i] = v.w;
            } else {
                const int kk = c / (BN / 4), nq = (c - kk * (BN / 4)) * 4, n = n0 + nq;
                if (MODE == 2) {
                    const float one = ok ? 1.0f : 0.0f;
                    v.x = n == g.N - 1 ? one : v.x; v.y = n + 1 == g.N - 1 ? one : v.y;
                    v.z = n + 2 == g.N - 1 ? one : v.z; v.w = n + 3 == g.N - 1 ? one : v.w;
                }
                *reinterpret_cast<float4 *>(&Bs[buf][kk][nq]) = v;
            }
        }
    };
    bool contributed = false;
    // the K-steps are walked in windows of 512 (the capacity of the step list; one window unless mode 2 reduces over more
    // than 8192 samples)
    for (int win = 0; win < total_steps; win += 512) {
        int nsteps = total_steps - win < 512 ? total_steps - win : 512;
        if (MODE == 2) {
            // the reduction runs over sample rows: only the steps with a flagged (dense) one take part, in ascending order
            // (the order of the float32 sums is fixed).  Ordered compaction: ballot + prefix inside a wave, wave totals
            // through LDS.
            __syncthreads();
            if (t == 0) s_nact = 0;
            __syncthreads();
            for (int base = 0; base < nsteps; base += 256) {
                const int sidx = base + t;
                bool any = false;
                if (sidx < nsteps)
                    for (int j = 0; j < BK; j++) {
                        const int k = kbeg + (win + sidx) * BK + j;
                        if (k < kend && g.flags[(size_t)k * 8] != 0.0f) any = true;
                    }
                const unsigned long long bal = __ballot(any);
                if (lane == 0) s_wtot[w] = __popcll(bal);
                __syncthreads();
                int off = s_nact;
                for (int q = 0; q < w; q++) off += s_wtot[q];
                if (any) s_steps[off + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)sidx;
                __syncthreads();
                if (t == 0) s_nact += s_wtot[0] + s_wtot[1] + s_wtot[2] + s_wtot[3];
                __syncthreads();
            }
            nsteps = s_nact;
            if (nsteps == 0) continue;                     // block-uniform: nothing to add from this window
        }
        contributed = true;
        auto step_k0 = [&](int sidx) { return kbeg + (win + (MODE == 2 ? (int)s_steps[sidx] : sidx)) * BK; };
        // one K-step out of LDS buffer `buf`: operand fragments are read one group (two k-pairs) AHEAD of the MFMAs that use
        // them, into registers of their own
        auto mfma_step = [&](int buf) {
            constexpr int NG = BK / 4;                             // groups of two k-pairs
            float av[2][2][TM], bv[2][2][TN];
            auto frags = [&](int gi, int slot) {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    const int kk = 4 * gi + 2 * p + h;
#pragma unroll
                    for (int i = 0; i < TM; i++) av[slot][p][i] = As[buf][kk][wm * WM + 32 * i + r];
#pragma unroll
                    for (int j = 0; j < TN; j++) bv[slot][p][j] = Bs[buf][kk][wn * WN + 32 * j + r];
                }
            };
            frags(0, 0);
#pragma unroll
            for (int gi = 0; gi < NG; gi++) {
                if (gi + 1 < NG) frags(gi + 1, (gi + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);                 // (keeps the reads above ahead of the MFMAs below)
#pragma unroll
                for (int p = 0; p < 2; p++)
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int j = 0; j < TN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[gi & 1][p][i], bv[gi & 1][p][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // schedule: the range's last step (the only one that may need the general quad logic) is loaded first and multiplied
        // last; the interior steps run two deep: step s computes from LDS buffer s & 1 while the quads of step s+1 sit in
        // register set (s+1) & 1 and the loads of step s+2 are issued into set s & 1 (whose step is already in LDS)
        const bool has_tail = step_k0(nsteps - 1) + BK >= kend;    // block-uniform
        const int nfast = nsteps - (has_tail ? 1 : 0);
        if (has_tail) load_gen(rat, rbt, fat, step_k0(nsteps - 1));
        if (nfast > 0) {
            load_fast(ra0, rb0, fa0, step_k0(0));
            stage_fast(ra0, rb0, fa0, 0);
            if (nfast > 1) load_fast(ra1, rb1, fa1, step_k0(1));
            __syncthreads();
            for (int sidx = 0; sidx < nfast; sidx += 2) {
                if (sidx + 2 < nfast) load_fast(ra0, rb0, fa0, step_k0(sidx + 2));
                mfma_step(0);
                __builtin_amdgcn_sched_barrier(0);                 // the quads of step sidx+1 are first touched below this line
                if (sidx + 1 < nfast) stage_fast(ra1, rb1, fa1, 1);    // (buffer 1 was last read in step sidx-1)
                __syncthreads();
                if (sidx + 1 >= nfast) break;
                if (sidx + 3 < nfast) load_fast(ra1, rb1, fa1, step_k0(sidx + 3));
                mfma_step(1);
                __builtin_amdgcn_sched_barrier(0);
                if (sidx + 2 < nfast) stage_fast(ra0, rb0, fa0, 0);
                __syncthreads();
            }
        }
        if (has_tail) {
            stage_gen(rat, rbt, fat, 0, step_k0(nsteps - 1));      // (every wave has left the buffers: the loop ends with a barrier)
            __syncthreads();
            mfma_step(0);
            __syncthreads();
        }
    }
    if (MODE == 2 && !contributed && !split) return;       // (a split slice always writes its partial tile: zeros when it added nothing)
    // ---- epilogue: register q of a 32x32 tile is row (q & 3) + 8 (q >> 2) + 4 h, column r.  Everything an element needs
    // from memory (its row's flag, its column's bias, the old value it is added to) is loaded in BATCHES before the stores of
    // a tile: a load per element in the store loop is 64 round trips to memory per lane (the first version spent more time
    // there than in its MFMAs).
#pragma unroll
    for (int i = 0; i < TM; i++) {
        const int mb = m0 + wm * WM + 32 * i + 4 * h;
        float fl[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int m = mb + (q & 3) + 8 * (q >> 2);
            fl[q] = g.flags[(size_t)(m < g.M ? m : g.M - 1) * 8];   // (unconditional: 16 loads in flight, one wait)
            if (split || MODE == 2) fl[q] = 1.0f;
        }
        uint32_t okrow = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) okrow |= ((mb + (q & 3) + 8 * (q >> 2) < g.M && fl[q] != 0.0f) ? 1u : 0u) << q;
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * WN + 32 * j + r;
            const bool nok = n < g.N;
            if (split) {            // partial tile of this reduction slice (every row of a tile that holds a flagged row)
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int m = mb + (q & 3) + 8 * (q >> 2);
                    if (nok && ((okrow >> q) & 1u))
                        g.Cpart[((size_t)blockIdx.z * (size_t)g.M + (size_t)m) * (size_t)g.N + (size_t)n] = acc[i][j][q];
                }
            } else if (MODE == 2) {
                // += : the old values are loaded four rows at a time (all sixteen at once cost 249 VGPRs: one wave per SIMD)
                const bool last = n == g.N - 1;          // the constant-1 column: dbias
                float *cbase = last ? g.C2 : g.C + n;
                const long long cstride = last ? 1 : g.sCm;
                const bool cok = nok && cbase != nullptr;
#pragma unroll
                for (int qg = 0; qg < 4; qg++) {
                    float old[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int m = mb + u + 8 * qg;
                        old[u] = (cok && m < g.M) ? cbase[(long long)m * cstride] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int m = mb + u + 8 * qg;
                        if (cok && m < g.M) cbase[(long long)m * cstride] = old[u] + acc[i][j][4 * qg + u];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                const float bn = (nok && g.bias != nullptr) ? g.bias[n] : 0.0f;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int m = mb + (q & 3) + 8 * (q >> 2);
                    if (nok && ((okrow >> q) & 1u)) g.C[(long long)m * g.sCm + n] = acc[i][j][q] + bn;
                }
            }
            __builtin_amdgcn_sched_barrier(0);             // one tile's batch of loads at a time (all tiles at once: 249 VGPRs)
        }
    }
}

// C[M][N] (=, +=) A[M][Kd] * B[Kd][N]; one kernel per product of the dense path (each carries only its own LDS tiles).
// Grids: logits (ceil(M / 128), ceil(N / 128)); dfeat (ceil(M / 128), ceil(N / 192), ksplit); dw (ceil(M / 64), ceil(N / 192)).
// *n_over == 0 (no dense row in this minibatch: the production case) ends every block at once.
extern "C" __global__ void __launch_bounds__(256) ksh_head_gemm_logits_kernel(KshGemmArgs g) {     // mode 0
    if (*g.n_over == 0u) return;
    ksh_head_gemm_body<0, KSH_GEMM_M0, KSH_GEMM_N0, 16, 1, 1>(g);     // K = 162: 11 steps of 16 (6 of 32 would pad 18 %)
}
extern "C" __global__ void __launch_bounds__(256) ksh_head_gemm_dfeat_kernel(KshGemmArgs g) {      // mode 1
    if (*g.n_over == 0u) return;
    ksh_head_gemm_body<1, KSH_GEMM_M1, KSH_GEMM_N1, KSH_GEMM_BK_LONG, 1, 0>(g);
}
extern "C" __global__ void __launch_bounds__(256) ksh_head_gemm_dw_kernel(KshGemmArgs g) {         // mode 2
    if (*g.n_over == 0u) return;
    ksh_head_gemm_body<2, KSH_GEMM_M2, KSH_GEMM_N2, KSH_GEMM_BK_LONG, 0, 0>(g);
}

// dW[m][n] += sum over the ksplit slices (ascending) of Cpart[z][m][n]; column N-1 goes to dbias.  Four rows per block.
extern "C" __global__ void __launch_bounds__(256) ksh_head_splitk_reduce_dw_kernel(KshGemmArgs g) {
    if (*g.n_over == 0u) return;
    const int m = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (m >= g.M) return;
    for (int n = (int)(threadIdx.x & 63); n < g.N; n += 64) {
        float v = 0.0f;
        for (int z = 0; z < g.ksplit; z++) v += g.Cpart[((size_t)z * (size_t)g.M + (size_t)m) * (size_t)g.N + (size_t)n];
        if (n == g.N - 1) { if (g.C2 != nullptr) g.C2[m] += v; }
        else g.C[(long long)m * g.sCm + n] += v;
    }
}

// C[m][n] = sum over the ksplit slices (ascending) of Cpart[z][m][n], for the flagged rows only.
extern "C" __global__ void __launch_bounds__(256) ksh_head_splitk_reduce_kernel(KshGemmArgs g) {
    if (*g.n_over == 0u) return;
    const int m = (int)blockIdx.x;
    if (g.flags[(size_t)m * 8] == 0.0f) return;           // block-uniform
    for (int n = (int)threadIdx.x; n < g.N; n += (int)blockDim.x) {
        // the slices are added in ascending order (fixed float32 sum), but LOADED eight at a time: a load per addition was 32
        // dependent round trips to memory per element (12 us for 21 MB)
        float v = 0.0f;
        for (int z0 = 0; z0 < g.ksplit; z0 += 8) {
            float p[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                p[u] = z0 + u < g.ksplit ? g.Cpart[((size_t)(z0 + u) * (size_t)g.M + (size_t)m) * (size_t)g.N + (size_t)n] : 0.0f;
#pragma unroll
            for (int u = 0; u < 8; u++) v += p[u];
        }
        g.C[(long long)m * g.sCm + n] = v;
    }
}
