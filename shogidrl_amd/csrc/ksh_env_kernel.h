// ksh_env_kernel.h -- the batched Shogi environment kernel for gfx950 (CDNA4), one game per wavefront.
//
// What it replaces (reference = /root/reference/keisei/):
//   shogi/shogi_rules_logic.py:486-635  generate_all_legal_moves (+ :82-208 pseudo-legal, :234-272
//       attack test, :275-359 uchi-fu-zume, :382-483 promotion / drop rules)
//   shogi/shogi_move_execution.py:24-156, shogi/shogi_game.py:574-660  make_move
//   shogi/shogi_game.py:408-450 termination, shogi_rules_logic.py:638-695 sennichite
//   shogi/shogi_game_io.py:434-539 46-plane observation, utils/utils.py:310-336 legal mask
//
// The reference simulates every candidate move on a deep copy and regenerates every enemy piece's
// moves to test for check.  This kernel computes the same legal SET with a different algorithm:
// attack bitboards (three 27-bit words) from LDS-staged ray / step tables, an enemy attack map with
// the king lifted off the board (king moves), checkers + pin rays (all other moves), a drop-square
// set, and one nested reply search for the single possible uchi-fu-zume square.
//
// Mapping: 64 lanes cooperate on one game.  Lane = piece for attack generation and move emission,
// lane = square for occupancy ballots / drops, lane = ray for the pin scan, and lane = 16-byte
// chunk for the coalesced observation / mask stores.  All cross-lane traffic is wave-uniform
// control flow + LDS, so the waves of a block never synchronise after the table load.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ksh_tables.h"

#define KSH_A 13527
#define KSH_OBSF 3726
#define KSH_WPB 4             // waves (= games) per block
#define KSH_MASKW 432         // dwords of the per-game legal bit-set (423 used + funnel-shift slack)
#define KSH_BITROW 424        // dwords per row of the packed legal mask (13 527 bits, rows 16-byte aligned)
#define KSH_M27 0x7FFFFFFu

// state record byte offsets (include/keisei_hip.h, ksh_env_set_state)
#define S_HAND 81
#define S_SIDE 95
#define S_MC 96
#define S_OVER 98
#define S_REASON 99
#define S_WINNER 100
#define S_HBASE 102
#define S_HASH KSH_S_HASH      // 8 bytes: position key of the current state
#define S_NFRAC KSH_S_NFRAC    // float: (move_count + 1) / max_moves

struct KshStepArgs {
    uint8_t *state;            // [N][128]
    unsigned long long *hist;  // [N][hist_cap] position hashes of post-move states
    const KshTables *tables;
    const long long *actions;  // [N] or null (mode != 2)
    const uint8_t *prev_mask;  // [N][A] or null
    const uint32_t *prev_bits; // [N][KSH_BITROW] packed form of prev_mask, or null
    const uint8_t *reset_mask; // [N] or null (mode 1)
    float *obs;                // [N][46*81]
    uint8_t *mask;             // [N][A] one byte per action, or null when only the packed form is wanted
    uint32_t *mask_bits;       // [N][KSH_BITROW] bit i of a row <-> action id i, or null
    int *n_legal;              // [N]
    float *reward;             // [N] (mode 2)
    uint8_t *done;
    uint8_t *reason;
    signed char *winner;
    int n_games;
    int max_moves;
    int hist_cap;
    int mode;                  // 0 observe, 1 reset (masked), 2 step, 3 observe + evaluate termination (from_sfen)
    int auto_reset;
    unsigned long long *stamps; // diagnostic build only (-DKSH_STAMPS): [N][16] s_memtime values
};

// Diagnostic build: per-phase shader-clock stamps of lane 0 of every wave (never compiled into the product .so).
#ifdef KSH_STAMPS
#define KSH_STAMP(k)                                                                         \
    do {                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        if (lane == 0 && a.stamps) a.stamps[(size_t)g * 32 + (k)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    } while (0)
#else
#define KSH_STAMP(k) do { } while (0)
#endif

struct BB3 {
    uint32_t w0, w1, w2;
};

__device__ __forceinline__ BB3 bb_and(BB3 a, BB3 b) { return BB3{a.w0 & b.w0, a.w1 & b.w1, a.w2 & b.w2}; }
__device__ __forceinline__ BB3 bb_or(BB3 a, BB3 b) { return BB3{a.w0 | b.w0, a.w1 | b.w1, a.w2 | b.w2}; }
__device__ __forceinline__ BB3 bb_andn(BB3 a, BB3 b) { return BB3{a.w0 & ~b.w0, a.w1 & ~b.w1, a.w2 & ~b.w2}; }
__device__ __forceinline__ bool bb_any(BB3 a) { return (a.w0 | a.w1 | a.w2) != 0; }
__device__ __forceinline__ BB3 bb_zero() { return BB3{0, 0, 0}; }
__device__ __forceinline__ int sq_word(int sq) { return (sq >= 54) ? 2 : (sq >= 27 ? 1 : 0); }
__device__ __forceinline__ BB3 bb_sq(int sq) {
    int w = sq_word(sq);
    uint32_t b = 1u << (sq - 27 * w);
    return BB3{w == 0 ? b : 0u, w == 1 ? b : 0u, w == 2 ? b : 0u};
}
__device__ __forceinline__ bool bb_test(BB3 a, int sq) {
    int w = sq_word(sq);
    uint32_t x = (w == 0) ? a.w0 : (w == 1 ? a.w1 : a.w2);
    return (x >> (sq - 27 * w)) & 1u;
}
__device__ __forceinline__ BB3 bb_from_ballots(unsigned long long lo, unsigned long long hi) {
    return BB3{(uint32_t)(lo & KSH_M27), (uint32_t)((lo >> 27) & KSH_M27),
               (uint32_t)(((lo >> 54) | (hi << 10)) & KSH_M27)};
}
__device__ __forceinline__ int sq_row(int sq) { return (sq * 57) >> 9; }  // exact for 0 <= sq < 81

// Squares of ray M (from its origin outwards) up to and including the nearest blocker in B = M & occ.
__device__ __forceinline__ BB3 ray_attack(BB3 M, BB3 B, bool positive) {
    BB3 a;
    if (positive) {  // nearest blocker = lowest set bit; x ^ (x-1) = bits up to and including it (all ones for x = 0)
        a.w0 = M.w0 & (B.w0 ^ (B.w0 - 1u));
        a.w1 = B.w0 ? 0u : (M.w1 & (B.w1 ^ (B.w1 - 1u)));
        a.w2 = (B.w0 | B.w1) ? 0u : (M.w2 & (B.w2 ^ (B.w2 - 1u)));
    } else {         // nearest blocker = highest set bit
        uint32_t h2 = B.w2 ? (0xFFFFFFFFu << (31 - __clz((int)B.w2))) : 0xFFFFFFFFu;
        uint32_t h1 = B.w1 ? (0xFFFFFFFFu << (31 - __clz((int)B.w1))) : 0xFFFFFFFFu;
        uint32_t h0 = B.w0 ? (0xFFFFFFFFu << (31 - __clz((int)B.w0))) : 0xFFFFFFFFu;
        a.w2 = M.w2 & h2;
        a.w1 = B.w2 ? 0u : (M.w1 & h1);
        a.w0 = (B.w2 | B.w1) ? 0u : (M.w0 & h0);
    }
    return a;
}
// Square index of the blocker nearest to the ray origin, or -1.
__device__ __forceinline__ int bb_nearest(BB3 B, bool positive) {
    if (positive) {
        if (B.w0) return __ffs((int)B.w0) - 1;
        if (B.w1) return 27 + __ffs((int)B.w1) - 1;
        if (B.w2) return 54 + __ffs((int)B.w2) - 1;
        return -1;
    }
    if (B.w2) return 54 + 31 - __clz((int)B.w2);
    if (B.w1) return 27 + 31 - __clz((int)B.w1);
    if (B.w0) return 31 - __clz((int)B.w0);
    return -1;
}
// Direction index from square a to square b when they share a rank, file or diagonal, else -1.
__device__ __forceinline__ int sq_dir(int a, int b) {
    int ra = sq_row(a), rb = sq_row(b);
    int dr = rb - ra, dc = (b - 9 * rb) - (a - 9 * ra);
    if (dr == 0 && dc == 0) return -1;
    if (dr == 0) return dc < 0 ? 2 : 3;
    if (dc == 0) return dr < 0 ? 0 : 1;
    if (dr == dc) return dr < 0 ? 4 : 5;
    if (dr == -dc) return dr < 0 ? 6 : 7;
    return -1;
}

// Compiler-level ordering point for LDS traffic between lanes of ONE wave.  The hardware executes a
// wave's DS instructions in order, so no instruction is needed; this only pins the order in the IR.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Wave-wide reductions by xor-butterfly (ds_bpermute): every lane ends up with the result.  (LDS atomics
// on one address are turned by hipcc into a serial readlane loop over the active lanes -- far slower.)
__device__ __forceinline__ uint32_t wave_or(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v |= (uint32_t)__shfl_xor((int)v, o);
    return v;
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v ^= (uint32_t)__shfl_xor((int)v, o);
    return v;
}
// sum of small per-lane counts (< 64) from six ballots, no LDS round trip
__device__ __forceinline__ int wave_sum_small(int v) {
    int s = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) s += __popcll(__ballot((v >> k) & 1)) << k;
    return s;
}

// 16-byte store of an output row chunk (plain global_store_dwordx4: the non-temporal form measured 20 % slower).
__device__ __forceinline__ void ksh_store16(void *p, uint4 v) { *reinterpret_cast<uint4 *>(p) = v; }

// per-wave LDS scratch
struct KshWaveLds {
    uint32_t maskbits[KSH_MASKW];  // legal action bit-set, bit i <-> action id i
    uint32_t st32[32];             // the 128-byte state record
    uint8_t plist[64];             // squares of the pieces on the board (compacted)
    uint32_t obsbits[80];          // planes 0-27 of the observation as a bit-set: bit plane*81 + pov_sq (2268 bits)
    float metav[20];               // planes 28-45: hands (own P..R, opp P..R), side-to-move flag, move fraction, 0, 0
    uint32_t pad_[8];
};

struct KshBlockLds {
    KshTables tab;
    KshWaveLds wave[KSH_WPB];
};
// Variant without the LDS copy of the attack tables (TABG = 1): the lookups go to global memory (24.9 KB that every CU's
// vector L1 / the L2 keep hot), a block needs 8.8 KB of LDS instead of 33.6 KB and no barrier.  Measured, not assumed:
// DESIGN.md section 4.1.
struct KshBlockLdsG {
    KshLuts lut;               // the small expansion tables are staged in every variant
    KshWaveLds wave[KSH_WPB];
};
template <int TABG> struct KshLdsSel { typedef KshBlockLds type; };
template <> struct KshLdsSel<1> { typedef KshBlockLdsG type; };
__device__ __forceinline__ const KshTables &ksh_tab_of(const KshBlockLds &L, const KshTables *) { return L.tab; }
__device__ __forceinline__ const KshTables &ksh_tab_of(const KshBlockLdsG &, const KshTables *g) { return *g; }
__device__ __forceinline__ const KshLuts &ksh_lut_of(const KshBlockLds &L) { return L.tab.lut; }
__device__ __forceinline__ const KshLuts &ksh_lut_of(const KshBlockLdsG &L) { return L.lut; }

#define ST8(W) (reinterpret_cast<uint8_t *>((W).st32))

__device__ __forceinline__ BB3 lds_bb(const uint32_t (*tbl)[81][4], int k, int sq) {
    const uint32_t *p = tbl[k][sq];
    return BB3{p[0], p[1], p[2]};
}

// Observation of the state in W.st32 (shogi/shogi_game_io.py:434-539), POV of the side to move, written with
// coalesced 16-byte stores.  Planes 0-27 are one-hot piece planes: they are first collected as a 2 268-bit
// set in LDS (one LDS atomic per piece) and expanded 4 bits -> 4 floats per lane; planes 28-45 are constant
// per plane (hand counts / 18 and move_count / max_moves as the reference's double division rounded to
// float32, the side-to-move flag, two zero planes).
#define KSH_BOARD_F 2268  // 28 * 81
__device__ __forceinline__ float ksh_obs_elem(const KshWaveLds &W, int f) {
    if (f < KSH_BOARD_F) return ((W.obsbits[f >> 5] >> (f & 31)) & 1u) ? 1.0f : 0.0f;
    return W.metav[(unsigned)(f - KSH_BOARD_F) / 81u];
}
// planes 28-45 as float32 values into W.metav (callers sync afterwards)
__device__ __forceinline__ void ksh_obs_meta(KshWaveLds &W, const KshLuts &LU, const KshStepArgs &a, int lane, int mf_idx,
                                             float mf_hint) {
    const uint8_t *st8 = ST8(W);
    const int side = st8[S_SIDE];
    // move_count / max_moves in double, stored as float32 (:535-536).  The state record carries the value for its NEXT
    // move count (computed by the previous step after its stores were out); anything else -- a freshly reset or loaded
    // position -- is divided here, behind a wave-uniform branch so that the common path never issues the f64 division.
    const int mcount = __builtin_amdgcn_readfirstlane((int)(st8[S_MC] | (st8[S_MC + 1] << 8)));
    float mfv = mf_hint;
    if (mcount != __builtin_amdgcn_readfirstlane(mf_idx))
        mfv = a.max_moves > 0 ? (float)((double)mcount / (double)a.max_moves) : 0.0f;
    if (lane < 14) {
        // planes 28-34 own hand, 35-41 opponent hand; count/18.0 in double, stored as float32 (:513-527): host-built table
        const int cidx = lane < 7 ? side : 1 - side;
        const int cnt = st8[S_HAND + 7 * cidx + (lane < 7 ? lane : lane - 7)];
        W.metav[lane] = cnt < 20 ? LU.hand18[cnt] : (float)((double)cnt / 18.0);
    } else if (lane == 14) {
        W.metav[14] = side == 0 ? 1.0f : 0.0f;
    } else if (lane == 15) {
        W.metav[15] = mfv;
    } else if (lane < 20) {
        W.metav[lane] = 0.0f;
    }
}
// The float32 observation writer in parts: ksh_obs_begin builds the bit-set of planes 0-27 and the 18 plane constants in
// LDS, writes the few floats outside the aligned chunks and returns the lane's plan; ksh_obs_board / ksh_obs_metar issue
// rounds [IT0, IT0 + N) of it.  (Round 3 measured issuing the 15 rounds in pieces BETWEEN the phases of the move generator,
// to overlap the store path with the generator's latency chain: 24.6-24.9 us against 24.1-24.2 us for all rounds at once at
// 4096 games -- profiles/r03_env_variants.log -- so ksh_write_obs issues them in one go.)
struct KshObsPlan {
    float *o0;            // planes 0-27: address of this lane's chunk in round 0 (round it: + 256 floats)
    const uint32_t *bw;   //   bit-set word holding that chunk's first bit (round it: + 8 words)
    float *om;            // planes 28-45: address of this lane's chunk in round 0
    int sh;               //   shift of the chunk's first bit inside bw[0] (the same in every round)
    int nb;               // rounds' bounds: a board chunk exists while lane + 64 it < nb, a meta chunk while lane + 64 it < nm
    int nm;
    int rem, p0;          // meta: offset of the chunk's first float inside its plane, metav index of that plane (advance per round)
};
__device__ __forceinline__ KshObsPlan ksh_obs_begin(KshWaveLds &W, const KshLuts &LU, const KshStepArgs &a, int g, int lane, int mf_idx,
                                                    float mf_hint) {
    const uint8_t *st8 = ST8(W);
    const int side = st8[S_SIDE];
    for (int i = lane; i < 80; i += 64) W.obsbits[i] = 0;
    wave_lds_sync();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int s = pass * 64 + lane;
        if (s < 81 && st8[s] != 0) {
            const int pc = st8[s];
            const int col = pc >= 15 ? 1 : 0;
            const int t = pc - 1 - 14 * col;
            const int bit = ((col == side ? 0 : 14) + t) * 81 + (side == 0 ? s : 80 - s);   // both r and c mirrored (:468-469)
            atomicOr(&W.obsbits[bit >> 5], 1u << (bit & 31));
        }
    }
    ksh_obs_meta(W, LU, a, lane, mf_idx, mf_hint);
    wave_lds_sync();
    // Row g starts at a 4-byte aligned address (rows are 14 904 B apart, so even when the tensor itself is 16-byte
    // aligned every other row is not): `head` floats are written singly up to the first 16-byte boundary, then
    // aligned 16-byte chunks, then the 0-3 remaining floats.
    float *orow = a.obs + (size_t)g * KSH_OBSF;
    const int head = (int)(((16 - (reinterpret_cast<size_t>(orow) & 15)) & 15) >> 2);
    const int nfull = (KSH_OBSF - head) >> 2;       // 930 or 931 aligned chunks
    const int nboard = (KSH_BOARD_F - head) >> 2;   // chunks lying entirely inside planes 0-27
    KshObsPlan pl;
    // planes 0-27: round `it` of lane l is the chunk 256 floats after round it-1's, so the bit-set word index advances by 8
    // and the shift never changes; the four floats come out of the 16-entry nibble table (one 16-byte LDS read)
    const int f0 = head + 4 * lane;
    pl.sh = f0 & 31;
    pl.bw = &W.obsbits[f0 >> 5];
    pl.o0 = orow + f0;
    pl.nb = nboard;
    // planes 28-45 are constant per plane; a chunk touches at most two of them.  Plane and offset of a lane's chunk are
    // divided out once and advanced by 256 = 3 * 81 + 13 floats per round.  (One plane of bias keeps the division unsigned
    // for the chunk that straddles plane 27 / 28; that chunk is written on its own right here.)
    const int fm = head + 4 * (nboard + lane);
    const int mm = fm - KSH_BOARD_F + 81;
    const int pp = (int)((unsigned)mm / 81u);
    pl.rem = mm - 81 * pp;
    pl.p0 = pp - 1;                                 // -1: the straddling chunk (lane 0, round 0)
    pl.om = orow + fm;
    pl.nm = nfull - nboard;
    if (((KSH_BOARD_F - head) & 3) != 0 && lane == 0) {   // wave-uniform outer condition: every other row has such a chunk
        const int f = head + 4 * nboard;
        ksh_store16(orow + f, uint4{__float_as_uint(ksh_obs_elem(W, f)), __float_as_uint(ksh_obs_elem(W, f + 1)),
                                    __float_as_uint(ksh_obs_elem(W, f + 2)), __float_as_uint(ksh_obs_elem(W, f + 3))});
    }
    if (lane < head) orow[lane] = ((W.obsbits[0] >> lane) & 1u) ? 1.0f : 0.0f;     // head <= 3: plane 0
    const int tail0 = head + 4 * nfull;
    if (lane < KSH_OBSF - tail0) orow[tail0 + lane] = W.metav[17];                  // the last floats of plane 45
    return pl;
}
template <int IT0, int N>
__device__ __forceinline__ void ksh_obs_board(const KshLuts &LU, const KshObsPlan &pl, int lane) {
    const uint4 *nib = reinterpret_cast<const uint4 *>(LU.nib_f32);
#pragma unroll
    for (int it = IT0; it < IT0 + N; it++) {        // 9 rounds in all: 9 * 64 >= 567
        if (lane + 64 * it < pl.nb) {
            const uint32_t x = __funnelshift_r(pl.bw[8 * it], pl.bw[8 * it + 1], pl.sh);
            ksh_store16(pl.o0 + 256 * it, nib[x & 15u]);
        }
    }
}
template <int IT0, int N>
__device__ __forceinline__ void ksh_obs_metar(const KshWaveLds &W, KshObsPlan &pl, int lane) {
#pragma unroll
    for (int it = IT0; it < IT0 + N; it++) {        // 6 rounds in all: 6 * 64 >= 931 - 566
        if (lane + 64 * it < pl.nm && (it > 0 || pl.p0 >= 0)) {
            const float v0 = W.metav[pl.p0], v1 = W.metav[pl.p0 + 1];      // p0 + 1 <= 18 < 20: in bounds
            const int r = 81 - pl.rem;              // floats of plane p0 from the chunk's first one on
            const float4 v = make_float4(v0, 1 < r ? v0 : v1, 2 < r ? v0 : v1, 3 < r ? v0 : v1);
            ksh_store16(pl.om + 256 * it, uint4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)});
        }
        pl.rem += 13;
        pl.p0 += 3;
        if (pl.rem >= 81) { pl.rem -= 81; pl.p0 += 1; }
    }
}
// the whole observation at once (reset / observe paths, and the rewrite after an auto-reset)
__device__ __forceinline__ void ksh_write_obs(KshWaveLds &W, const KshLuts &LU, const KshStepArgs &a, int g, int lane, int mf_idx,
                                              float mf_hint) {
    KshObsPlan pl = ksh_obs_begin(W, LU, a, g, lane, mf_idx, mf_hint);
    ksh_obs_board<0, 9>(LU, pl, lane);
    ksh_obs_metar<0, 6>(W, pl, lane);
}

// Compact observation (not part of the reference contract): the same 46 planes as bfloat16 in
// channels-last order -- element (pov_sq, ch) at row[pov_sq * 46 + ch], 7 452 B per game instead of 14 904 --
// which is bit for bit what `obs.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)` holds (round to
// nearest even of the float32 plane values), i.e. exactly the tensor the bf16-autocast stem convolution consumes.
// A dword holds channels (2k, 2k+1) of one square (46 is even): for k < 14 it is a one-hot test against the
// square's plane index, for k >= 14 one of nine precomputed pairs of the constant planes.
#define KSH_OBS16_DW 1863  // 81 * 46 / 2
__device__ __forceinline__ uint32_t ksh_obs_bf16(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b + 0x7FFFu + ((b >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint32_t ksh_obs16_dword(const uint8_t *plane, const uint32_t *pair, int d) {
    const int sq = (int)((unsigned)d / 23u);
    const int k = d - 23 * sq;
    const uint32_t pl = plane[sq];
    if (k >= 14) return pair[k - 14];
    return (int)(pl >> 1) == k ? (0x3F80u << ((pl & 1u) * 16u)) : 0u;
}
__device__ __forceinline__ void ksh_write_obs16(KshWaveLds &W, const KshLuts &LU, const KshStepArgs &a, int g, int lane, int mf_idx,
                                                float mf_hint) {
    const uint8_t *st8 = ST8(W);
    const int side = st8[S_SIDE];
    uint8_t *plane = reinterpret_cast<uint8_t *>(W.obsbits);       // [81] plane index of the piece on each POV square, 0xFF = empty
    uint32_t *pair = W.obsbits + 24;                                // [9] bf16 pairs of planes 28-45
    if (lane < 24) W.obsbits[lane] = 0xFFFFFFFFu;
    wave_lds_sync();
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const int s = pass * 64 + lane;
        if (s < 81 && st8[s] != 0) {
            const int pc = st8[s];
            const int col = pc >= 15 ? 1 : 0;
            const int t = pc - 1 - 14 * col;
            plane[side == 0 ? s : 80 - s] = (uint8_t)((col == side ? 0 : 14) + t);
        }
    }
    ksh_obs_meta(W, LU, a, lane, mf_idx, mf_hint);
    wave_lds_sync();
    if (lane < 9) pair[lane] = ksh_obs_bf16(W.metav[2 * lane]) | (ksh_obs_bf16(W.metav[2 * lane + 1]) << 16);
    wave_lds_sync();
    uint32_t *orow = reinterpret_cast<uint32_t *>(a.obs) + (size_t)g * KSH_OBS16_DW;   // rows are 4-byte aligned (7 452 B apart)
    const int head = (int)(((16 - (reinterpret_cast<size_t>(orow) & 15)) & 15) >> 2);
    const int nfull = (KSH_OBS16_DW - head) >> 2;   // 465 aligned 16-byte chunks
#pragma unroll
    for (int it = 0; it < 8; it++) {                // 8 * 64 >= 465
        const int c = lane + 64 * it;
        if (c < nfull) {
            const int d = head + 4 * c;
            int sq = (int)((unsigned)d / 23u);
            int k = d - 23 * sq;
            uint32_t pl = plane[sq];
            const uint32_t pl_next = plane[sq + 1];  // sq + 1 <= 81: still inside obsbits
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                w[j] = k >= 14 ? pair[k - 14] : ((int)(pl >> 1) == k ? (0x3F80u << ((pl & 1u) * 16u)) : 0u);
                k++;
                if (k == 23) { k = 0; pl = pl_next; }
            }
            ksh_store16(orow + d, uint4{w[0], w[1], w[2], w[3]});
        }
    }
    if (lane < head) orow[lane] = ksh_obs16_dword(plane, pair, lane);
    const int tail0 = head + 4 * nfull;
    if (lane < KSH_OBS16_DW - tail0) orow[tail0 + lane] = ksh_obs16_dword(plane, pair, tail0 + lane);
}

// ------------------------------------------------------------------------------------------------
// Legal move generation for the side `me` on the board in W.st32.  With emit: sets the action bits in
// W.maskbits (pawn drops on the squares in ufz_cand excluded).  Returns the number of legal moves
// (without those pawn drops), whether `me` is in check, and ufz_cand: the pawn-drop squares whose
// legality hinges on the uchi-fu-zume reply search (empty unless want_ufz).  In any position reachable
// by legal play that is at most the one square in front of the enemy king; when the side NOT to
// move is already in check (hand-made positions only) the reference's test "the king is attacked
// after the drop" (shogi_rules_logic.py:329-332) holds for every pawn drop, so all of them are.
// ------------------------------------------------------------------------------------------------

// ---- emission helpers: the action ids of one from-square are the 160 consecutive bits [160*from, 160*from+160)
// of the legal bit-set (utils/utils.py:208-241: id = ((from*80 + to - (to > from))*2 + promote), i.e. exactly five
// dwords that no other piece writes.  They are built without loops: the two 81-bit target sets (non-promoting /
// promoting) are made contiguous, the from-square is squeezed out of each stream once (three funnel shifts and three
// bit-field inserts), and the two 80-bit streams are bit-interleaved 16 + 16 bits at a time through the byte-spread
// table in LDS (round 3: the arithmetic interleave was 324 of the kernel's 1 795 VALU instructions per game-step).
struct Stream81 {
    uint32_t c0, c1, c2;
};
__device__ __forceinline__ Stream81 bb_stream(BB3 b) {
    return Stream81{b.w0 | (b.w1 << 27), (b.w1 >> 5) | (b.w2 << 22), b.w2 >> 10};
}
// (a & m) | (b & ~m): one v_bfi_b32
__device__ __forceinline__ uint32_t ksh_bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
// the three words of the 81-bit mask "bits below `from`"
struct Low81 {
    uint32_t m0, m1, m2;
};
__device__ __forceinline__ Low81 low_mask81(int from) {
    Low81 m;
    m.m0 = from >= 32 ? 0xFFFFFFFFu : ((1u << (from & 31)) - 1u);
    m.m1 = from <= 32 ? 0u : (from >= 64 ? 0xFFFFFFFFu : ((1u << ((from - 32) & 31)) - 1u));
    m.m2 = from <= 64 ? 0u : ((1u << ((from - 64) & 31)) - 1u);
    return m;
}
// drop bit `from` out of the stream: the bits above it move down by one (80 bits remain)
__device__ __forceinline__ Stream81 stream_squeeze(Stream81 x, Low81 m) {
    const uint32_t t0 = __funnelshift_r(x.c0, x.c1, 1), t1 = __funnelshift_r(x.c1, x.c2, 1), t2 = x.c2 >> 1;
    return Stream81{ksh_bfi(m.m0, x.c0, t0), ksh_bfi(m.m1, x.c1, t1), ksh_bfi(m.m2, x.c2, t2)};
}
// dword K (0..4) of the slice: bits [16K, 16K+16) of the two squeezed streams, interleaved (non-promoting even, promoting odd)
template <int K>
__device__ __forceinline__ uint32_t slice_word(const KshLuts &LU, Stream81 n, Stream81 q) {
    const uint32_t nw = (K >> 1) == 0 ? n.c0 : ((K >> 1) == 1 ? n.c1 : n.c2);
    const uint32_t qw = (K >> 1) == 0 ? q.c0 : ((K >> 1) == 1 ? q.c1 : q.c2);
    constexpr int sh = (K & 1) * 16;
    const uint32_t n0 = LU.spread8[(nw >> sh) & 0xFFu], n1 = LU.spread8[(nw >> (sh + 8)) & 0xFFu];
    const uint32_t q0 = LU.spread8[(qw >> sh) & 0xFFu], q1 = LU.spread8[(qw >> (sh + 8)) & 0xFFu];
    return (n0 | (n1 << 16)) | ((q0 | (q1 << 16)) << 1);
}

#ifdef KSH_STAMPS
#define GEN_STAMP(k)                                                              \
    do {                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                        \
        if (lane == 0 && dbg) dbg[k] = __builtin_amdgcn_s_memtime();              \
        __builtin_amdgcn_sched_barrier(0);                                        \
    } while (0)
#else
#define GEN_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ int ksh_gen(const KshTables &TB, const KshLuts &LU, KshWaveLds &W, int lane, int me, bool emit,
                                       bool want_ufz, bool &in_check, BB3 &ufz_cand, unsigned long long *dbg = nullptr) {
    const uint8_t *brd = ST8(W);
    const int p0 = brd[lane];
    const int p1 = (lane < 17) ? brd[64 + lane] : 0;
    const bool own0 = p0 != 0 && ((p0 >= 15) == (me == 1));
    const bool own1 = p1 != 0 && ((p1 >= 15) == (me == 1));
    const unsigned long long occ_lo = __ballot(p0 != 0), occ_hi = __ballot(p1 != 0);
    const BB3 occ = bb_from_ballots(occ_lo, occ_hi);
    const BB3 own = bb_from_ballots(__ballot(own0), __ballot(own1));
    const int myking = 1 + 7 + 14 * me, opking = 1 + 7 + 14 * (1 - me);
    const unsigned long long k_lo = __ballot(p0 == myking), k_hi = __ballot(p1 == myking);
    const int ksq = k_lo ? (__ffsll((long long)k_lo) - 1) : (k_hi ? 64 + __ffsll((long long)k_hi) - 1 : -1);
    const unsigned long long e_lo = __ballot(p0 == opking), e_hi = __ballot(p1 == opking);
    const int eksq = e_lo ? (__ffsll((long long)e_lo) - 1) : (e_hi ? 64 + __ffsll((long long)e_hi) - 1 : -1);
    const BB3 occx = (ksq >= 0) ? bb_andn(occ, bb_sq(ksq)) : occ;  // own king lifted: enemy sliders x-ray through it

    // compact the occupied squares into plist, clear the scratch
    const int n_lo = __popcll(occ_lo);
    const int np = n_lo + __popcll(occ_hi);
    if (p0 != 0) W.plist[__popcll(occ_lo & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    if (p1 != 0) W.plist[n_lo + __popcll(occ_hi & ((1ull << lane) - 1ull))] = (uint8_t)(64 + lane);
    wave_lds_sync();

    GEN_STAMP(11);  // occupancy ballots + compaction
    // ---- lane = piece: attack set of every piece on the board
    int sq = 0, pc = 0;
    if (lane < np) { sq = W.plist[lane]; pc = brd[sq]; }
    const bool has = lane < np;
    const int color = pc >= 15 ? 1 : 0;
    const int type = has ? (pc - 1 - 14 * color) : 7;
    const bool mine = has && color == me;
    const bool foe = has && color != me;
    // step class per type (Black numbering; +4 for White's P,N,S,G), 15 = none:   P N S G B R K +P +L +N +S +B +R ; L = none
    //   types: 0 P,1 L,2 N,3 S,4 G,5 B,6 R,7 K,8 +P,9 +L,10 +N,11 +S,12 +B,13 +R
    const unsigned long long CLS = 0xA933338FF321F0ull;  // nibble t = class of type t
    int cls = (int)((CLS >> (4 * type)) & 0xF);
    if (cls < 4 && color) cls += 4;
    // slider kind per type: 0 none, 1 lance, 2 bishop-like, 3 rook-like
    const unsigned long long KIND = 0x32000003200010ull;
    const int kind = has ? (int)((KIND >> (4 * type)) & 0xF) : 0;
    const BB3 O = foe ? occx : occ;
    BB3 att = bb_zero();
    if (has && cls != 15) att = lds_bb(TB.step, cls, sq);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const bool act = kind >= 2 || (kind == 1 && i == 0);
        const int d = kind == 3 ? i : (kind == 2 ? 4 + i : color);  // lance: Black slides N (0), White S (1)
        if (act) {
            BB3 M = lds_bb(TB.ray, d, sq);
            att = bb_or(att, ray_attack(M, bb_and(M, O), (d & 1) != 0));
        }
    }
    GEN_STAMP(12);  // attack sets
    // files holding an own unpromoted pawn (nifu)
    const uint32_t nifu = wave_or((mine && type == 0) ? (1u << (sq - 9 * sq_row(sq))) : 0u);
    // checkers: enemy pieces attacking the king square
    const unsigned long long chk = __ballot(foe && ksq >= 0 && bb_test(att, ksq));
    const int ncheck = __popcll(chk);
    const int csq = __shfl(sq, chk ? (__ffsll((long long)chk) - 1) : 0);
    in_check = (ksq < 0) || ncheck > 0;  // king missing => "in check" (shogi_rules_logic.py:42-55)
    const bool opp_in_check = __ballot(mine && eksq >= 0 && bb_test(att, eksq)) != 0ull;
    // king flight squares attacked by the enemy (sliders see through the lifted king): one ballot per candidate
    // square instead of OR-reducing whole attack maps across the wave
    BB3 katt = bb_zero();
    if (ksq >= 0) {
        const int kr = sq_row(ksq), kc = ksq - 9 * kr;
#pragma unroll
        for (int d = 0; d < 8; d++) {                   // same direction order as the ray tables
            const int r = kr + ((d == 0 || d == 4 || d == 6) ? -1 : ((d == 1 || d == 5 || d == 7) ? 1 : 0));
            const int c = kc + ((d == 2 || d == 4 || d == 7) ? -1 : ((d == 3 || d == 5 || d == 6) ? 1 : 0));
            const bool on = r >= 0 && r < 9 && c >= 0 && c < 9;
            const int t = on ? r * 9 + c : 0;
            const bool hit = __ballot(foe && on && bb_test(att, t)) != 0ull;
            if (hit) katt = bb_or(katt, bb_sq(t));
        }
    }

    GEN_STAMP(13);  // nifu + checkers + king flights
    // ---- lane = own piece: is it pinned?  (aligned with the king, nothing in between, enemy slider right behind)
    bool pinned = false;
    int pin_d = 0;
    if (mine && ksq >= 0 && sq != ksq) {
        const int d = sq_dir(ksq, sq);
        if (d >= 0) {
            const BB3 between = bb_and(lds_bb(TB.ray, d, ksq), lds_bb(TB.ray, d ^ 1, sq));
            if (!bb_any(bb_and(between, occ))) {
                const int n2 = bb_nearest(bb_and(lds_bb(TB.ray, d, sq), occ), (d & 1) != 0);
                if (n2 >= 0) {
                    const int q2 = brd[n2];
                    if ((q2 >= 15) != (me == 1)) {
                        const int t2 = q2 - 1 - (q2 >= 15 ? 14 : 0);
                        // rook-likes pin along ranks/files, bishop-likes along diagonals, a lance only from in front
                        const bool slides = (d < 4) ? (t2 == 6 || t2 == 13 || (t2 == 1 && d == me)) : (t2 == 5 || t2 == 12);
                        if (slides) { pinned = true; pin_d = d; }
                    }
                }
            }
        }
    }

    // squares a non-king move must land on while in check; squares a drop may use
    BB3 block = BB3{KSH_M27, KSH_M27, KSH_M27};
    BB3 dropok = bb_andn(BB3{KSH_M27, KSH_M27, KSH_M27}, occ);
    if (ncheck == 1) {
        const int d = sq_dir(ksq, csq);
        BB3 between = bb_zero();
        if (d >= 0) between = bb_and(lds_bb(TB.ray, d, ksq), lds_bb(TB.ray, d ^ 1, csq));
        block = bb_or(between, bb_sq(csq));
        dropok = bb_and(dropok, between);
    } else if (ncheck >= 2) {
        block = bb_zero();
        dropok = bb_zero();
    }
    if (ksq < 0) dropok = bb_zero();

    GEN_STAMP(14);  // pins + block sets
    // ---- lane = own piece: filter to legal targets, emit action ids
    int cnt = 0;
    if (mine && ksq >= 0) {
        BB3 T = bb_andn(att, own);
        if (sq == ksq) {
            T = bb_andn(T, katt);
        } else {
            T = bb_and(T, block);
            if (pinned) T = bb_and(T, lds_bb(TB.ray, pin_d, ksq));
        }
        // promotion rules (shogi_rules_logic.py:382-421)
        const bool promotable = type <= 3 || type == 5 || type == 6;
        const int zw = me == 0 ? 0 : 2;                      // promotion-zone word
        const BB3 zone = BB3{zw == 0 ? KSH_M27 : 0u, 0u, zw == 2 ? KSH_M27 : 0u};
        const uint32_t last1 = me == 0 ? 0x1FFu : (0x1FFu << 18);           // last rank inside the zone word
        const uint32_t last2 = me == 0 ? 0x3FFFFu : (0x3FFFFu << 9);        // last two ranks
        BB3 must = bb_zero();
        if (type == 0 || type == 1) { if (me == 0) must.w0 = last1; else must.w2 = last1; }
        if (type == 2) { if (me == 0) must.w0 = last2; else must.w2 = last2; }
        const bool from_zone = sq_word(sq) == zw;
        const BB3 Tp = promotable ? (from_zone ? T : bb_and(T, zone)) : bb_zero();
        const BB3 Tn = bb_andn(T, must);
        cnt = __popc(Tp.w0) + __popc(Tp.w1) + __popc(Tp.w2) + __popc(Tn.w0) + __popc(Tn.w1) + __popc(Tn.w2);
        if (emit && cnt != 0) {
            const Low81 lm = low_mask81(sq);
            const Stream81 sn = stream_squeeze(bb_stream(Tn), lm), sp = stream_squeeze(bb_stream(Tp), lm);
            uint32_t *out = &W.maskbits[5 * sq];             // this from-square's private 160-bit slice
            out[0] = slice_word<0>(LU, sn, sp);
            out[1] = slice_word<1>(LU, sn, sp);
            out[2] = slice_word<2>(LU, sn, sp);
            out[3] = slice_word<3>(LU, sn, sp);
            out[4] = slice_word<4>(LU, sn, sp);
        }
    }

    GEN_STAMP(15);  // board moves emitted
    // ---- lane = square: drops (shogi_rules_logic.py:424-483, 562-583)
    const uint8_t *hand = brd + S_HAND + 7 * me;
    uint32_t ht = 0;
#pragma unroll
    for (int t = 0; t < 7; t++) ht |= (hand[t] != 0 ? 1u : 0u) << t;
    ufz_cand = bb_zero();
    if (want_ufz && (ht & 1u) && eksq >= 0) {
        // squares where a pawn may be dropped at all: drop-ok, not the last rank, no own pawn on the file
        const uint32_t files = (nifu & 0x1FFu) * 0x40201u;
        BB3 pawnok = bb_andn(dropok, BB3{files, files, files});
        if (me == 0) pawnok.w0 &= ~0x1FFu; else pawnok.w2 &= ~(0x1FFu << 18);
        const int u = eksq + (me == 0 ? 9 : -9);  // a pawn here attacks the enemy king
        if (opp_in_check) ufz_cand = pawnok;
        else if (u >= 0 && u < 81) ufz_cand = bb_and(pawnok, bb_sq(u));
    }
    if (ht != 0) {
        const int last_r = me == 0 ? 0 : 8, second_r = me == 0 ? 1 : 7;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            const int s = pass * 64 + lane;
            if (s < 81 && bb_test(dropok, s)) {
                const int r = sq_row(s), c = s - 9 * r;
                uint32_t f = ht;
                if (r == last_r) f &= ~7u;          // P, L, N
                if (r == second_r) f &= ~4u;        // N
                if ((nifu >> c) & 1u) f &= ~1u;     // nifu
                if (bb_test(ufz_cand, s)) f &= ~1u; // decided by the nested reply search
                cnt += __popc(f);
                if (emit && f) {
                    const int bit = 12960 + 7 * s;  // utils/utils.py:253-266: 12960 + sq*7 + piece
                    const int wd = bit >> 5, sh = bit & 31;
                    atomicOr(&W.maskbits[wd], f << sh);
                    if (sh > 25) atomicOr(&W.maskbits[wd + 1], f >> (32 - sh));
                }
            }
        }
    }
    // a lane emits at most ~40 board targets x2 + 2x7 drop bits; keep the ballot sum exact for any count
    const int total = wave_sum_small(cnt & 63) + (wave_sum_small(cnt >> 6) << 6);
    wave_lds_sync();
    return total;
}


// Legal moves of `me` including the uchi-fu-zume decision (shogi_rules_logic.py:275-359): for every
// candidate square drop the pawn on the LDS board, generate the opponent's replies (nested pawn drops
// skip their own test, :461-467) and keep the drop unless it gives check AND leaves no reply.
__device__ __forceinline__ int ksh_gen_full(const KshTables &TB, const KshLuts &LU, KshWaveLds &W, int lane, int me,
                                            bool &in_check, unsigned long long *dbg = nullptr) {
    uint8_t *st8 = ST8(W);
    BB3 cand;
    int n = ksh_gen(TB, LU, W, lane, me, true, true, in_check, cand, dbg);
#ifdef KSH_STAMPS
    if (lane == 0 && dbg) { dbg[10] = __builtin_amdgcn_s_memtime(); dbg[9] = bb_any(cand) ? 1 : 0; }
#endif
    for (int it = 0; it < 81 && bb_any(cand); it++) {  // wave-uniform; one iteration in reachable positions
        const int s = bb_nearest(cand, true);
        cand = bb_andn(cand, bb_sq(s));
        if (lane == 0) st8[s] = (uint8_t)(1 + 14 * me);
        wave_lds_sync();
        bool gives_check;
        BB3 none;
        const int replies = ksh_gen(TB, LU, W, lane, 1 - me, false, false, gives_check, none);
        if (lane == 0) st8[s] = 0;
        wave_lds_sync();
        if (!(gives_check && replies == 0)) {
            const int bit = 12960 + 7 * s;
            if (lane == 0) atomicOr(&W.maskbits[bit >> 5], 1u << (bit & 31));
            n += 1;
        }
    }
    wave_lds_sync();
    return n;
}

// Order of one step, chosen around two hardware facts: (1) the kernel is HBM-write bound at scale, so the
// observation stores leave as soon as the move is applied, before the move generator runs; (2) on gfx9 loads
// and stores share one in-order vmcnt, so every global LOAD of a wave (state, action, history, tables) is issued
// AND consumed before its first observation store.  Since round 3 the tables are staged into LDS first thing (the
// observation writer reads its expansion tables from there); measured neutral against staging them after the
// observation (profiles/r03_env_variants.log).
template <int OBS16, int TABG>
__device__ __forceinline__ void ksh_env_step_body(const KshStepArgs &a) {
    __shared__ typename KshLdsSel<TABG>::type L;
    const KshTables &T = ksh_tab_of(L, a.tables);
    const KshLuts &LU = ksh_lut_of(L);
#ifdef KSH_STAMPS
    const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: makes g and every address built from it uniform
    const int g = blockIdx.x * KSH_WPB + wv;
    const bool live = g < a.n_games;
    KshWaveLds &W = L.wave[wv];
    uint8_t *st8 = ST8(W);

    // ---- all loads up front: this game's state + action, then the block's share of the attack tables
    uint32_t st_word = 0;
    long long act = -1;
    unsigned long long hv[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // prefetched history keys
    unsigned long long *hrow = nullptr;
    if (live) {
        if (lane < 32) st_word = reinterpret_cast<const uint32_t *>(a.state + (size_t)g * 128)[lane];
        if (a.mode == 2) {
            act = a.actions[g];
        }
    }
    // table loads are issued now and land in registers while the state is decoded and the move applied; they go to LDS
    // right before the first consumer (the observation writer's expansion tables)
    constexpr int NT = 64 * KSH_WPB;
    constexpr int NV = (int)(sizeof(KshTables) / 16);      // the whole block: 6 full rounds of 256 threads + a remainder
    constexpr int NL = (int)(sizeof(KshLuts) / 16);        // the expansion tables alone: one partial round
    static_assert(NV >= 6 * NT && NV < 7 * NT, "staging below is written for 6 full rounds + a remainder");
    static_assert(NL <= NT, "one round");
    uint4 t0 = {0u, 0u, 0u, 0u}, t1 = t0, t2 = t0, t3 = t0, t4 = t0, t5 = t0, t6 = t0;
    const bool trem = (int)threadIdx.x < NV - 6 * NT;
    if constexpr (!TABG) {
        const uint4 *tsrc = reinterpret_cast<const uint4 *>(a.tables) + threadIdx.x;
        t0 = tsrc[0]; t1 = tsrc[NT]; t2 = tsrc[2 * NT]; t3 = tsrc[3 * NT]; t4 = tsrc[4 * NT]; t5 = tsrc[5 * NT];
        if (trem) t6 = tsrc[6 * NT];
    } else {
        if ((int)threadIdx.x < NL) t0 = reinterpret_cast<const uint4 *>(&a.tables->lut)[threadIdx.x];
    }

    int out_done = 0, out_reason = 0, out_winner = -1, n_out = 0;
    bool obs_written = false, need_gen = true, moved = false;
    float out_reward = 0.0f;
    int me = 0, mover = 0, mc = 0, rep = 1, mf_idx = -1;
    float mf_hint = 0.0f;
    unsigned long long hkey = 0;   // this move's key
    int k_from = -1, k_to = 0, k_pcf = 0, k_pct = 0, k_newpc = 0, k_hslot = -1, k_hold = 0, k_hnew = 0;   // the applied move (late key update)
    int hidx = 0;
    bool hist_ok = false;

    if (live) {
        // ---- state record into LDS, clear the legal bit-set
        if (lane < 32) W.st32[lane] = st_word;
        for (int i = lane; i < KSH_MASKW; i += 64) W.maskbits[i] = 0;
        wave_lds_sync();
#ifdef KSH_STAMPS
        if (lane == 0 && a.stamps) a.stamps[(size_t)g * 32 + 0] = t_entry;
#endif
        KSH_STAMP(1);  // state in LDS

        // ---- history prefetch (step mode): this move appends entry hidx = move_count - base; the first 512
        // earlier entries go to registers now (max_moves = 500 never needs more)
        hidx = (int)(st8[S_MC] | (st8[S_MC + 1] << 8)) - (int)(st8[S_HBASE] | (st8[S_HBASE + 1] << 8));
        hist_ok = a.mode == 2 && hidx >= 0 && hidx < a.hist_cap;
        hrow = a.hist + (size_t)g * (size_t)a.hist_cap;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (hist_ok && 64 * k + lane < hidx) hv[k] = hrow[64 * k + lane];

        if (a.mode == 1) {
            const bool rs = a.reset_mask == nullptr || a.reset_mask[g] != 0;
            if (rs) {
                if (lane < 32) W.st32[lane] = reinterpret_cast<const uint32_t *>(a.tables->init_state)[lane];
                wave_lds_sync();
            }
        }
        me = st8[S_SIDE];
        if (a.mode == 2) {
            mover = me;
            const bool over = st8[S_OVER] != 0;
            if (over) {
                // stepping a finished game (no auto-reset): nothing moves (shogi_game.py:589-593)
                const int w = (int)st8[S_WINNER] - 1;
                out_done = 1;
                out_reason = st8[S_REASON];
                out_winner = w;
                out_reward = (w == 1 - me) ? 1.0f : (w < 0 ? 0.0f : -1.0f);
            } else {
                bool ok = act >= 0 && act < KSH_A;
                if (ok && a.prev_mask != nullptr) ok = a.prev_mask[(size_t)g * KSH_A + (size_t)act] != 0;
                if (ok && a.prev_bits != nullptr)
                    ok = ((a.prev_bits[(size_t)g * KSH_BITROW + (size_t)(act >> 5)] >> (int)(act & 31)) & 1u) != 0;
                int from = -1, to = 0, aux = 0, pcf = 0, pct = 0;
                if (ok) {
                    const int ai = (int)act;
                    if (ai >= 12960) {
                        const int k = ai - 12960;
                        to = k / 7;
                        aux = k - 7 * to;
                        ok = st8[to] == 0 && st8[S_HAND + 7 * mover + aux] != 0;
                    } else {
                        const int pair = ai >> 1;
                        aux = ai & 1;
                        from = pair / 80;
                        const int j = pair - 80 * from;
                        to = j + (j >= from ? 1 : 0);
                        pcf = st8[from];
                        pct = st8[to];
                        ok = pcf != 0 && ((pcf >= 15) == (mover == 1)) && !(pct != 0 && ((pct >= 15) == (mover == 1)));
                        const int tf = pcf - 1 - (pcf >= 15 ? 14 : 0);
                        if (aux && !(tf <= 3 || tf == 5 || tf == 6)) ok = false;
                    }
                }
                if (!ok) {
                    // the reference raises ValueError and StepManager resets the episode (step_manager.py:303-348)
                    out_done = 1;
                    out_reason = 5;
                    if (a.auto_reset) {
                        if (lane < 32) W.st32[lane] = reinterpret_cast<const uint32_t *>(a.tables->init_state)[lane];
                        wave_lds_sync();
                        me = 0;
                    }
                } else {
                    // ---- apply the move on the LDS board (shogi_move_execution.py:24-156): every lane computes the same values,
                    // lane 0 writes the record.  The position key is updated AFTER the observation stores have been issued (they
                    // do not need it): -0.45 us at 4096 games, profiles/r03_env_variants.log
                    mf_hint = *reinterpret_cast<const float *>(st8 + S_NFRAC);
                    mc = (st8[S_MC] | (st8[S_MC + 1] << 8)) + 1;
                    mf_idx = mc;
                    int hslot = -1, hold = 0, hnew = 0, newpc;
                    if (from < 0) {
                        newpc = 1 + aux + 14 * mover;
                        hslot = 7 * mover + aux;
                        hold = st8[S_HAND + hslot];
                        hnew = hold - 1;
                    } else {
                        newpc = pcf;
                        if (aux) {
                            const int tf = pcf - 1 - 14 * mover;
                            newpc = 1 + 14 * mover + (tf <= 3 ? tf + 8 : tf + 7);
                        }
                        if (pct != 0) {
                            int tt = pct - 1 - (pct >= 15 ? 14 : 0);
                            if (tt >= 8) tt = (tt >= 12) ? tt - 7 : tt - 8;  // +P..+S -> P..S, +B,+R -> B,R
                            if (tt < 7) {                                    // a captured king is not kept
                                hslot = 7 * mover + tt;
                                hold = st8[S_HAND + hslot];
                                hnew = hold + 1;
                            }
                        }
                    }
                    k_from = from; k_to = to; k_pcf = pcf; k_pct = pct; k_newpc = newpc; k_hslot = hslot; k_hold = hold; k_hnew = hnew;
                    wave_lds_sync();
                    if (lane == 0) {
                        st8[to] = (uint8_t)newpc;
                        if (from >= 0) st8[from] = 0;
                        if (hslot >= 0) st8[S_HAND + hslot] = (uint8_t)hnew;
                        st8[S_MC] = (uint8_t)(mc & 0xFF);
                        st8[S_MC + 1] = (uint8_t)(mc >> 8);
                        st8[S_SIDE] = (uint8_t)(1 - mover);
                    }
                    wave_lds_sync();
                    me = 1 - mover;
                    moved = true;
                    KSH_STAMP(3);  // move applied
                }
            }
        }
    }

    // ---- tables into LDS (their loads were issued at the top); the only block barrier, reached by every wave
    if constexpr (!TABG) {
        uint4 *tdst = reinterpret_cast<uint4 *>(const_cast<KshTables *>(&T)) + threadIdx.x;
        tdst[0] = t0; tdst[NT] = t1; tdst[2 * NT] = t2; tdst[3 * NT] = t3; tdst[4 * NT] = t4; tdst[5 * NT] = t5;
        if (trem) tdst[6 * NT] = t6;
    } else {
        // attack tables stay in global memory; only the 3.3 KB of expansion tables go to LDS
        if ((int)threadIdx.x < NL) reinterpret_cast<uint4 *>(const_cast<KshLuts *>(&LU))[threadIdx.x] = t0;
    }
    __syncthreads();
    if (!live) return;  // wave-uniform
    KSH_STAMP(2);  // tables staged

    if (moved) {
        // ---- append the key to the game's history and count its earlier occurrences (shogi_rules_logic.py:638-695).  The
        // keys were prefetched at the top; the verdict is only needed for the termination test after the move generator.
        // Where the scan runs was measured (profiles/r03_env_variants.log, 4096 games): after the generator 24.2 us; between
        // the observation stores and the generator 24.8 us (loads and stores share one in-order vmcnt, and behind the store
        // rounds' branches the compiler can only wait with vmcnt(0), i.e. for every store to be acknowledged); before the
        // observation stores 25.8 us (the keys have not landed yet).  Scanning late keeps 16 registers alive across the
        // generator (78 VGPRs), which the LDS-table kernel can afford (4 waves per SIMD either way); the global-table
        // kernel, whose point is 7 waves per SIMD, scans before the generator.
        auto scan_history = [&]() {
            if (hist_ok) {
                if (lane == 0) hrow[hidx] = hkey;
                int matches = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) matches += __popcll(__ballot(64 * k + lane < hidx && hv[k] == hkey));
                for (int base = 512; base < hidx; base += 64) {  // bounded: hidx < hist_cap; only for max_moves > 512
                    const int i = base + lane;
                    const bool m = (i < hidx) && hrow[i] == hkey;
                    matches += __popcll(__ballot(m));
                }
                rep += matches;
            }
        };
        // ---- observation of the post-move state: issued NOW so that the stores drain while the move generator runs;
        // rewritten below in the rare case the game ends and is reset
        if (OBS16) ksh_write_obs16(W, LU, a, g, lane, mf_idx, mf_hint);
        else ksh_write_obs(W, LU, a, g, lane, mf_idx, mf_hint);
        obs_written = true;
        KSH_STAMP(5);  // obs stores issued
        {
            // ---- position key of the new state (ksh_tables.h): XOR out / in the items the move touched
            unsigned long long h = *reinterpret_cast<const unsigned long long *>(st8 + S_HASH) ^ KSH_SIDE_KEY;
            if (k_from < 0) {
                h ^= ksh_mix_sq(k_to, k_newpc);
            } else {
                h ^= ksh_mix_sq(k_from, k_pcf) ^ ksh_mix_sq(k_to, k_newpc);
                if (k_pct != 0) h ^= ksh_mix_sq(k_to, k_pct);
            }
            if (k_hslot >= 0) {
                if (k_hold > 0) h ^= ksh_mix_hand(k_hslot, k_hold);
                if (k_hnew > 0) h ^= ksh_mix_hand(k_hslot, k_hnew);
            }
            wave_lds_sync();
            if (lane == 0) *reinterpret_cast<unsigned long long *>(st8 + S_HASH) = h;
            wave_lds_sync();
            hkey = h;
        }
        if constexpr (TABG) scan_history();
        // ---- legal moves of the side now to move (+ nested uchi-fu-zume reply search)
        bool chk;
        unsigned long long *dbg = nullptr;
#ifdef KSH_STAMPS
        dbg = a.stamps ? a.stamps + (size_t)g * 32 + 16 : nullptr;
#endif
        const int n = ksh_gen_full(T, LU, W, lane, me, chk, dbg);
        KSH_STAMP(6);  // movegen done
        if constexpr (!TABG) scan_history();
        KSH_STAMP(4);  // history scanned
        need_gen = false;
        n_out = n;
        // ---- termination (shogi_game.py:408-450): no moves -> max moves -> sennichite
        if (n == 0) {
            out_done = 1;
            if (chk) { out_reason = 1; out_winner = mover; out_reward = 1.0f; }
            else { out_reason = 2; }
        } else if (mc >= a.max_moves) {
            out_done = 1;
            out_reason = 3;
        } else if (rep >= 4) {
            out_done = 1;
            out_reason = 4;
        }
        if (out_done) {
            if (a.auto_reset) {
                if (lane < 32) W.st32[lane] = reinterpret_cast<const uint32_t *>(T.init_state)[lane];
                for (int i = lane; i < KSH_MASKW; i += 64) W.maskbits[i] = 0;
                wave_lds_sync();
                me = 0;
                need_gen = true;
                obs_written = false;   // the observation must show the reset position
            } else if (lane == 0) {
                st8[S_OVER] = 1;
                st8[S_REASON] = (uint8_t)out_reason;
                st8[S_WINNER] = (uint8_t)(out_winner + 1);
            }
            wave_lds_sync();
        }
    }

    if (need_gen) {
        bool chk;
        int n = ksh_gen_full(T, LU, W, lane, me, chk);
        n_out = n;
        if (a.mode == 3 && st8[S_OVER] != 0) {          // already marked finished: report the stored verdict
            out_done = 1;
            out_reason = st8[S_REASON];
            out_winner = (int)st8[S_WINNER] - 1;
        } else if (a.mode == 3) {
            // termination of a freshly loaded position, as ShogiGame.from_sfen evaluates it (shogi_game.py:331-343):
            // no legal move -> Tsumi (winner = the side NOT to move) / stalemate; else move_count >= max_moves
            const int mcount = st8[S_MC] | (st8[S_MC + 1] << 8);
            if (n == 0) { out_done = 1; out_reason = chk ? 1 : 2; out_winner = chk ? 1 - me : -1; }
            else if (mcount >= a.max_moves) { out_done = 1; out_reason = 3; }
            if (out_done && lane == 0) {
                st8[S_OVER] = 1;
                st8[S_REASON] = (uint8_t)out_reason;
                st8[S_WINNER] = (uint8_t)(out_winner + 1);
            }
        }
    }
    wave_lds_sync();
    KSH_STAMP(7);  // termination / reset-regen done

    // ---- observation, unless it was already issued right after the move
    if (!obs_written) {
        if (OBS16) ksh_write_obs16(W, LU, a, g, lane, mf_idx, mf_hint);
        else ksh_write_obs(W, LU, a, g, lane, mf_idx, mf_hint);
    }
    KSH_STAMP(9);

    // ---- packed legal mask (optional): the LDS bit-set as it is, 106 aligned 16-byte stores per game
    if (a.mask_bits != nullptr) {
        uint32_t *brow = a.mask_bits + (size_t)g * KSH_BITROW;
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int c = lane + 64 * it;
            if (c < KSH_BITROW / 4) {
                uint4 v = {W.maskbits[4 * c], W.maskbits[4 * c + 1], W.maskbits[4 * c + 2], W.maskbits[4 * c + 3]};
                if (c == KSH_BITROW / 4 - 1) { v.z &= 0x007FFFFFu; v.w = 0u; }   // bits >= 13527 (dword 422 keeps 23) stay zero
                ksh_store16(brow + 4 * c, v);
            }
        }
    }
    // ---- legal mask: expand the bit-set to one byte per action, 16-byte aligned stores
    if (a.mask != nullptr) {
        uint8_t *mrow = a.mask + (size_t)g * KSH_A;
        const size_t addr = reinterpret_cast<size_t>(mrow);
        const int head = (int)((16 - (addr & 15)) & 15);        // bytes before the first aligned chunk
        const int nfull = (KSH_A - head) >> 4;
        const int tail0 = head + 16 * nfull;                     // first byte of the ragged tail
        if (lane < head) mrow[lane] = (uint8_t)((W.maskbits[0] >> lane) & 1u);
        {
            // round `it` of lane l covers the 16 actions 1024 ids after round it-1's: the bit-set word index advances by 32,
            // the shift never changes, and each byte of the 16 bits becomes eight 0/1 bytes through one 8-byte LDS lookup
            const int b0 = head + 16 * lane;
            const int sh = b0 & 31;
            const uint32_t *mw = &W.maskbits[b0 >> 5];
            const uint2 *b8 = reinterpret_cast<const uint2 *>(LU.byte8);
            uint8_t *m0 = mrow + b0;
#pragma unroll
            for (int it = 0; it < 14; it++) {                     // 14 * 64 >= 845: constant trip count
                if (lane + 64 * it < nfull) {
                    const uint32_t x = __funnelshift_r(mw[32 * it], mw[32 * it + 1], sh);  // 32 bits starting at this chunk's first id
                    const uint2 lo = b8[x & 0xFFu], hi = b8[(x >> 8) & 0xFFu];
                    ksh_store16(m0 + 1024 * it, uint4{lo.x, lo.y, hi.x, hi.y});
                }
            }
        }
        const int ntail = KSH_A - tail0;
        if (lane < ntail) {
            const int b = tail0 + lane;
            mrow[b] = (uint8_t)((W.maskbits[b >> 5] >> (b & 31)) & 1u);
        }
    }
    KSH_STAMP(10);  // mask stores issued

    // ---- with every output store in flight: finish the record for the next step and write it back
    if (lane == 0) {
        const int mcn = (st8[S_MC] | (st8[S_MC + 1] << 8)) + 1;
        *reinterpret_cast<float *>(st8 + S_NFRAC) = a.max_moves > 0 ? (float)((double)mcn / (double)a.max_moves) : 0.0f;
        a.n_legal[g] = n_out;
        if (a.mode == 2 || (a.mode == 3 && a.done != nullptr)) {
            if (a.reward != nullptr) a.reward[g] = out_reward;
            a.done[g] = (uint8_t)out_done;
            a.reason[g] = (uint8_t)out_reason;
            a.winner[g] = (signed char)out_winner;
        }
    }
    wave_lds_sync();
    if (lane < 32) reinterpret_cast<uint32_t *>(a.state + (size_t)g * 128)[lane] = W.st32[lane];
}

// Two entry points over one body: the reference contract (float32 planes, NCHW) and the compact observation format.
extern "C" __global__ void __launch_bounds__(64 * KSH_WPB) ksh_env_step_kernel(KshStepArgs a) { ksh_env_step_body<0, 0>(a); }
extern "C" __global__ void __launch_bounds__(64 * KSH_WPB) ksh_env_step_c16_kernel(KshStepArgs a) { ksh_env_step_body<1, 0>(a); }
// same step, attack tables read from global memory instead of an LDS copy (ksh_env_set_table_mode, include/keisei_hip.h)
extern "C" __global__ void __launch_bounds__(64 * KSH_WPB) ksh_env_step_gt_kernel(KshStepArgs a) { ksh_env_step_body<0, 1>(a); }
