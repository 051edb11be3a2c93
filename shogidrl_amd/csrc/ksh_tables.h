// ksh_tables.h -- host-side construction of the lookup tables the env kernel stages in LDS.
//
// Bitboard convention (device and host): a set of squares is three 27-bit words, word k holding
// rows 3k..3k+2 (bit = sq - 27k, sq = row*9 + col, row 0 = White's back rank as in the reference,
// shogi/shogi_game.py:83-111).  Word 0 is therefore Black's promotion zone and word 2 White's
// (shogi_game.py:821-825).
//
// Ray directions d (opposite = d^1; odd d = increasing square index):
//   0 N(-1,0) 1 S(+1,0) 2 W(0,-1) 3 E(0,+1) 4 NW(-1,-1) 5 SE(+1,+1) 6 NE(-1,+1) 7 SW(+1,-1)
// Step classes: 0..3 Black P,N,S,G   4..7 White P,N,S,G   8 K   9 orthogonal king steps (+B extra,
//   shogi_rules_logic.py:135-140)   10 diagonal king steps (+R extra, :129-134)
#pragma once
#include <stdint.h>
#include <string.h>

// Small expansion tables, always staged into LDS (3.3 KB).  The env kernel is bound by its VALU instruction count at the
// 4096-game size, while the LDS pipe is nearly idle: every bit -> byte / bit -> float /
// bit-interleave expansion that used to be shifts, multiplies and selects is one LDS lookup.
struct KshLuts {
    uint32_t byte8[256][2];    // byte b -> eight bytes, byte i = bit i of b                  (legal-mask row: bits -> bool bytes)
    uint32_t nib_f32[16][4];   // nibble n -> four float32 words, word i = bit i ? 1.0f : 0.0f (observation planes 0-27)
    uint32_t spread8[256];     // byte b -> 16 bits with bit i of b at position 2i            (non-promoting / promoting ids interleave)
    float hand18[20];          // c -> (float)((double)c / 18.0): hand-count planes (shogi_game_io.py:513-527), c = 0..19
};
static_assert(sizeof(KshLuts) % 16 == 0, "staged as uint4");

struct KshTables {
    uint32_t ray[8][81][4];    // [dir][sq] -> {w0,w1,w2,0}: squares from sq (exclusive) to the edge
    uint32_t step[11][81][4];  // [class][sq] -> {w0,w1,w2,0}
    uint8_t init_state[128];   // packed initial position (shogi_game.py:79-130)
    uint8_t pad[16];
    KshLuts lut;               // last member: the table-less kernel variant stages only this part
};
static_assert(sizeof(KshTables) % 16 == 0, "tables are copied to LDS as uint4");

// ---- position key (sennichite).  The reference compares exact (board, hands, side) tuples
// (shogi/shogi_game.py:347-372); here a 64-bit XOR of mixed per-item keys stands in for it.  The kernel
// maintains it incrementally (a move touches at most five items); the host computes it from scratch for the
// initial position and for ksh_env_set_state.  State record: bytes 104-111 hash, 112-115 float
// (move_count+1)/max_moves (the move-count plane value of the NEXT position, so no step needs a division or
// a dependent table load before it can start writing the observation).
#define KSH_S_HASH 104
#define KSH_S_NFRAC 112
#define KSH_SIDE_KEY 0xA5A5F00D5EED1234ull

__host__ __device__ static inline uint32_t ksh_fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
__host__ __device__ static inline unsigned long long ksh_mix_key(uint32_t key) {
    const uint32_t a = ksh_fmix32(key * 0x9E3779B1u + 0x7F4A7C15u);
    const uint32_t b = ksh_fmix32((key ^ 0x5bd1e995u) * 0x2545F491u + 0x1B873593u);
    return ((unsigned long long)b << 32) | a;
}
// piece code pc (1..28) standing on square sq
__host__ __device__ static inline unsigned long long ksh_mix_sq(int sq, int pc) { return ksh_mix_key((uint32_t)(sq * 32 + pc)); }
// `cnt` (> 0) pieces of hand slot (0..13 = black P..R, white P..R)
__host__ __device__ static inline unsigned long long ksh_mix_hand(int slot, int cnt) {
    return ksh_mix_key(0x10000u + (uint32_t)slot * 256u + (uint32_t)cnt);
}
static inline unsigned long long ksh_full_hash(const uint8_t *rec) {
    unsigned long long h = 0;
    for (int s = 0; s < 81; s++) if (rec[s]) h ^= ksh_mix_sq(s, rec[s]);
    for (int k = 0; k < 14; k++) if (rec[81 + k]) h ^= ksh_mix_hand(k, rec[81 + k]);
    if (rec[95]) h ^= KSH_SIDE_KEY;
    return h;
}
// fills the derived fields (hash, next move fraction) of a packed state record
static inline void ksh_finish_record(uint8_t *rec, int max_moves) {
    const unsigned long long h = ksh_full_hash(rec);
    memcpy(rec + KSH_S_HASH, &h, 8);
    const int mc = rec[96] | (rec[97] << 8);
    const float f = max_moves > 0 ? (float)((double)(mc + 1) / (double)max_moves) : 0.0f;
    memcpy(rec + KSH_S_NFRAC, &f, 4);
}

static inline void ksh_bb_set(uint32_t *w, int sq) { w[sq / 27] |= 1u << (sq % 27); }

static inline void ksh_build_luts(KshLuts *u) {
    for (int b = 0; b < 256; b++) {
        uint32_t lo = 0, hi = 0, sp = 0;
        for (int i = 0; i < 4; i++) {
            lo |= (uint32_t)((b >> i) & 1) << (8 * i);
            hi |= (uint32_t)((b >> (4 + i)) & 1) << (8 * i);
        }
        for (int i = 0; i < 8; i++) sp |= (uint32_t)((b >> i) & 1) << (2 * i);
        u->byte8[b][0] = lo;
        u->byte8[b][1] = hi;
        u->spread8[b] = sp;
    }
    for (int n = 0; n < 16; n++)
        for (int i = 0; i < 4; i++) u->nib_f32[n][i] = ((n >> i) & 1) ? 0x3F800000u : 0u;
    for (int c = 0; c < 20; c++) u->hand18[c] = (float)((double)c / 18.0);
}

static inline void ksh_build_tables(KshTables *t) {
    memset(t, 0, sizeof(*t));
    ksh_build_luts(&t->lut);
    static const int DR[8] = {-1, 1, 0, 0, -1, 1, -1, 1};
    static const int DC[8] = {0, 0, -1, 1, -1, 1, 1, -1};
    for (int d = 0; d < 8; d++)
        for (int sq = 0; sq < 81; sq++) {
            int r = sq / 9 + DR[d], c = sq % 9 + DC[d];
            while (r >= 0 && r < 9 && c >= 0 && c < 9) {
                ksh_bb_set(t->ray[d][sq], r * 9 + c);
                r += DR[d];
                c += DC[d];
            }
        }
    for (int cls = 0; cls < 11; cls++)
        for (int sq = 0; sq < 81; sq++) {
            int offs[8][2];
            int n = 0;
            int fw = (cls >= 4 && cls < 8) ? 1 : -1;  // White moves toward larger rows (shogi_rules_logic.py:96-98)
            int k = (cls < 8) ? (cls & 3) : cls;
            auto add = [&](int a, int b) { offs[n][0] = a; offs[n][1] = b; n++; };
            if (cls < 8 && k == 0) { add(fw, 0); }
            else if (cls < 8 && k == 1) { add(2 * fw, -1); add(2 * fw, 1); }
            else if (cls < 8 && k == 2) { add(fw, 0); add(fw, -1); add(fw, 1); add(-fw, -1); add(-fw, 1); }
            else if (cls < 8 && k == 3) { add(fw, 0); add(fw, -1); add(fw, 1); add(0, -1); add(0, 1); add(-fw, 0); }
            else if (cls == 8) { for (int d = 0; d < 8; d++) add(DR[d], DC[d]); }
            else if (cls == 9) { for (int d = 0; d < 4; d++) add(DR[d], DC[d]); }
            else { for (int d = 4; d < 8; d++) add(DR[d], DC[d]); }
            for (int i = 0; i < n; i++) {
                int r = sq / 9 + offs[i][0], c = sq % 9 + offs[i][1];
                if (r >= 0 && r < 9 && c >= 0 && c < 9) ksh_bb_set(t->step[cls][sq], r * 9 + c);
            }
        }
    // initial position; piece code = 1 + type + 14*color, types P,L,N,S,G,B,R,K = 0..7
    uint8_t *b = t->init_state;
    static const int back[9] = {1, 2, 3, 4, 7, 4, 3, 2, 1};
    for (int i = 0; i < 9; i++) {
        b[2 * 9 + i] = 1 + 0 + 14;
        b[6 * 9 + i] = 1 + 0;
        b[0 * 9 + i] = (uint8_t)(1 + back[i] + 14);
        b[8 * 9 + i] = (uint8_t)(1 + back[i]);
    }
    b[1 * 9 + 1] = 1 + 6 + 14;
    b[1 * 9 + 7] = 1 + 5 + 14;
    b[7 * 9 + 1] = 1 + 5;
    b[7 * 9 + 7] = 1 + 6;
}
