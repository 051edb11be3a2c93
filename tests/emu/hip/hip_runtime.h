// TEST INFRASTRUCTURE: a fake <hip/hip_runtime.h> that lets g++ compile the *unmodified* HIP kernel
// sources of shogidrl_amd/csrc for the host.  Every thread of a block becomes a ucontext fiber;
// cross-lane intrinsics (__ballot, __shfl, wave barrier, __syncthreads) park the fiber until all live
// lanes of the wave (block) reach the same operation, which also verifies that such operations only
// occur under wave-uniform control flow and that every LDS hand-off between lanes has an ordering
// point.  Used by tests/test_emu_*.py only; the product never builds or loads this.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <math.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
struct EmuDim3 { int x, y, z; };

namespace emu {
enum Op { OP_NONE = 0, OP_BALLOT, OP_SHFL, OP_WAVE_BARRIER, OP_SYNCTHREADS, OP_MFMA32 };
struct MfmaOperands { float a, b; float *acc; };
struct Lane {
    ucontext_t ctx;
    char *stack;
    int tid;
    bool done, parked;
    int op;
    uint64_t arg;
    int arg2;
    uint64_t res;
};
struct Block {
    int nthreads;
    Lane *lanes;
    ucontext_t sched;
    int cur;
    EmuDim3 bidx, bdim;
    void (*entry)(void *);
    void *entry_arg;
};
extern Block *B;
inline Lane &me() { return B->lanes[B->cur]; }
uint64_t park(int op, uint64_t arg, int arg2);
void run_block(void (*entry)(void *), void *arg, int block_idx, int nthreads, int block_idy = 0, int block_idz = 0);
inline EmuDim3 tid() { return EmuDim3{B->lanes[B->cur].tid, 0, 0}; }
inline EmuDim3 bid() { return B->bidx; }
inline EmuDim3 bdim() { return B->bdim; }
extern EmuDim3 g_grid;
inline void set_grid_dim(int x, int y = 1) { g_grid = EmuDim3{x, y, 1}; }
inline EmuDim3 gdim() { return g_grid; }
}  // namespace emu

#define threadIdx (emu::tid())
#define blockIdx (emu::bid())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())

static inline unsigned long long __ballot(int pred) { return emu::park(emu::OP_BALLOT, pred ? 1 : 0, 0); }
template <class T> static inline T __shfl(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl");
    uint64_t raw = 0;
    memcpy(&raw, &v, sizeof(T));
    uint64_t r = emu::park(emu::OP_SHFL, raw, src & 63);
    T out;
    memcpy(&out, &r, sizeof(T));
    return out;
}
#define __syncthreads() ((void)emu::park(emu::OP_SYNCTHREADS, 0, 0))
#define __builtin_amdgcn_wave_barrier() ((void)emu::park(emu::OP_WAVE_BARRIER, 0, 0))
#define __builtin_amdgcn_fence(order, scope) ((void)0)

static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x == 0 ? 32 : __builtin_clz((unsigned)x); }
static inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned sh) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> (sh & 31));
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned atomicXor(unsigned *p, unsigned v) { unsigned o = *p; *p = o ^ v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }

template <class T> static inline T __shfl_xor(T v, int m) { return __shfl(v, (emu::me().tid & 63) ^ m); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32); }
// individually rounded float32 ops (volatile blocks x87/FMA style contraction on the host)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }

namespace emu {
struct bf16 {
    uint16_t x;
    bf16() = default;
    explicit bf16(float f) {  // round to nearest even, NaN stays NaN
        uint32_t u;
        memcpy(&u, &f, 4);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) { x = (uint16_t)((u >> 16) | 0x40); return; }
        x = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    explicit operator float() const { uint32_t u = (uint32_t)x << 16; float f; memcpy(&f, &u, 4); return f; }
};
}  // namespace emu
#define __bf16 emu::bf16
// value of the first lane of the wave (all lanes are active wherever the kernels use it)
#define __builtin_amdgcn_readfirstlane(x) (__shfl((int)(x), 0))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_readlane(x, l) (__shfl((int)(x), (l)))
// v_mfma_f32_32x32x2_f32: one rendez-vous of the wave; the scheduler computes D = A(32x2) B(2x32) + C for all lanes with
// the hardware's lane maps (A[i = l & 31][k = l >> 5], B[k = l >> 5][j = l & 31], D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)]
// [col = l & 31] in register r) and its arithmetic (a k-ordered fmaf chain).
typedef float emu_f32x16 __attribute__((vector_size(64)));
static inline emu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    float acc[16];
    for (int r = 0; r < 16; r++) acc[r] = c[r];
    emu::MfmaOperands ops = {a, b, acc};
    (void)emu::park(emu::OP_MFMA32, (uint64_t)(uintptr_t)&ops, 0);
    emu_f32x16 d;
    for (int r = 0; r < 16; r++) d[r] = acc[r];
    return d;
}
