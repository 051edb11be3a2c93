// ksh_device_util.h -- small device helpers shared by the learner-side and tower kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 ksh_bf16;

__device__ __forceinline__ float ksh_u2f(uint32_t u) {
    union { uint32_t u; float f; } c;
    c.u = u;
    return c.f;
}
__device__ __forceinline__ float ksh_ld(const float *p) { return *p; }
__device__ __forceinline__ float ksh_ld(const ksh_bf16 *p) {
    return ksh_u2f((uint32_t)(*reinterpret_cast<const uint16_t *>(p)) << 16);
}
__device__ __forceinline__ void ksh_st(float *p, float v) { *p = v; }
__device__ __forceinline__ void ksh_st(ksh_bf16 *p, float v) { *p = (ksh_bf16)v; }
__device__ __forceinline__ uint16_t ksh_bf16_bits(float v) {
    ksh_bf16 b = (ksh_bf16)v;
    return *reinterpret_cast<uint16_t *>(&b);
}

