// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libamid_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AMID_OK 0
#define AMID_ERR_ARG (-1)        // bad argument (null pointer, unsupported size)
#define AMID_ERR_UNSUPPORTED (-2)  // shape outside what the kernels are built for

#define AMID_CHECK_ARG(cond) do { if (!(cond)) return AMID_ERR_ARG; } while (0)
#define AMID_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

namespace amid {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// wave index made provably wave-uniform for the compiler (scalar addressing, no waterfall loops)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// butterfly reductions over `width` consecutive lanes (width = power of two <= 64)
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// store through an explicitly GLOBAL pointer (a pointer that went through a struct can lose its address space: flat_store counts on
// lgkmcnt too and forces full waits in front of LDS reads)
typedef float amid_gv4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_global(float* p, float4 v) {
    amid_gv4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(__attribute__((address_space(1))) amid_gv4*)(p) = t;
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float f4hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

}  // namespace amid
