// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libamid_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define AMID_OK 0
#define AMID_ERR_ARG (-1)        // bad argument (null pointer, unsupported size)
#define AMID_ERR_UNSUPPORTED (-2)  // shape outside what the kernels are built for
#define AMID_FLAG_INDEX_RANGE (1)
#define AMID_FLAG_UMAX_EXCEEDED (2)

#define AMID_CHECK_ARG(cond) do { if (!(cond)) return AMID_ERR_ARG; } while (0)
#define AMID_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

namespace amid {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// The LDS byte offset of a pointer into the workgroup's shared memory, for instructions that take the address as a number (ds_read through
// inline asm, M0 of the LDS-DMA loads).  A flat LDS address is { shared aperture, offset }: the low word IS the offset.  (The address-space
// cast `(__attribute__((address_space(3))) T*)p` computes the same number behind a null check; where the compiler could not fold that check
// it compared the aperture register itself -- "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base", a crash of the gfx950
// backend that came and went with unrelated edits of sasrec_strip.hip.)
__device__ __forceinline__ unsigned lds_offset(const void* p) { return (unsigned)(unsigned long long)p; }
// wave index made provably wave-uniform for the compiler (scalar addressing, no waterfall loops)
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// All-reduce over `WIDTH` consecutive lanes (WIDTH = power of two <= 64).  Inside a row of 16 lanes the exchange is done by DPP
// (quad_perm, row_half_mirror, row_mirror: plain VALU operands, a few cycles each); rows are combined through v_readlane.  The
// __shfl_xor form (ds_bpermute_b32 through the LDS crossbar) cost ~200 cycles per dependent level -- 1 000 cycles for one
// 32-lane LayerNorm statistic, seven of them back to back in every row pass.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
struct SumOp { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct MaxOp { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };

template <int WIDTH, class OP>
__device__ __forceinline__ float group_allreduce(float v, OP op) {
    if constexpr (WIDTH >= 2) v = op(v, dpp_move<0xB1>(v));           // quad_perm [1,0,3,2]  : lane ^ 1
    if constexpr (WIDTH >= 4) v = op(v, dpp_move<0x4E>(v));           // quad_perm [2,3,0,1]  : lane ^ 2
    if constexpr (WIDTH >= 8) v = op(v, dpp_move<0x141>(v));          // row_half_mirror      : i <-> 7 - i  (the other quad)
    if constexpr (WIDTH >= 16) v = op(v, dpp_move<0x140>(v));         // row_mirror           : i <-> 15 - i (the other half row)
    if constexpr (WIDTH == 32) {
        const float r0 = lane_value(v, 0), r1 = lane_value(v, 16), r2 = lane_value(v, 32), r3 = lane_value(v, 48);
        v = (threadIdx.x & 32) ? op(r2, r3) : op(r0, r1);
    }
    if constexpr (WIDTH == 64) {
        const float r0 = lane_value(v, 0), r1 = lane_value(v, 16), r2 = lane_value(v, 32), r3 = lane_value(v, 48);
        v = op(op(r0, r1), op(r2, r3));
    }
    return v;
}
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) { return group_allreduce<WIDTH>(v, SumOp()); }
template <int WIDTH>
__device__ __forceinline__ float group_max(float v) { return group_allreduce<WIDTH>(v, MaxOp()); }

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

// dynamic LDS above 64 KB needs the kernel's limit raised -- once per DEVICE (the attribute belongs to the device's code object: a
// process-wide flag left a second device of the same process at the default limit); `done`: the caller's per-kernel device mask
static inline int lds_attr_once(const void* kern, size_t bytes, unsigned long long& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 64 && ((done >> dev) & 1ull)) return AMID_OK;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
    if (dev < 64) done |= 1ull << dev;
    return AMID_OK;
}

}  // namespace amid
