// What a v_mfma_f32_16x16x32_bf16 costs a wave that is alone on its SIMD: 96 instructions (one pass of a strip's piece product) in four
// arrangements, timed with s_memtime inside the kernel.  hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mma(const v4u& a, const v4u& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int MODE> __global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, const unsigned* src) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    v4u a[3], b[8];
    for (int i = 0; i < 3; ++i) a[i] = v4u{src[lane + i], src[lane + 64 + i], src[lane + 128 + i], src[lane + 192 + i]};
    for (int i = 0; i < 8; ++i) b[i] = v4u{src[lane + 3 * i], src[lane + 70 + i], src[lane + 140 + i], src[lane + 210 + i]};
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // every operand in its registers before the clock starts (an asm that takes and returns them: the loads cannot sink below it)
    for (int i = 0; i < 3; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[i]));
    for (int i = 0; i < 8; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[i]));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 0) {               // one accumulator: 96 dependent instructions
#pragma unroll
        for (int t = 0; t < 96; ++t) acc[0] = mma(b[t & 7], a[t % 3], acc[0]);
    } else if constexpr (MODE == 1) {        // eight accumulators in rotation
#pragma unroll
        for (int t = 0; t < 96; ++t) acc[t & 7] = mma(b[t & 7], a[t % 3], acc[t & 7]);
    } else if constexpr (MODE == 2) {        // three in a row per accumulator (the strips' first form)
#pragma unroll
        for (int t = 0; t < 32; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[t & 7] = mma(b[t & 7], a[p], acc[t & 7]);
    } else {                                  // MODE 3: as 2, the weight fragment of every step read from LDS four steps ahead
        const unsigned base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + lane * 16;
        f32x4 w[5];
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[u]) : "v"(base), "n"(u * 1024) : "memory");
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t + 4 < 32) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w[(t + 4) % 5]) : "v"(base), "n"(((t + 4) % 28) * 1024) : "memory");
            if (t + 4 < 32) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w[t % 5]));
            else if (t == 28) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w[t % 5]));
            else if (t == 29) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w[t % 5]));
            else if (t == 30) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w[t % 5]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[t % 5]));
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[t & 7] = mma(__builtin_bit_cast(v4u, w[t % 5]), a[p], acc[t & 7]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 0 && lane == 0) { out[threadIdx.x >> 6] = t1 - t0; out[4 + (threadIdx.x >> 6)] = r1 - r0; }
}
int main() {
    unsigned long long* out; float* sink; unsigned* src;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 256 * 4); hipMalloc(&src, 4096); hipMemset(src, 0x3c, 4096);
    const char* names[4] = {"96 on one accumulator", "96 over eight accumulators in rotation", "32 x 3 in a row per accumulator",
                            "32 x 3 in a row, fragment from LDS four steps ahead"};
    for (int grid : {256, 1})
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) probe<0><<<grid, 256, 32768>>>(out, sink, src);
            if (mode == 1) probe<1><<<grid, 256, 32768>>>(out, sink, src);
            if (mode == 2) probe<2><<<grid, 256, 32768>>>(out, sink, src);
            if (mode == 3) probe<3><<<grid, 256, 32768>>>(out, sink, src);
            unsigned long long h[8];
            hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            if (rep == 2) printf("%3d workgroups  %-58s %5llu s_memtime ticks (%.2f us on the 100 MHz clock) for 96 MFMAs = %.1f ticks per instruction\n", grid, names[mode], h[0], h[4] / 100.0, h[0] / 96.0);
        }
    return 0;
}
