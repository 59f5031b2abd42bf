// fixed-order sum of partial buffers: dst[e] = sum_k src[k * stride + e]  (shared by sasrec_bwd.hip and segreduce.hip)
#pragma once
#include "common.h"

namespace amid {

struct ReduceEntry { const float* src; float* dst; long long stride; int n_part; int count; };
// (sink: what else happens to a finished slice -- nothing, or the folded optimizer's Adam on the spot: tail_parts.h NoSink, adam.hip)
struct ReduceNoSink {
    __device__ __forceinline__ void quad(float*, float4) const {}
    __device__ __forceinline__ void one(float*, float) const {}
};

// 256 threads = 32 consecutive elements (or float4s) x 8 partial groups; group pg sums partials pg, pg+8, ... with four (eight when
// there are many partials) independent loads in flight, then the eight group sums are added in group order (fixed order => reproducible).
// (bx, nbx): this block's index / the number of blocks along the element axis of entry `en`.
template <class Sink = ReduceNoSink>
__device__ __forceinline__ void reduce_partials_block(const ReduceEntry en, int bx, int nbx, const Sink sink = Sink()) {
    __shared__ float4 red[8][33];
    const int el = threadIdx.x & 31, pg = threadIdx.x >> 5;
    // entries whose rows are whole, aligned float4s (every weight / bias matrix) move 16 bytes per lane: a wave covers two
    // 512-byte runs instead of two 128-byte ones.  Same partial order per element as the scalar path => the same bits.
    const bool vec = ((en.count | (int)(en.stride & 3)) & 3) == 0 && ((((unsigned long long)en.src) | ((unsigned long long)en.dst)) & 15) == 0;
    if (vec && en.n_part <= 32) {
        // few partials (weight-gradient splits, tile partials of a short batch): a thread owns one float4 and adds the partials in
        // order, every load in flight at once -- 1024 elements per block pass, no LDS, no barrier
        for (int e = (bx * 256 + (int)threadIdx.x) * 4; e < en.count; e += nbx * 1024) {
            const float* __restrict__ p = en.src + e;
            float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k0 = 0; k0 < en.n_part; k0 += 8) {      // eight loads in flight, added in partial order
                float4 r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (k0 + j < en.n_part) ? ld4(p + (long long)(k0 + j) * en.stride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) s4 = f4add(s4, r[j]);
            }
            st4(en.dst + e, s4);
            sink.quad(en.dst + e, s4);
        }
        return;
    }
    if (vec) {
        for (int e0 = bx * 128; e0 < en.count; e0 += nbx * 128) {     // block-uniform
            const int e = e0 + 4 * el;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < en.count) {
                const float* __restrict__ p = en.src + e;
                int k = pg;
                for (; k + 56 < en.n_part; k += 64) {      // many partials (per-row partials of the head: one per batch row): 8 in flight
                    float4 r[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] = ld4(p + (long long)(k + 8 * j) * en.stride);
#pragma unroll
                    for (int j = 0; j < 8; ++j) s = f4add(s, r[j]);
                }
                for (; k + 24 < en.n_part; k += 32) {
                    const float4 a = ld4(p + (long long)k * en.stride), b = ld4(p + (long long)(k + 8) * en.stride);
                    const float4 c = ld4(p + (long long)(k + 16) * en.stride), d = ld4(p + (long long)(k + 24) * en.stride);
                    s = f4add(s, a); s = f4add(s, b); s = f4add(s, c); s = f4add(s, d);
                }
                for (; k < en.n_part; k += 8) s = f4add(s, ld4(p + (long long)k * en.stride));
            }
            red[pg][el] = s;
            __syncthreads();
            if (pg == 0 && e < en.count) {
                float4 t = red[0][el];
#pragma unroll
                for (int g = 1; g < 8; ++g) t = f4add(t, red[g][el]);
                st4(en.dst + e, t);
                sink.quad(en.dst + e, t);
            }
            __syncthreads();
        }
        return;
    }
    float* reds = reinterpret_cast<float*>(&red[0][0]);               // scalar path: [8][33] floats of the same buffer
    for (int e0 = bx * 32; e0 < en.count; e0 += nbx * 32) {       // block-uniform
        const int e = e0 + el;
        float s = 0.f;
        if (e < en.count) {
            const float* __restrict__ p = en.src + e;
            int k = pg;
            for (; k + 56 < en.n_part; k += 64) {
                float r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = p[(long long)(k + 8 * j) * en.stride];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += r[j];
            }
            for (; k + 24 < en.n_part; k += 32) {
                const float a = p[(long long)k * en.stride], b = p[(long long)(k + 8) * en.stride];
                const float c = p[(long long)(k + 16) * en.stride], d = p[(long long)(k + 24) * en.stride];
                s += a; s += b; s += c; s += d;
            }
            for (; k < en.n_part; k += 8) s += p[(long long)k * en.stride];
        }
        reds[pg * 33 + el] = s;
        __syncthreads();
        if (pg == 0 && e < en.count) {
            float t = reds[el];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += reds[g * 33 + el];
            en.dst[e] = t;
            sink.one(en.dst + e, t);
        }
        __syncthreads();
    }
}

}  // namespace amid
