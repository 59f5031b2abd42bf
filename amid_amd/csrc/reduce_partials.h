// fixed-order sum of partial buffers: dst[e] = sum_k src[k * stride + e]  (shared by sasrec_bwd.hip and segreduce.hip)
#pragma once
#include "common.h"

namespace amid {

struct ReduceEntry { const float* src; float* dst; long long stride; int n_part; int count; };

// 256 threads = 32 consecutive elements x 8 partial groups; group pg sums partials pg, pg+8, ... with four
// independent loads in flight, then the eight group sums are added in group order (fixed order => reproducible).
// (bx, nbx): this block's index / the number of blocks along the element axis of entry `en`.
__device__ __forceinline__ void reduce_partials_block(const ReduceEntry en, int bx, int nbx) {
    __shared__ float red[8][33];
    const int el = threadIdx.x & 31, pg = threadIdx.x >> 5;
    for (int e0 = bx * 32; e0 < en.count; e0 += nbx * 32) {       // block-uniform
        const int e = e0 + el;
        float s = 0.f;
        if (e < en.count) {
            const float* __restrict__ p = en.src + e;
            int k = pg;
            for (; k + 24 < en.n_part; k += 32) {
                const float a = p[(long long)k * en.stride], b = p[(long long)(k + 8) * en.stride];
                const float c = p[(long long)(k + 16) * en.stride], d = p[(long long)(k + 24) * en.stride];
                s += a; s += b; s += c; s += d;
            }
            for (; k < en.n_part; k += 8) s += p[(long long)k * en.stride];
        }
        red[pg][el] = s;
        __syncthreads();
        if (pg == 0 && e < en.count) {
            float t = red[0][el];
#pragma unroll
            for (int g = 1; g < 8; ++g) t += red[g][el];
            en.dst[e] = t;
        }
        __syncthreads();
    }
}

}  // namespace amid
