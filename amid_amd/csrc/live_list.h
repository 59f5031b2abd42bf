// The live-sequence list of a batch as a device function of one 256-thread workgroup (embed.hip: the packing launches; adam.hip: the
// one-launch step head): live[0 .. n0) = batch rows whose own domain is 0 (ascending), live[n0 .. B) = domain 1's, live[B] = n0.
#pragma once
#include "common.h"

namespace amid {

// body for one 256-thread workgroup
__device__ __forceinline__ void live_list_block(const long long* __restrict__ domain, int B, int* __restrict__ live) {
    __shared__ int tot[2][4];
    const int lane = lane_id(), w = wave_id();
    const int chunks = (B + 63) / 64, per = (chunks + 3) / 4;          // chunks of 64 batch rows, `per` consecutive chunks per wave
    const int c_beg = w * per, c_end = min(chunks, c_beg + per);
    int n0w = 0, n1w = 0;
    for (int c = c_beg; c < c_end; ++c) {
        const int b = c * 64 + lane;
        const bool in = b < B;
        const bool d = in && domain[b] != 0;
        n0w += __popcll(__ballot(in && !d));
        n1w += __popcll(__ballot(d));
    }
    if (lane == 0) { tot[0][w] = n0w; tot[1][w] = n1w; }
    __syncthreads();
    const int n0 = tot[0][0] + tot[0][1] + tot[0][2] + tot[0][3];
    int off0 = 0, off1 = n0;
    for (int ww = 0; ww < w; ++ww) { off0 += tot[0][ww]; off1 += tot[1][ww]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int c = c_beg; c < c_end; ++c) {
        const int b = c * 64 + lane;
        const bool in = b < B;
        const bool d = in && domain[b] != 0;
        const unsigned long long m0 = __ballot(in && !d), m1 = __ballot(d);
        if (in && !d) live[off0 + __popcll(m0 & below)] = b;
        if (d) live[off1 + __popcll(m1 & below)] = b;
        off0 += __popcll(m0);
        off1 += __popcll(m1);
    }
    if (threadIdx.x == 0) live[B] = n0;
}

}  // namespace amid
