// bf16 fragment images of square [D][D] fp32 matrices (see amid_sas_weights_bf16 in sasrec_seq.hip for the layout): one 256-thread block's
// share of one matrix -- shared by the standalone launch and by the riders of the gather K1 (embed.hip).
#pragma once
#include "common.h"
#include "bf16_pieces.h"

namespace amid {

// chunk q = n * (D / 8) + c of the image: row n, chunk c = 4 s + g = W[n][32 s + 4 g + 0..3], W[n][32 s + 16 + 4 g + 0..3];
// planes = 3: every element as hi + mid + lo, one image per piece ([3][D][D] bf16 per matrix)
// ld: floats between the source's rows (0: D -- a [D][D] matrix of its own; BERT4Rec's feed-forward weights are walked as D x D blocks of
// a [512][128] / [128][512] matrix: bert_strip.hip)
__device__ __forceinline__ void weights_image_block(const float* __restrict__ W, unsigned short* __restrict__ out, int D, int transposed,
                                                    int planes, int block, int nblocks, int ld = 0) {
    if (ld == 0) ld = D;
    const int cpr = D / 8;
    for (int q = block * 256 + threadIdx.x; q < D * cpr; q += nblocks * 256) {
        const int n = q / cpr, c = q % cpr;
        const int s = c >> 2, g = c & 3;
        unsigned pk[3][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 32 * s + 16 * h + 4 * g + 2 * e;
                const float v0 = transposed ? W[(size_t)k * ld + n] : W[(size_t)n * ld + k];
                const float v1 = transposed ? W[(size_t)(k + 1) * ld + n] : W[(size_t)n * ld + k + 1];
                const WgSplit2 sp = wg_split3(v0, v1);        // (hi = the round-to-nearest-even bf16 pair of the one-plane image)
                pk[0][2 * h + e] = sp.hi; pk[1][2 * h + e] = sp.mid; pk[2][2 * h + e] = sp.lo;
            }
        for (int p = 0; p < planes; ++p)
            *reinterpret_cast<uint4*>(out + (size_t)p * D * D + (size_t)q * 8) = make_uint4(pk[p][0], pk[p][1], pk[p][2], pk[p][3]);
    }
}

// a launch's weight-image riders (embed.hip: the gather K1; adam.hip: the step head): n tiles of D x D floats -- tile i = src[i][r * ld[i] + c]
// (tr[i] = 0) or its transpose; the first n_fwd images go to dst, the others to dstT; `per` workgroups of 256 threads per tile
constexpr int W16_MAX = 96;
struct W16Rider { const float* src[W16_MAX]; unsigned short ld[W16_MAX]; unsigned char tr[W16_MAX]; unsigned short* dst; unsigned short* dstT; int n, n_fwd, planes, per, D; };
__device__ __forceinline__ void w16_rider_block(const W16Rider& wr, int blk) {
    const int wi = blk / wr.per;
    unsigned short* out = wi < wr.n_fwd ? wr.dst + (size_t)wi * wr.planes * wr.D * wr.D : wr.dstT + (size_t)(wi - wr.n_fwd) * wr.planes * wr.D * wr.D;
    weights_image_block(wr.src[wi], out, wr.D, wr.tr[wi], wr.planes, blk - wi * wr.per, wr.per, wr.ld[wi]);
}

}  // namespace amid
