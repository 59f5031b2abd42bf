// The scorer's weight gradients summed over the batch from per-sample hidden gradients, as a device function of one 256-thread workgroup:
// rides as extra workgroups of a launch with idle CUs (sasrec_strip.hip: the middle backward strip launch of the live-sequence step) or of
// the gradient tail (segreduce.hip).
#pragma once
#include "common.h"
#include <type_traits>

#ifndef SCORER_FENCE
#define SCORER_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
namespace amid {

// The scorer's weight gradients from the per-sample hidden gradients the head launch left (head_fused.hip
// HeadArgs::hidg: da [2][hid] | dc [NI][hid] | dW2's [hid] | db2's per sample) -- dW1[j][0 .. D) = sum_b da[b][0][j] u[0][b] + da[b][1][j] u[1][b],
// dW1[j][D .. 2D) = sum_b sum_n dc[b][n][j] items[b][n], db1, dW2, db2 = sums over b: workgroup (j, half) = 32 float4 columns x 8 groups of
// samples (group p takes b = p, p + 8, ... in order), the eight group sums added in order; a workgroup per 32 of the 2 hid + 1 scalars.
struct ScorerSum { const float* hidg; const float* u; const float* items; int B, NI, D, hid, HG; float* dW1; float* db1; float* dW2; float* db2; int nblk; int front; };
// front: the rider workgroups stand IN FRONT of the host launch's own (long batches: a workgroup walks B / 8 samples per group -- behind the tiles
// of a many-round launch it would start when the last tile does and add its whole length to the launch: + 130 us at B 4096, round 6)

// red: 8 x 33 float4 of LDS (the caller's: static in the gradient tail, a corner of the dynamic allocation in the strip launch -- hipcc 7.2
// dies in instruction selection on a third static array beside a dynamic one there: "Illegal instruction detected ... $src_shared_base")
// (an explicit LDS pointer: through a generic one the address-space cast's null check is what hipcc 7.2 cannot select there)
typedef __attribute__((address_space(3))) f32x4 scorer_lds_f4;
typedef __attribute__((address_space(3))) float scorer_lds_f;
__device__ __forceinline__ void scorer_sum_block(const ScorerSum& ss, int blk, scorer_lds_f4* red) {
    scorer_lds_f4 (*sred)[33] = (scorer_lds_f4 (*)[33])red;
    const int el = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int D = ss.D, hid = ss.hid, B = ss.B, NI = ss.NI, HG = ss.HG;
    if (blk < 2 * hid) {
        const int j = blk >> 1, half = blk & 1;
        const bool on = el < (D >> 2);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        // NB pairs in flight per round trip: 8 at the headline batch, 32 for long batches (the sums add in the same order either way)
        auto walk = [&](auto NBC, auto&& coef, auto&& row, int npair) {
            constexpr int NB = decltype(NBC)::value;
            for (int q0 = 0; q0 < npair; q0 += NB) {
                float c[NB];
                float4 r[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int q = min(q0 + k, npair - 1);
                    c[k] = coef(q);
                    r[k] = row(q);
                }
                SCORER_FENCE();
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    if (q0 + k < npair) { s.x = fmaf(c[k], r[k].x, s.x); s.y = fmaf(c[k], r[k].y, s.y); s.z = fmaf(c[k], r[k].z, s.z); s.w = fmaf(c[k], r[k].w, s.w); }
                }
            }
        };
        const int per = (B - pg + 7) / 8;               // samples of this group
        if (on && half == 0) {
            // (sample, domain) pairs of the group flattened -- pair q <-> sample pg + 8 (q / 2), domain q % 2 -- NB at a time, as the item half below
            // (the loads of a batch are issued before its first use: the compiler sinks every load to its use again, one round trip per sample)
            const int npair = per * 2;
            auto coef = [&](int q) { return ss.hidg[(long long)(pg + 8 * (q >> 1)) * HG + (q & 1) * hid + j]; };
            auto row = [&](int q) { return ld4(ss.u + ((long long)(q & 1) * B + (pg + 8 * (q >> 1))) * D + 4 * el); };
            if (npair >= 256) walk(std::integral_constant<int, 32>{}, coef, row, npair); else walk(std::integral_constant<int, 8>{}, coef, row, npair);
        } else if (on) {
            // (sample, item) pairs of the group flattened -- pair q <-> sample pg + 8 (q / NI), item q % NI
            const int npair = per * NI;
            auto coef = [&](int q) { const int bi = q / NI, n = q - bi * NI; return ss.hidg[(long long)(pg + 8 * bi) * HG + (2 + n) * hid + j]; };
            auto row = [&](int q) { const int bi = q / NI, n = q - bi * NI; return ld4(ss.items + ((long long)(pg + 8 * bi) * NI + n) * D + 4 * el); };
            if (npair >= 256) walk(std::integral_constant<int, 32>{}, coef, row, npair); else walk(std::integral_constant<int, 8>{}, coef, row, npair);
        }
        sred[pg][el] = f32x4{s.x, s.y, s.z, s.w};
        __syncthreads();
        if (pg == 0 && on) {
            f32x4 tv = sred[0][el];
#pragma unroll
            for (int q = 1; q < 8; ++q) tv += sred[q][el];
            const float4 t = make_float4(tv[0], tv[1], tv[2], tv[3]);
            st4(ss.dW1 + (long long)j * 2 * D + half * D + 4 * el, t);
        }
        return;
    }
    // db1[j] = sum_b da[b][0][j] + da[b][1][j] ; dW2[j], db2 = sums of the samples' own: a workgroup per 32 of the 2 hid + 1 outputs
    scorer_lds_f* sr = (scorer_lds_f*)red;
    {
        const int o = (blk - 2 * hid) * 32 + el;
        const bool ok = o < 2 * hid + 1;
        const int oa = ok ? (o < hid ? o : (2 + NI) * hid + (o - hid)) : 0;        // first addend's offset in a sample's vector
        float acc = 0.f;
        for (int b0 = pg; b0 < B; b0 += 64) {               // eight samples in flight
            float x[8], y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const long long b = min(b0 + 8 * k, B - 1);
                x[k] = ss.hidg[b * HG + oa];
                y[k] = (ok && o < hid) ? ss.hidg[b * HG + hid + o] : 0.f;
            }
            SCORER_FENCE();
#pragma unroll
            for (int k = 0; k < 8; ++k) if (b0 + 8 * k < B) acc += x[k] + y[k];
        }
        sr[pg * 33 + el] = acc;
        __syncthreads();
        if (pg == 0 && ok) {
            float t = sr[el];
#pragma unroll
            for (int q = 1; q < 8; ++q) t += sr[q * 33 + el];
            if (o < hid) ss.db1[o] = t; else if (o < 2 * hid) ss.dW2[o - hid] = t; else ss.db2[0] = t;
        }
    }
}

static inline ScorerSum scorer_sum_args(const float* hidg, const float* u, const float* items, int B, int NI, int D, int hid, float* dW1,
                                        float* db1, float* dW2, float* db2) {
    ScorerSum ss = {};
    if (hidg != nullptr) {
        ss.hidg = hidg; ss.u = u; ss.items = items; ss.B = B; ss.NI = NI; ss.D = D; ss.hid = hid; ss.HG = ((3 + NI) * hid + 1 + 3) & ~3;
        ss.dW1 = dW1; ss.db1 = db1; ss.dW2 = dW2; ss.db2 = db2; ss.nblk = 2 * hid + (2 * hid + 1 + 31) / 32;
        ss.front = B >= 1024 ? 1 : 0;
    }
    return ss;
}

}  // namespace amid
