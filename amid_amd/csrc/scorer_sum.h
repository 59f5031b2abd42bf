// The scorer's weight gradients summed over the batch from per-sample hidden gradients, as a device function of one 256-thread workgroup:
// rides as extra workgroups of a launch with idle CUs (sasrec_strip.hip: the middle backward strip launch of the live-sequence step) or of
// the gradient tail (segreduce.hip).
#pragma once
#include "common.h"

#ifndef SCORER_FENCE
#define SCORER_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
namespace amid {

// The scorer's weight gradients from the per-sample hidden gradients the head launch left (head_fused.hip
// HeadArgs::hidg: da [2][hid] | dc [NI][hid] | dW2's [hid] | db2's per sample) -- dW1[j][0 .. D) = sum_b da[b][0][j] u[0][b] + da[b][1][j] u[1][b],
// dW1[j][D .. 2D) = sum_b sum_n dc[b][n][j] items[b][n], db1, dW2, db2 = sums over b: workgroup (j, half) = 32 float4 columns x 8 groups of
// samples (group p takes b = p, p + 8, ... in order), the eight group sums added in order; a workgroup per 32 of the 2 hid + 1 scalars.
struct ScorerSum { const float* hidg; const float* u; const float* items; int B, NI, D, hid, HG; float* dW1; float* db1; float* dW2; float* db2; int nblk; };

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
        if (on && half == 0) {
            // (sample, domain) pairs of the group flattened -- pair q <-> sample pg + 8 (q / 2), domain q % 2 -- eight at a time, as the item half below
            // (sixteen samples' loads in one batch: the compiler sinks every load to its use again, one round trip per sample)
            const int per = (B - pg + 7) / 8;
            const int npair = per * 2;
            for (int q0 = 0; q0 < npair; q0 += 8) {
                float c[8];
                float4 r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = min(q0 + k, npair - 1);
                    const int d = q & 1;
                    const long long b = pg + 8 * (q >> 1);
                    c[k] = ss.hidg[b * HG + d * hid + j];
                    r[k] = ld4(ss.u + ((long long)d * B + b) * D + 4 * el);
                }
                SCORER_FENCE();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (q0 + k < npair) { s.x = fmaf(c[k], r[k].x, s.x); s.y = fmaf(c[k], r[k].y, s.y); s.z = fmaf(c[k], r[k].z, s.z); s.w = fmaf(c[k], r[k].w, s.w); }
                }
            }
        } else if (on) {
            // (sample, item) pairs of the group flattened -- pair q <-> sample pg + 8 (q / NI), item q % NI -- and taken eight at a time: every
            // load of a batch is issued before its first use (one pair per round trip took 32 dependent round trips at NI = 2)
            const int per = (B - pg + 7) / 8;               // samples of this group
            const int npair = per * NI;
            for (int q0 = 0; q0 < npair; q0 += 8) {
                float c[8];
                float4 r[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int q = min(q0 + k, npair - 1);
                    const int bi = q / NI, n = q - bi * NI;
                    const long long b = pg + 8 * bi;
                    c[k] = ss.hidg[b * HG + (2 + n) * hid + j];
                    r[k] = ld4(ss.items + (b * NI + n) * D + 4 * el);
                }
                SCORER_FENCE();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (q0 + k < npair) { s.x = fmaf(c[k], r[k].x, s.x); s.y = fmaf(c[k], r[k].y, s.y); s.z = fmaf(c[k], r[k].z, s.z); s.w = fmaf(c[k], r[k].w, s.w); }
                }
            }
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
    }
    return ss;
}

}  // namespace amid
