// K2 on the matrix cores for LONGER sequences: the causal SASRec shape (head dim 16) at 64 < T <= 256 -- isInC doubles the
// sequence (2 x 50 = 100 tokens, model_seq.py:398-401) and the reference's amazon setting is seq_len 150 (train_sr_dr.py:550).
// Same lane layouts and MFMA chains as attention_mfma.hip (read its header first); what changes is that a wave no longer holds a
// whole (sequence, head) in registers: queries and keys are walked in BLOCKS of 64 (4 MFMA tiles).
//   forward : for each query block, key blocks 0 .. diagonal with an online softmax (running row max m and sum l per query; the
//             accumulated P~ V and l are rescaled by exp(m_old - m_new) when a block raises the max); stats = final (m, 1 / l).
//   backward: phase 1 (lanes = queries) walks the key blocks of a query block with the SAVED statistics (no rescaling needed)
//             and accumulates dQ; phase 2 (lanes = keys) walks the query blocks at and below the diagonal for dK, dV.
//             Row statistics (max, 1/sum, delta) and one 64-bit dropout keep word per (query row, key block) travel through LDS.
// Causality skips the blocks above the diagonal and, inside the diagonal block, the tiles above it.
// Dropout: the same counters as everywhere (rng.h): at p = 0.5 one Philox call decides 128 keys of a query row = two key blocks.
#include "attention_mfma.h"

namespace amid {


// keep bits (1 = keep) of keys 64 kb .. 64 kb + 63 of attention row `row`
__device__ __forceinline__ unsigned long long row_keep_word_blk(unsigned long long seed, unsigned site, unsigned step, unsigned long long row,
                                                                int T, unsigned spec, int kb) {
    const int b = spec_bits(spec), per = 128 / b;
    const unsigned thr = spec_thr(spec);
    const int calls = (T + per - 1) / per;
    if (b == 1) {
        const uint4 r = rng_call(seed, row * calls + (kb >> 1), site, step);
        const unsigned long long w = (kb & 1) ? (((unsigned long long)r.w << 32) | r.z) : (((unsigned long long)r.y << 32) | r.x);
        return thr ? w : ~0ull;
    }
    unsigned long long w = 0;
    for (int f0 = 0; f0 < 64; f0 += per) {                     // per <= 64 here (b >= 2)
        const int key0 = kb * 64 + f0;
        const uint4 r = rng_call(seed, row * calls + key0 / per, site, step);
        for (int f = 0; f < per && f0 + f < 64; ++f)
            if (rng_field(r, f, b) >= thr) w |= 1ull << (f0 + f);
    }
    return w;
}

__global__ __launch_bounds__(256) void attn_fwd_long_kernel(const AttnArgs a) {
    const int T = a.T, D = a.D, H = a.H;
    const int hw = blockDim.x >> 6, parts = H / hw;
    const int seq = blockIdx.x / parts, part = blockIdx.x - seq * parts, g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int h = part * hw + wave_id(), lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NB = (T + 63) >> 6;
    const int col4 = h * AHD + 4 * gq, colm = h * AHD + m;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    const unsigned site = site_id(g, a.layer, SITE_ATTN);
    for (int qb = 0; qb < NB; ++qb) {
        float4 qfr[4];
        float mx[4], l[4];
        f32x4 oacc[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            qfr[qi] = f4scale(ld4_row(a.q, rowbase, qb * 64 + qi * 16 + m, T, D, col4), a.scale);
            mx[qi] = -INFINITY; l[qi] = 0.f; oacc[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int qrow_own = min(qb * 64 + gq * 16 + m, T - 1);           // lane (m, gq) generates the keep words of this query row
        for (int kb = 0; kb <= qb; ++kb) {
            const bool diag = kb == qb;
            float4 kf[4];
            float vt[4][4];
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                kf[kj] = ld4_row(a.k, rowbase, kb * 64 + kj * 16 + m, T, D, col4);
#pragma unroll
                for (int r = 0; r < 4; ++r) vt[kj][r] = ld1_row(a.v, rowbase, kb * 64 + kj * 16 + 4 * gq + r, T, D, colm);
            }
            unsigned long long kw_own = ~0ull;
            if (a.train) kw_own = row_keep_word_blk(seed, site, step, (unsigned long long)(b * H + h) * T + qrow_own, T, a.thr16, kb);
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int q = qb * 64 + qi * 16 + m;
                const unsigned long long kw = shfl64(kw_own, qi * 16 + m);
                f32x4 s[4];
                float bm = -INFINITY;
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) {
                    s[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (!diag || kj <= qi) {
                        s[kj] = mfma_frag(kf[kj], qfr[qi], s[kj]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int n = kb * 64 + kj * 16 + 4 * gq + r;
                            s[kj][r] = (n > q) ? -INFINITY : s[kj][r];
                            bm = fmaxf(bm, s[kj][r]);
                        }
                    }
                }
                bm = quad_group_max(bm);
                const float mnew = fmaxf(mx[qi], bm);              // finite: key 64 kb <= q is always visible
                const float alpha = fast_exp(mx[qi] - mnew);       // exp(-inf) = 0 on the first block
                float lb = 0.f;
                f32x4 ob = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) {
                    if (!diag || kj <= qi) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int nl = kj * 16 + 4 * gq + r;
                            const float p = fast_exp(s[kj][r] - mnew);
                            lb += p;
                            const float pd = ((kw >> nl) & 1ull) ? p * a.dscale : 0.f;
                            ob = mfma4(vt[kj][r], pd, ob);
                        }
                    }
                }
                lb = quad_group_sum(lb);
                l[qi] = l[qi] * alpha + lb;
                oacc[qi] = f32x4{oacc[qi][0] * alpha + ob[0], oacc[qi][1] * alpha + ob[1], oacc[qi][2] * alpha + ob[2], oacc[qi][3] * alpha + ob[3]};
                mx[qi] = mnew;
            }
        }
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int q = qb * 64 + qi * 16 + m;
            if (q < T) {
                const float rl = 1.0f / l[qi];
                st4(a.o + (rowbase + q) * D + col4, make_float4(oacc[qi][0] * rl, oacc[qi][1] * rl, oacc[qi][2] * rl, oacc[qi][3] * rl));
                if (gq == 0 && a.stats) {
                    float* sp = a.stats + ((rowbase + q) * H + h) * 2;
                    sp[0] = mx[qi]; sp[1] = rl;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_long_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, H = a.H;
    const int hw = blockDim.x >> 6, parts = H / hw;
    const int seq = blockIdx.x / parts, part = blockIdx.x - seq * parts, g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int wv = wave_id(), h = part * hw + wv, lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NB = (T + 63) >> 6, RT = NB * 64;
    const int col4 = h * AHD + 4 * gq, colm = h * AHD + m;
    float4* rstat = reinterpret_cast<float4*>(smem) + wv * RT;                                               // [hw][RT] (max, 1/sum, delta, -)
    unsigned long long* keepw = reinterpret_cast<unsigned long long*>(smem + hw * RT * 4) + wv * RT * NB;    // [hw][RT][NB]
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    const unsigned site = site_id(g, a.layer, SITE_ATTN);

    // ---------------- phase 1: lanes = queries -> dQ; row stats + keep words to LDS ----------------
    for (int qb = 0; qb < NB; ++qb) {
        float4 qfr[4], dofr[4];
        float mrow[4], rl[4], delta[4];
        f32x4 dq[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int row = qb * 64 + qi * 16 + m;
            qfr[qi] = f4scale(ld4_row(a.q, rowbase, row, T, D, col4), a.scale);
            dofr[qi] = ld4_row(a.d_o, rowbase, row, T, D, col4);
            const float4 of = ld4_row(a.o, rowbase, row, T, D, col4);
            const float2 st = *reinterpret_cast<const float2*>(a.stats + ((rowbase + min(row, T - 1)) * H + h) * 2);
            mrow[qi] = st.x; rl[qi] = st.y;
            delta[qi] = quad_group_sum(f4hsum(f4mul(dofr[qi], of)));
            if (gq == 0) rstat[row] = make_float4(st.x, st.y, delta[qi], 0.f);
            dq[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const int qrow_own = qb * 64 + lane;                       // row index = 16 gq + m = lane inside the block
        for (int kb = 0; kb <= qb; ++kb) {
            const bool diag = kb == qb;
            float4 kfr[4], vfr[4];
            float kt[4][4];
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                kfr[kj] = ld4_row(a.k, rowbase, kb * 64 + kj * 16 + m, T, D, col4);
                vfr[kj] = ld4_row(a.v, rowbase, kb * 64 + kj * 16 + m, T, D, col4);
#pragma unroll
                for (int r = 0; r < 4; ++r) kt[kj][r] = ld1_row(a.k, rowbase, kb * 64 + kj * 16 + 4 * gq + r, T, D, colm);
            }
            unsigned long long kw_own = ~0ull;
            if (a.train) kw_own = row_keep_word_blk(seed, site, step, (unsigned long long)(b * H + h) * T + min(qrow_own, T - 1), T, a.thr16, kb);
            keepw[qrow_own * NB + kb] = kw_own;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int q = qb * 64 + qi * 16 + m;
                const unsigned long long kw = shfl64(kw_own, qi * 16 + m);
#pragma unroll
                for (int kj = 0; kj < 4; ++kj) {
                    if (!diag || kj <= qi) {
                        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
                        s = mfma_frag(kfr[kj], qfr[qi], s);
                        dp = mfma_frag(vfr[kj], dofr[qi], dp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int nl = kj * 16 + 4 * gq + r, n = kb * 64 + nl;
                            const float p = (n > q) ? 0.f : fast_exp(s[r] - mrow[qi]) * rl[qi];
                            const float dpk = ((kw >> nl) & 1ull) ? dp[r] * a.dscale : 0.f;
                            const float ds = p * (dpk - delta[qi]);
                            dq[qi] = mfma4(kt[kj][r], ds, dq[qi]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int q = qb * 64 + qi * 16 + m;
            if (q < T)
                st4(a.dq + (rowbase + q) * D + col4, make_float4(dq[qi][0] * a.scale, dq[qi][1] * a.scale, dq[qi][2] * a.scale, dq[qi][3] * a.scale));
        }
    }
    // rstat / keepw of a wave are written and read by that wave only (LDS operations of one wave complete in order): no barrier
    // ---------------- phase 2: lanes = keys -> dK, dV ----------------
    for (int kb = 0; kb < NB; ++kb) {
        float4 kfr[4], vfr[4];
        f32x4 dk[4], dv[4];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            kfr[kj] = ld4_row(a.k, rowbase, kb * 64 + kj * 16 + m, T, D, col4);
            vfr[kj] = ld4_row(a.v, rowbase, kb * 64 + kj * 16 + m, T, D, col4);
            dk[kj] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kj] = dk[kj];
        }
        for (int qb = kb; qb < NB; ++qb) {
            const bool diag = kb == qb;
            float4 qfr[4], dofr[4];
            float qts[4][4], dots[4][4];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                qfr[qi] = f4scale(ld4_row(a.q, rowbase, qb * 64 + qi * 16 + m, T, D, col4), a.scale);
                dofr[qi] = ld4_row(a.d_o, rowbase, qb * 64 + qi * 16 + m, T, D, col4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qts[qi][r] = ld1_row(a.q, rowbase, qb * 64 + qi * 16 + 4 * gq + r, T, D, colm) * a.scale;
                    dots[qi][r] = ld1_row(a.d_o, rowbase, qb * 64 + qi * 16 + 4 * gq + r, T, D, colm);
                }
            }
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                const int key = kb * 64 + kj * 16 + m, keyl = kj * 16 + m;
#pragma unroll
                for (int qi = 0; qi < 4; ++qi) {
                    if (diag && qi < kj) continue;
                    f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = st;
                    st = mfma_frag(qfr[qi], kfr[kj], st);        // S^T: lane (key m, gq), reg r <-> query qb*64 + qi*16 + 4 gq + r
                    dpt = mfma_frag(dofr[qi], vfr[kj], dpt);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qq = qb * 64 + qi * 16 + 4 * gq + r;
                        const float4 rs = rstat[qq];
                        const bool live = (qq < T) && (key <= qq);
                        const bool keep = (keepw[qq * NB + kb] >> keyl) & 1ull;
                        const float p = live ? fast_exp(st[r] - rs.x) * rs.y : 0.f;
                        const float pd = keep ? p * a.dscale : 0.f;
                        const float dpk = keep ? dpt[r] * a.dscale : 0.f;
                        const float ds = p * (dpk - rs.z);
                        dv[kj] = mfma4(dots[qi][r], pd, dv[kj]);
                        dk[kj] = mfma4(qts[qi][r], ds, dk[kj]);
                    }
                }
            }
        }
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            const int key = kb * 64 + kj * 16 + m;
            if (key < T) {
                st4(a.dk + (rowbase + key) * D + col4, make_float4(dk[kj][0], dk[kj][1], dk[kj][2], dk[kj][3]));
                st4(a.dv + (rowbase + key) * D + col4, make_float4(dv[kj][0], dv[kj][1], dv[kj][2], dv[kj][3]));
            }
        }
    }
}

}  // namespace amid

using namespace amid;

// called by the entry points in attention.hip when the shape fits (causal, head dim 16, 64 < T <= 256, no key mask, H % 4 == 0)
int amid_attn_long_fwd_launch(const void* args, void* stream) {
    const AttnArgs a = *(const AttnArgs*)args;
    const int hw = 4, grid = 2 * a.B * (a.H / hw);
    attn_fwd_long_kernel<<<grid, hw * 64, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

int amid_attn_long_bwd_launch(const void* args, void* stream) {
    const AttnArgs a = *(const AttnArgs*)args;
    const int hw = 4, grid = 2 * a.B * (a.H / hw);
    const int NB = (a.T + 63) / 64, RT = NB * 64;
    const size_t lds = (size_t)hw * RT * (16 + 8 * NB);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_long_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    attn_bwd_long_kernel<<<grid, hw * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
