// K2 on the matrix cores for the BERT4Rec shape: bidirectional attention with a key mask, head dim 32 (the reference hard-codes
// hidden 128 / 4 heads, model_seq.py:264-267), T <= 64.  Reference: Attention.forward model_seq.py:149-162 -- scores / sqrt(d_k),
// masked_fill(mask == 0, -1e9), softmax, dropout(p) -- and its autograd.  Same construction as attention_mfma.h (one wave per
// head, operands loaded straight into MFMA lane layout, P~ registers reused as the B operand of P~ V, backward in two phases with
// row statistics and the 64-bit keep word of every query row travelling through LDS), with two 16-wide halves of the head dim,
// every key tile visited, keys >= T excluded, masked keys at -1e9 (a row whose keys are ALL masked gets the reference's uniform
// softmax) and no score gradient through masked keys (masked_fill).
#include "attention_mfma.h"

namespace amid {

constexpr int BHD = 32;

__device__ __forceinline__ f32x4 frag2(const float4 (&a)[2], const float4 (&b)[2], f32x4 c) {
    c = mfma_frag(a[0], b[0], c);
    return mfma_frag(a[1], b[1], c);
}

// bit (kj * 4 + r) of the result: key n = kj * 16 + 4 gq + r is inside the sequence (valid) / also visible (ok)
__device__ __forceinline__ void key_bits(const unsigned char* __restrict__ kk, int T, int gq, unsigned& valid, unsigned& ok) {
    valid = 0; ok = 0;
#pragma unroll
    for (int kj = 0; kj < 4; ++kj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = kj * 16 + 4 * gq + r;
            if (n < T) {
                valid |= 1u << (kj * 4 + r);
                if (kk == nullptr || kk[n] != 0) ok |= 1u << (kj * 4 + r);
            }
        }
}

__global__ __launch_bounds__(512) void attn_fwd_bert_kernel(const AttnArgs a) {
    const int T = a.T, D = a.D, H = a.H;
    const int seq = blockIdx.x, g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int h = wave_id(), lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NT = (T + 15) >> 4;
    const float inv = 1.0f / a.scale;
    const unsigned char* kk = a.key_keep ? a.key_keep + (long long)b * T : nullptr;
    float4 kf[4][2];
    float vt[4][4][2];
#pragma unroll
    for (int kj = 0; kj < 4; ++kj)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            kf[kj][c] = ld4_row(a.k, rowbase, kj * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
#pragma unroll
            for (int r = 0; r < 4; ++r) vt[kj][r][c] = ld1_row(a.v, rowbase, kj * 16 + 4 * gq + r, T, D, h * BHD + 16 * c + m);
        }
    unsigned valid, okb;
    key_bits(kk, T, gq, valid, okb);
    unsigned long long kw_own = ~0ull;
    if (a.train) {
        const int qrow = min(gq * 16 + m, T - 1);
        kw_own = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step,
                               (unsigned long long)(b * H + h) * T + qrow, T, a.thr16);
    }
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        if (qi >= NT) break;
        const int q = qi * 16 + m;
        float4 qf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) qf[c] = f4scale(ld4_row(a.q, rowbase, q, T, D, h * BHD + 16 * c + 4 * gq), inv);
        const unsigned long long kw = shfl64(kw_own, qi * 16 + m);
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            s[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kj < NT) {
                s[kj] = frag2(kf[kj], qf, s[kj]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned bit = 1u << (kj * 4 + r);
                    s[kj][r] = !(valid & bit) ? -INFINITY : ((okb & bit) ? s[kj][r] : -1e9f);
                    mx = fmaxf(mx, s[kj][r]);
                }
            }
        }
        mx = quad_group_max(mx);
        float l = 0.f;
        f32x4 oacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (kj < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = kj * 16 + 4 * gq + r;
                    const float p = fast_exp(s[kj][r] - mx);
                    l += p;
                    const float pd = ((kw >> n) & 1ull) ? p * a.dscale : 0.f;
                    oacc[0] = mfma4(vt[kj][r][0], pd, oacc[0]);
                    oacc[1] = mfma4(vt[kj][r][1], pd, oacc[1]);
                }
            }
        }
        l = quad_group_sum(l);
        const float rl = 1.0f / l;
        if (q < T) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                st4(a.o + (rowbase + q) * D + h * BHD + 16 * c + 4 * gq,
                    make_float4(oacc[c][0] * rl, oacc[c][1] * rl, oacc[c][2] * rl, oacc[c][3] * rl));
            if (gq == 0 && a.stats) {
                float* sp = a.stats + ((rowbase + q) * H + h) * 2;
                sp[0] = mx; sp[1] = rl;
            }
        }
    }
}

__global__ __launch_bounds__(512) void attn_bwd_bert_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, H = a.H;
    bool live = true;
    const int seq = a.row_domain != nullptr ? live_rows_remap(a.row_domain, a.B, blockIdx.x, live) : (int)blockIdx.x;
    const int g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int h = wave_id(), lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NT = (T + 15) >> 4;
    const float inv = 1.0f / a.scale;
    const unsigned char* kk = a.key_keep ? a.key_keep + (long long)b * T : nullptr;
    if (!live) {                                                                       // no gradient reaches this sequence: exact zeros
        const int hd = D / H, q4 = hd >> 2;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = lane; i < T * q4; i += 64) {
            const long long off = (rowbase + i / q4) * D + h * hd + 4 * (i % q4);
            st4(a.dq + off, z); st4(a.dk + off, z); st4(a.dv + off, z);
        }
        return;
    }
    float4* rstat = reinterpret_cast<float4*>(smem) + h * 64;                                       // [H][64] (max, 1/sum, delta, -)
    unsigned long long* keepw = reinterpret_cast<unsigned long long*>(smem + H * 64 * 4) + h * 64;   // [H][64]
    // ---------------- phase 1: lanes = queries -> dQ; row stats + keep words to LDS ----------------
    {
        float4 kf[4][2], vf[4][2];
        float kt[4][4][2];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                kf[kj][c] = ld4_row(a.k, rowbase, kj * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
                vf[kj][c] = ld4_row(a.v, rowbase, kj * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) kt[kj][r][c] = ld1_row(a.k, rowbase, kj * 16 + 4 * gq + r, T, D, h * BHD + 16 * c + m);
            }
        unsigned valid, okb;
        key_bits(kk, T, gq, valid, okb);
        unsigned long long kw_own = ~0ull;
        if (a.train) {
            const int qrow = min(gq * 16 + m, T - 1);
            kw_own = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step,
                                   (unsigned long long)(b * H + h) * T + qrow, T, a.thr16);
        }
        keepw[lane] = kw_own;                                                            // row index = 16 gq + m = lane
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            if (qi >= NT) break;
            const int q = qi * 16 + m;
            float4 qf[2], dof[2];
            float dsum = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = h * BHD + 16 * c + 4 * gq;
                qf[c] = f4scale(ld4_row(a.q, rowbase, q, T, D, col), inv);
                dof[c] = ld4_row(a.d_o, rowbase, q, T, D, col);
                dsum += f4hsum(f4mul(dof[c], ld4_row(a.o, rowbase, q, T, D, col)));
            }
            const float delta = quad_group_sum(dsum);
            const float2 st2 = *reinterpret_cast<const float2*>(a.stats + ((rowbase + min(q, T - 1)) * H + h) * 2);
            const float mrow = st2.x, rl = st2.y;
            if (gq == 0) rstat[q] = make_float4(mrow, rl, delta, 0.f);
            const unsigned long long kw = shfl64(kw_own, qi * 16 + m);
            f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                if (kj < NT) {
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
                    s = frag2(kf[kj], qf, s);
                    dp = frag2(vf[kj], dof, dp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = kj * 16 + 4 * gq + r;
                        const unsigned bit = 1u << (kj * 4 + r);
                        const bool vld = valid & bit, ok = okb & bit;
                        const float p = vld ? fast_exp((ok ? s[r] : -1e9f) - mrow) * rl : 0.f;
                        const float dpk = ((kw >> n) & 1ull) ? dp[r] * a.dscale : 0.f;
                        const float ds = ok ? p * (dpk - delta) : 0.f;               // masked_fill: no gradient through a masked score
                        dq[0] = mfma4(kt[kj][r][0], ds, dq[0]);
                        dq[1] = mfma4(kt[kj][r][1], ds, dq[1]);
                    }
                }
            }
            if (q < T) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    st4(a.dq + (rowbase + q) * D + h * BHD + 16 * c + 4 * gq,
                        make_float4(dq[c][0] * inv, dq[c][1] * inv, dq[c][2] * inv, dq[c][3] * inv));
            }
        }
    }
    // rstat / keepw of a wave are written and read by that wave only (LDS operations of one wave complete in order)
    // ---------------- phase 2: lanes = keys -> dK, dV ----------------------------------------------
    float4 qfr[4][2], dofr[4][2];
    float qts[4][4][2], dots[4][4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            qfr[t][c] = f4scale(ld4_row(a.q, rowbase, t * 16 + m, T, D, h * BHD + 16 * c + 4 * gq), inv);
            dofr[t][c] = ld4_row(a.d_o, rowbase, t * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                qts[t][r][c] = ld1_row(a.q, rowbase, t * 16 + 4 * gq + r, T, D, h * BHD + 16 * c + m) * inv;
                dots[t][r][c] = ld1_row(a.d_o, rowbase, t * 16 + 4 * gq + r, T, D, h * BHD + 16 * c + m);
            }
        }
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
        if (kj >= NT) break;
        const int key = kj * 16 + m;
        float4 kf[2], vf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            kf[c] = ld4_row(a.k, rowbase, key, T, D, h * BHD + 16 * c + 4 * gq);
            vf[c] = ld4_row(a.v, rowbase, key, T, D, h * BHD + 16 * c + 4 * gq);
        }
        const bool key_in = key < T;
        const bool key_ok = key_in && (kk == nullptr || kk[min(key, T - 1)] != 0);
        f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {dk[0], dk[0]};
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            if (qi >= NT) continue;
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = st;
            st = frag2(qfr[qi], kf, st);                 // S^T: lane (key m, gq), reg r <-> query qi*16 + 4 gq + r
            dpt = frag2(dofr[qi], vf, dpt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = qi * 16 + 4 * gq + r;
                const float4 rs = rstat[min(qq, 63)];
                const bool live = (qq < T) && key_in;
                const bool keep = (keepw[min(qq, 63)] >> key) & 1ull;
                const float p = live ? fast_exp((key_ok ? st[r] : -1e9f) - rs.x) * rs.y : 0.f;
                const float pd = keep ? p * a.dscale : 0.f;
                const float dpk = keep ? dpt[r] * a.dscale : 0.f;
                const float ds = (live && key_ok) ? p * (dpk - rs.z) : 0.f;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    dv[c] = mfma4(dots[qi][r][c], pd, dv[c]);
                    dk[c] = mfma4(qts[qi][r][c], ds, dk[c]);
                }
            }
        }
        if (key_in) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const long long off = (rowbase + key) * D + h * BHD + 16 * c + 4 * gq;
                st4(a.dk + off, make_float4(dk[c][0], dk[c][1], dk[c][2], dk[c][3]));
                st4(a.dv + off, make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]));
            }
        }
    }
}

}  // namespace amid

using namespace amid;

// called by the entry points in attention.hip when the shape fits (bidirectional, head dim 32, T <= 64)
int amid_attn_bert_fwd_launch(const void* args, void* stream) {
    const AttnArgs& a = *(const AttnArgs*)args;
    attn_fwd_bert_kernel<<<2 * a.B, a.H * 64, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

int amid_attn_bert_bwd_launch(const void* args, void* stream) {
    const AttnArgs& a = *(const AttnArgs*)args;
    const size_t lds = (size_t)a.H * 64 * (16 + 8);
    attn_bwd_bert_kernel<<<2 * a.B, a.H * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
