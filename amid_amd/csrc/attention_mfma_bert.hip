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
constexpr int BERT_KT_LD = 68;                                  // row stride of the backward's K^T image
constexpr int BERT_BWD_LDS_PER_WAVE = ATTN_BWD_LDS_PER_WAVE + BHD * BERT_KT_LD * 4;

__device__ __forceinline__ f32x4 frag2(const float4 (&a)[2], const float4 (&b)[2], f32x4 c) {
    c = mfma_frag(a[0], b[0], c);
    return mfma_frag(a[1], b[1], c);
}

// bit (kj * 4 + r) of the result: key n = kj * 16 + 4 gq + r is inside the sequence (valid) / also visible (ok)
__device__ __forceinline__ void key_bits(const unsigned char* __restrict__ kk, int T, int gq, unsigned& valid, unsigned& ok) {
    // (the sixteen mask bytes of the lane are requested back to back, clamped instead of branched around: behind a branch per key each
    // byte load waited out its own round trip -- sixteen dependent L2 latencies at the head of every wave)
    unsigned char kb[16];
#pragma unroll
    for (int kj = 0; kj < 4; ++kj)
#pragma unroll
        for (int r = 0; r < 4; ++r) kb[kj * 4 + r] = kk != nullptr ? kk[min(kj * 16 + 4 * gq + r, T - 1)] : (unsigned char)1;
    valid = 0; ok = 0;
#pragma unroll
    for (int kj = 0; kj < 4; ++kj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = kj * 16 + 4 * gq + r;
            const unsigned bit = n < T ? 1u << (kj * 4 + r) : 0u;
            valid |= bit;
            ok |= kb[kj * 4 + r] != 0 ? bit : 0u;
        }
}

// a.live (amid_live_list_i32): the launch covers the B listed sequences only -- workgroup j < live[B]: (0, live[j]), else (1, live[j])
__device__ __forceinline__ int bert_live_seq(const AttnArgs& a, int j) {
    const int n0 = a.live[a.B];
    return (j >= n0 ? a.B : 0) + a.live[j];
}

template <int WPH>                                         // waves per head: 2 (H <= 4: eight waves per sequence) or 1
__global__ __launch_bounds__(512) void attn_fwd_bert_kernel(const AttnArgs a) {
    constexpr int TPW = 4 / WPH;                           // query tiles per wave
    const int T = a.T, D = a.D, H = a.H;
    const int seq = a.live != nullptr ? bert_live_seq(a, blockIdx.x) : (int)blockIdx.x, g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    // TWO waves per head (eight per sequence): wave w serves head w % H with the query tiles {0, 1} (w < H) or {2, 3}; both load the
    // head's K / V (L2 hits the second time).  With one wave per head a CU held four waves and every query tile's chain -- scores,
    // softmax, P~ V -- ran behind the previous one's: 21 us per launch at B 256, T 50.
    const int h = wave_id() % H, qhalf = wave_id() / H, lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NT = (T + 15) >> 4;
    const float inv = 1.0f / a.scale;
    const unsigned char* kk = a.key_keep ? a.key_keep + (long long)b * T : nullptr;
    if (TPW * qhalf >= NT) return;                         // (T <= 32: the second wave of a head has no query tile)
    float4 kf[4][2];
    float vt[4][4][2];
    {   // K as row fragments; V^T (lane (m, g): key 16 kj + 4 g + r, dim 16 c + m) from V's row fragments by a 16 x 16 transpose through the
        // wave's LDS tile -- as loads the transposed fragments were 32 four-byte requests per lane, each touching four 64-byte segments
        extern __shared__ __attribute__((aligned(16))) float fsmem[];
        float* tile = fsmem + wave_id() * ATTN_BWD_TILE_FLOATS;
        float4 vf[4][2];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                kf[kj][c] = ld4_row(a.k, rowbase, kj * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
                vf[kj][c] = ld4_row(a.v, rowbase, kj * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
            }
#pragma unroll
        for (int kj = 0; kj < 4; ++kj)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float t4[4];
                tile_transpose(tile, vf[kj][c], t4);
#pragma unroll
                for (int r = 0; r < 4; ++r) vt[kj][r][c] = t4[r];
            }
    }
    unsigned valid, okb;
    key_bits(kk, T, gq, valid, okb);
    // dropout keep words (64 keys) of the wave's query rows.  WPH = 1: lane l draws row l's.  WPH = 2: the wave owns 32 query rows; lane l
    // draws the first four calls (keys 0 .. 31) of row 32 qhalf + (l & 31) if l < 32, the others (keys 32 .. 63) if not -- half the
    // Philox calls per lane --, and the row loop below puts a row's word together from its two lanes
    unsigned long long kw_own = ~0ull;
    if (a.train) {
        const unsigned site = site_id(g, a.layer, SITE_ATTN), step = (unsigned)a.st->step;
        if constexpr (WPH == 2) {
            const int qrow = min(32 * qhalf + (lane & 31), T - 1);
            kw_own = row_keep_word16(a.st->seed, site, step, (unsigned long long)(b * H + h) * T + qrow, T, spec_thr(a.thr16), lane < 32 ? 0 : 4,
                                     lane < 32 ? 4 : 8);
            if (spec_bits(a.thr16) != 16)                  // (other rates: the whole word on both lanes)
                kw_own = row_keep_word(a.st->seed, site, step, (unsigned long long)(b * H + h) * T + qrow, T, a.thr16);
        } else {
            const int qrow = min(gq * 16 + m, T - 1);
            kw_own = row_keep_word(a.st->seed, site, step, (unsigned long long)(b * H + h) * T + qrow, T, a.thr16);
        }
    }
    float4 qall[TPW][2];                 // the wave's query tiles' rows requested up front (inside the loop each tile waited out its own round trip)
#pragma unroll
    for (int qq = 0; qq < TPW; ++qq)
#pragma unroll
        for (int c = 0; c < 2; ++c) qall[qq][c] = ld4_row(a.q, rowbase, (TPW * qhalf + qq) * 16 + m, T, D, h * BHD + 16 * c + 4 * gq);
#pragma unroll
    for (int qq = 0; qq < TPW; ++qq) {
        const int qi = TPW * qhalf + qq;
        if (qi >= NT) break;
        const int q = qi * 16 + m;
        float4 qf[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) qf[c] = f4scale(qall[qq][c], inv);
        unsigned long long kw;
        if constexpr (WPH == 2) kw = shfl64(kw_own, qq * 16 + m) | shfl64(kw_own, 32 + qq * 16 + m);      // (both hold the whole word at other rates)
        else kw = shfl64(kw_own, qi * 16 + m);
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

// Backward in ONE pass over the (query tile, key tile) pairs, as attention_mfma.h's attn_bwd_compute: per pair, lanes = queries,
// S and dP~ (16 matrix instructions over the two halves of the head dim), the element-wise part, dQ += dS K; then the P~ and dS tiles
// are TRANSPOSED through the wave's LDS scratch and dV += P~^T dO, dK += dS^T Qs accumulate per key tile across the query tiles
// (the first version recomputed S^T and dP~^T with lanes = keys in a second pass: 56 matrix instructions per pair, now 40, and one
// exponential per element).  K, V and K^T stay in registers for all query tiles; Q, dO, O of a query tile are loaded when it is
// reached (two waves per SIMD cover the latency), their transposed fragments made through LDS.
template <int NT>
__device__ __forceinline__ void attn_bwd_bert_body(const AttnArgs& a, int g, int b, long long rowbase, int h, float* __restrict__ lds) {
    const int T = a.T, D = a.D, H = a.H;
    const int lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const float inv = 1.0f / a.scale;
    const unsigned char* kk = a.key_keep ? a.key_keep + (long long)b * T : nullptr;
    float* tile = lds;
    float* tile2 = lds + ATTN_BWD_TILE_FLOATS;
    const SeqBuf bq(a.q, rowbase, T, D), bk(a.k, rowbase, T, D), bv(a.v, rowbase, T, D), bo(a.o, rowbase, T, D), bdo(a.d_o, rowbase, T, D),
                 bst(a.stats, rowbase, T, 2 * H), bdq(a.dq, rowbase, T, D), bdk(a.dk, rowbase, T, D), bdv(a.dv, rowbase, T, D);
    const int colb = h * BHD + 4 * gq;
    float4 kf[4][2], vf[4][2];
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
        for (int c = 0; c < 2; ++c) { kf[kj][c] = bk.ld4(kj * 16 + m, colb + 16 * c); vf[kj][c] = bv.ld4(kj * 16 + m, colb + 16 * c); }
    unsigned valid, okb;
    key_bits(kk, T, gq, valid, okb);
    unsigned long long kw_own = ~0ull;
    if (a.train) {
        const int qrow = min(lane, T - 1);
        kw_own = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step, (unsigned long long)(b * H + h) * T + qrow, T,
                               a.thr16);
    }
    // K^T of the head as an LDS image [32 dims][68] (keys along the row; 68: the 16 dims of a read fall on 8 bank sets): the dQ product's
    // first operand -- K[key 16 kj + 4 gq + r][dim 16 c + m], r = 0 .. 3 -- is one 16-byte read instead of 32 registers held for all pairs
    float* ktimg = lds + 2 * ATTN_BWD_TILE_FLOATS;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) ktimg[(16 * c + 4 * gq + e) * BERT_KT_LD + kj * 16 + m] = f4comp(kf[kj][c], e);
    f32x4 dk[4][2], dv[4][2];
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
        for (int c = 0; c < 2; ++c) { dk[kj][c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kj][c] = dk[kj][c]; }
#pragma unroll 1                          // (a real loop: unrolled, the scheduler hoists every query tile's operand loads to the top and spills)
    for (int qi = 0; qi < NT; ++qi) {
        const int q = qi * 16 + m;
        float4 qf[2], dof[2];
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            qf[c] = f4scale(bq.ld4(q, colb + 16 * c), inv);
            dof[c] = bdo.ld4(q, colb + 16 * c);
            dsum += f4hsum(f4mul(dof[c], bo.ld4(q, colb + 16 * c)));
        }
        const float mrow = bst.ld1(q, 2 * h), rl = bst.ld1(q, 2 * h + 1);       // (a query row past T: both zero -> P = 0)
        const float delta = quad_group_sum(dsum);
        float qts[2][4], dots[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) { tile_transpose(tile, qf[c], qts[c]); tile_transpose(tile2, dof[c], dots[c]); }
        const unsigned long long kw = shfl64(kw_own, q);
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kj = 0; kj < NT; ++kj) { s[kj] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kj] = s[kj]; }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kj = 0; kj < NT; ++kj) {
                    s[kj] = mfma4(f4comp(kf[kj][c], e), f4comp(qf[c], e), s[kj]);
                    dp[kj] = mfma4(f4comp(vf[kj][c], e), f4comp(dof[c], e), dp[kj]);
                }
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kj = 0; kj < NT; ++kj) {
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = kj * 16 + 4 * gq + r;
                const unsigned bit = 1u << (kj * 4 + r);
                const bool vld = valid & bit, ok = okb & bit;
                const float p = vld ? fast_exp((ok ? s[kj][r] : -1e9f) - mrow) * rl : 0.f;
                const bool keep = (kw >> n) & 1ull;
                pd[r] = keep ? p * a.dscale : 0.f;
                const float dpk = keep ? dp[kj][r] * a.dscale : 0.f;
                ds[r] = ok ? p * (dpk - delta) : 0.f;                            // masked_fill: no gradient through a masked score
            }
            const float4 k0 = ld4(ktimg + m * BERT_KT_LD + kj * 16 + 4 * gq), k1 = ld4(ktimg + (16 + m) * BERT_KT_LD + kj * 16 + 4 * gq);
#pragma unroll
            for (int r = 0; r < 4; ++r) { dq[0] = mfma4(f4comp(k0, r), ds[r], dq[0]); dq[1] = mfma4(f4comp(k1, r), ds[r], dq[1]); }
            float pt[4], dst[4];                          // lane (key m, gq), r <-> query 16 qi + 4 gq + r
            tile_transpose(tile, make_float4(pd[0], pd[1], pd[2], pd[3]), pt);
            tile_transpose(tile2, make_float4(ds[0], ds[1], ds[2], ds[3]), dst);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    dv[kj][c] = mfma4(dots[c][r], pt[r], dv[kj][c]);
                    dk[kj][c] = mfma4(qts[c][r], dst[r], dk[kj][c]);
                }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
            bdq.st4(q, colb + 16 * c, make_float4(dq[c][0] * inv, dq[c][1] * inv, dq[c][2] * inv, dq[c][3] * inv));
    }
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bdk.st4(kj * 16 + m, colb + 16 * c, make_float4(dk[kj][c][0], dk[kj][c][1], dk[kj][c][2], dk[kj][c][3]));
            bdv.st4(kj * 16 + m, colb + 16 * c, make_float4(dv[kj][c][0], dv[kj][c][1], dv[kj][c][2], dv[kj][c][3]));
        }
}

__global__ __launch_bounds__(512) void attn_bwd_bert_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, H = a.H;
    bool live = true;
    const int seq = a.live != nullptr ? bert_live_seq(a, blockIdx.x)
                  : a.row_domain != nullptr ? live_rows_remap(a.row_domain, a.B, blockIdx.x, live) : (int)blockIdx.x;
    const int g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int h = wave_id(), lane = lane_id();
    if (!live) {                                                                       // no gradient reaches this sequence: exact zeros
        const int hd = D / H, q4 = hd >> 2;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = lane; i < T * q4; i += 64) {
            const long long off = (rowbase + i / q4) * D + h * hd + 4 * (i % q4);
            st4(a.dq + off, z); st4(a.dk + off, z); st4(a.dv + off, z);
        }
        return;
    }
    float* lds = smem + h * (BERT_BWD_LDS_PER_WAVE / 4);
    switch ((T + 15) >> 4) {
        case 1: attn_bwd_bert_body<1>(a, g, b, rowbase, h, lds); break;
        case 2: attn_bwd_bert_body<2>(a, g, b, rowbase, h, lds); break;
        case 3: attn_bwd_bert_body<3>(a, g, b, rowbase, h, lds); break;
        default: attn_bwd_bert_body<4>(a, g, b, rowbase, h, lds); break;
    }
}

}  // namespace amid

using namespace amid;

// called by the entry points in attention.hip when the shape fits (bidirectional, head dim 32, T <= 64)
int amid_attn_bert_fwd_launch(const void* args, void* stream) {
    const AttnArgs& a = *(const AttnArgs*)args;
    const size_t lds = (size_t)(a.H <= 4 ? 2 : 1) * a.H * ATTN_BWD_TILE_FLOATS * sizeof(float);      // a transpose tile per wave
    if (a.H <= 4) attn_fwd_bert_kernel<2><<<a.live != nullptr ? a.B : 2 * a.B, 2 * a.H * 64, lds, (hipStream_t)stream>>>(a);
    else attn_fwd_bert_kernel<1><<<a.live != nullptr ? a.B : 2 * a.B, a.H * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

int amid_attn_bert_bwd_launch(const void* args, void* stream) {
    const AttnArgs& a = *(const AttnArgs*)args;
    const size_t lds = (size_t)a.H * BERT_BWD_LDS_PER_WAVE;
    attn_bwd_bert_kernel<<<a.live != nullptr ? a.B : 2 * a.B, a.H * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
