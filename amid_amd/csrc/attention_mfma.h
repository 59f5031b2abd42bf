// K2 on the matrix cores (shared device code: attention_mfma.hip holds the kernels): causal multi-head attention for the SASRec shape (head dim 16, T <= 64), forward and
// backward.  Reference: softmax((q sqrt(1/hd)) k^T + causal(-inf)) -> dropout(p) -> . v inside nn.MultiheadAttention as
// called at model_seq.py:374, and its autograd.  (attention.hip keeps the general VALU kernels: any T, hd 8/16/32, key masks.)
//
// One workgroup per (domain, batch row), one wave per head.  Every product is a chain of v_mfma_f32_16x16x4_f32 on
// 16x16 tiles; operands are loaded straight from global/L2 into their MFMA lane layout (no LDS staging):
//   S   = Qs K^T   A-operand = K rows (lane i = key, k = 4 g + j), B-operand = Qs rows  -> lane (m, g) holds
//                  S[query m][key 4 g + r], r = 0..3, of the tile: a query's keys sit in 4 registers x 4 lane groups x
//                  (T/16) tiles, so the softmax max / sum is 15 in-lane ops + two wavefront shuffles (xor 16, xor 32).
//   O   = P~ V     contraction over keys: the P~ registers of step (tile, r) ARE the B-operand (lane group g supplies key
//                  4 g + r) -- no data movement; the A-operand is V^T, fetched as dwords V[key 4 g + r][d = lane & 15].
//   backward, phase 1 (lanes = queries): dP~ = dO V^T the same way, delta = dO . O (two shuffles), dS = P (dP - delta),
//                  dQ = dS K with dS as B-operand and K^T dwords as A-operand.
//   backward, phase 2 (lanes = keys): S^T = K Qs^T and dP~^T = V dO^T recomputed in the transposed layout, so that
//                  dV = P~^T dO and dK = dS^T Qs again take their own registers as B-operand.  Row statistics
//                  (max, 1/sum, delta) and the 64-bit dropout keep word of every query row travel through LDS.
// Causality skips the tiles above the diagonal (10 of 16 tiles at T = 50..64).
// Dropout: p = 0.5 needs ONE Philox call per query row (1 bit per key, rng.h); lane (m, g) computes the row 16 g + m,
// i.e. a single call sequence per wave covers all 64 rows; a row's word reaches its query tile with two shuffles.
#pragma once
#include "common.h"
#include "rng.h"

#ifndef STRIP_RSTAMP
#define STRIP_RSTAMP(i) do { } while (0)
#endif

namespace amid {

struct AttnArgs {          // must stay identical to the struct in attention.hip
    const float* q; const float* k; const float* v;
    float* o;
    float* stats;
    const float* d_o; float* dq; float* dk; float* dv;
    const unsigned char* key_keep;
    int B, T, D, H;
    int causal;
    float scale;
    const StepState* st; int train; unsigned thr16; float dscale; int layer;
    int stagger_from, stagger_sleeps;      // set by the MFMA backward launcher only
    const long long* row_domain;           // backward only, optional [B]: sequence (g, b) has a gradient only if (row_domain[b] != 0) == g
    const int* live;                       // matrix-core kernels only, optional [B + 1] (amid_live_list_i32): the launch covers the B
                                           // listed sequences only -- slot j < live[B]: (0, live[j]), else (1, live[j]); nothing else is touched
};


// ---- backward with a row_domain hint (AttnArgs::row_domain): which sequence does workgroup slot j work on? ----
// Sequence (g, b) carries a gradient only if (row_domain[b] != 0) == g.  Handing slot j the sequence j would leave the CUs
// unevenly loaded (the hardware spreads consecutive workgroups over the CUs in a fixed pattern: with a random half of them exiting at
// once some CUs keep twice the work of others -- measured at B 256: 36.5 us against 26.8 us for an all-active launch of half the size), so
// the slots are renumbered: [0, B) = the B live sequences (domain 0's, then domain 1's, each in batch order), [B, 2B) = the dead
// ones, which only store zeros.  Every workgroup derives the mapping itself from the B domain flags (ballots + popcounts, wave-uniform,
// under a microsecond for B <= 1024); larger batches keep the identity mapping.
__device__ __forceinline__ int nth_row_with(const long long* __restrict__ dom, int B, int val, int k) {
    const int lane = lane_id();
    for (int c = 0; c < B; c += 64) {
        const int b = c + lane;
        const bool f = b < B && ((dom[b] != 0 ? 1 : 0) == val);
        unsigned long long m = __ballot(f);
        const int n = __popcll(m);
        if (k < n) {
            for (int i = 0; i < k; ++i) m &= m - 1;          // drop the k lowest set bits
            return c + __ffsll((long long)m) - 1;
        }
        k -= n;
    }
    return 0;
}
__device__ __forceinline__ int live_rows_remap(const long long* __restrict__ dom, int B, int j, bool& live) {
    if (B > 1024) {                                           // identity: the sequence's own flag decides
        const int g = j / B, b = j - g * B;
        live = (dom[b] != 0 ? 1 : 0) == g;
        return j;
    }
    const int lane = lane_id();
    int n0 = 0;                                               // rows whose own domain is 0
    for (int c = 0; c < B; c += 64) n0 += __popcll(__ballot(c + lane < B && dom[c + lane] == 0));
    live = j < B;
    int g, val, k;
    if (j < n0)          { g = 0; val = 0; k = j; }                     // live sequences of domain 0
    else if (j < B)      { g = 1; val = 1; k = j - n0; }                // live sequences of domain 1
    else if (j - B < B - n0) { g = 0; val = 1; k = j - B; }             // dead ones: domain-0 encoder rows of domain-1 samples
    else                 { g = 1; val = 0; k = j - B - (B - n0); }      //            domain-1 encoder rows of domain-0 samples
    return g * B + nth_row_with(dom, B, val, k);
}

constexpr int AHD = 16;
constexpr float LOG2E = 1.4426950408889634f;
// exp(x) as one v_exp_f32 (2^y, ~1 ulp): the softmax here evaluates ~2 500 exponentials per head, and the library expf
// (range reduction + polynomial, ~12 instructions) was the largest VALU item of the kernel.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_frag(float4 a, float4 b, f32x4 c) {
    c = mfma4(a.x, b.x, c); c = mfma4(a.y, b.y, c); c = mfma4(a.z, b.z, c); c = mfma4(a.w, b.w, c);
    return c;
}
// rows past T read as zeros WITHOUT a branch: the address is clamped to the last row and the value selected afterwards -- a load inside
// a conditional block is waited for at the block's end when anything (a scale factor) is applied to it there, i.e. a kernel's operand
// loads complete one memory round trip after the other instead of all being in flight at once
__device__ __forceinline__ float4 ld4_row(const float* __restrict__ base, long long rowbase, int row, int T, int D, int col) {
    const float4 v = ld4(base + (rowbase + min(row, T - 1)) * D + col);
    return (row < T) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float ld1_row(const float* __restrict__ base, long long rowbase, int row, int T, int D, int col) {
    const float v = base[(rowbase + min(row, T - 1)) * D + col];
    return (row < T) ? v : 0.f;
}
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
    const unsigned lo = __shfl((unsigned)v, src, 64), hi = __shfl((unsigned)(v >> 32), src, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ float quad_group_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float quad_group_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// 16-bit decisions (p = 0.1: the BERT4Rec rate): a call decides eight keys.  The calls c0 <= c < c1 of the row (c < ceil(T / 8)), unrolled,
// eight compares per call, the word built in two 32-bit halves -- the generic loop below (runtime field width, 64-bit shifts) took 6 us of
// a 21 us attention launch.  A caller may split a row's calls over two lanes (c0 / c1) and OR the parts.
__device__ __forceinline__ unsigned long long row_keep_word16(unsigned long long seed, unsigned site, unsigned step, unsigned long long row, int T,
                                                              unsigned thr, int c0, int c1) {
    const int calls = (T + 7) >> 3;
    unsigned lo = 0u, hi = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (c >= c0 && c < c1 && c < calls) {              // (wave-uniform)
            const uint4 r = rng_call(seed, row * calls + c, site, step);
            const unsigned b8 = ((r.x & 0xFFFFu) >= thr ? 1u : 0u) | ((r.x >> 16) >= thr ? 2u : 0u) | ((r.y & 0xFFFFu) >= thr ? 4u : 0u) |
                                ((r.y >> 16) >= thr ? 8u : 0u) | ((r.z & 0xFFFFu) >= thr ? 16u : 0u) | ((r.z >> 16) >= thr ? 32u : 0u) |
                                ((r.w & 0xFFFFu) >= thr ? 64u : 0u) | ((r.w >> 16) >= thr ? 128u : 0u);
            if (c < 4) lo |= b8 << (8 * c); else hi |= b8 << (8 * (c - 4));
        }
    }
    return ((unsigned long long)hi << 32) | lo;
}

// keep bits (1 = keep) of keys 0..63 of attention row `row` (row-major [b, h, i]); same indexing as attention.hip / the oracle
__device__ __forceinline__ unsigned long long row_keep_word(unsigned long long seed, unsigned site, unsigned step, unsigned long long row, int T,
                                                            unsigned spec) {
    const int b = spec_bits(spec), per = 128 / b;
    const unsigned thr = spec_thr(spec);
    const int calls = (T + per - 1) / per;
    if (b == 1) {                                        // the SASRec case: one call, keep <=> bit >= thr (thr = 1)
        const uint4 r = rng_call(seed, row * calls, site, step);
        const unsigned long long w = ((unsigned long long)r.y << 32) | r.x;
        return thr ? w : ~0ull;
    }
    if (b == 16) return row_keep_word16(seed, site, step, row, T, thr, 0, 8);      // the BERT4Rec case (p = 0.1)
    unsigned long long w = 0;
    for (int c = 0; c * per < 64 && c < calls; ++c) {
        const uint4 r = rng_call(seed, row * calls + c, site, step);
        for (int f = 0; f < per; ++f)
            if (rng_field(r, f, b) >= thr) w |= 1ull << (c * per + f);
    }
    return w;
}

// forward of ONE head h of ONE sequence (domain g, batch row b, rows rowbase .. rowbase + T) by the calling wave; used by the
// standalone kernel below and by the fused per-layer forward kernel (sasrec_fwd.hip).
// PAIR (head dim 8: the reference's default --emb_dim 64 with its 8 heads, train_sr.py:364): h is a 16-column tile = the heads 2 h and
// 2 h + 1.  The tile is loaded as one head of 16 dims would be; lane groups 0, 1 hold head 2 h's dims, groups 2, 3 head 2 h + 1's.  Per
// head the K operand of S = Qs K^T is zeroed in the other head's lane groups (the matrix instruction contracts over the lane groups), and
// P~ V is computed for all 16 dims with that head's P~ -- a lane keeps the result of its own head.  Twice the matrix instructions of a
// 16-dim head per tile, half of each used: the core is latency-bound either way.
template <bool PAIR = false>
__device__ __forceinline__ void attn_fwd_head(const AttnArgs& a, int g, int b, long long rowbase, int h) {
    const int T = a.T, D = a.D, H = a.H;
    const int lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NT = (T + 15) >> 4;
    constexpr int NE = PAIR ? 2 : 1;
    const int col4 = h * AHD + 4 * gq, colm = h * AHD + m;
    // operands that every query tile reuses
    float4 kf[4];
    float vt[4][4];
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
        kf[kj] = ld4_row(a.k, rowbase, kj * 16 + m, T, D, col4);
#pragma unroll
        for (int r = 0; r < 4; ++r) vt[kj][r] = ld1_row(a.v, rowbase, kj * 16 + 4 * gq + r, T, D, colm);
    }
    unsigned long long kw_own[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        kw_own[e] = ~0ull;
        if (a.train) {
            const int qrow = min(gq * 16 + m, T - 1);
            kw_own[e] = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step,
                                      (unsigned long long)(b * H + (PAIR ? 2 * h + e : h)) * T + qrow, T, a.thr16);
        }
    }
    float4 qfr[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) qfr[qi] = f4scale(ld4_row(a.q, rowbase, qi * 16 + m, T, D, col4), a.scale);
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        if (qi >= NT) break;
        const int q = qi * 16 + m;
        const float4 qf = qfr[qi];
        float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool mine = !PAIR || (gq >> 1) == e;         // this lane's dims belong to head e
            const unsigned long long kw = shfl64(kw_own[e], qi * 16 + m);
            f32x4 s[4];
            float mx = -INFINITY;
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                s[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (kj <= qi) {
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    s[kj] = mfma_frag(mine ? kf[kj] : z, qf, s[kj]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = kj * 16 + 4 * gq + r;
                        s[kj][r] = (n > q) ? -INFINITY : s[kj][r];
                        mx = fmaxf(mx, s[kj][r]);
                    }
                }
            }
            mx = quad_group_max(mx);
            float l = 0.f;
            f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                if (kj <= qi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = kj * 16 + 4 * gq + r;
                        const float p = fast_exp(s[kj][r] - mx);
                        l += p;
                        const float pd = ((kw >> n) & 1ull) ? p * a.dscale : 0.f;
                        oacc = mfma4(vt[kj][r], pd, oacc);
                    }
                }
            }
            l = quad_group_sum(l);
            const float rl = 1.0f / l;
            if (mine) out = make_float4(oacc[0] * rl, oacc[1] * rl, oacc[2] * rl, oacc[3] * rl);
            if (q < T && gq == (PAIR ? 2 * e : 0) && a.stats) {
                float* sp = a.stats + ((rowbase + q) * H + (PAIR ? 2 * h + e : h)) * 2;
                sp[0] = mx; sp[1] = rl;
            }
        }
        if (q < T) st4(a.o + (rowbase + q) * D + col4, out);
    }
}


// ---- backward of ONE head of ONE sequence by the calling wave ---------------------------------------------------------------------
// a sequence's [T, D] slice of an activation tensor behind a buffer descriptor: rows past T read as zeros and stores to them vanish
// (the hardware's bounds check), so no load or store of the backward needs a branch or a select
typedef unsigned attn_v4u __attribute__((ext_vector_type(4)));
struct SeqBuf {
    __amdgpu_buffer_rsrc_t r;
    int ld;                                             // floats per row
    __device__ __forceinline__ SeqBuf(const float* base, long long rowbase, int T, int row_floats) : ld(row_floats) {
        const unsigned long long p = (unsigned long long)(base + rowbase * row_floats);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
        r = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, T * row_floats * 4, 0x00020000);
    }
    __device__ __forceinline__ float4 ld4(int row, int col) const {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (row * ld + col) * 4, 0, 0));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    // (no 8-byte load: hipcc 7.2 dropped the second dword of __builtin_amdgcn_raw_buffer_load_b64 in this kernel -- two ld1 instead)
    __device__ __forceinline__ float ld1(int row, int col) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (row * ld + col) * 4, 0, 0));
    }
    __device__ __forceinline__ void st4(int row, int col, float4 v) const {
#ifdef AMID_EXP_ATTN_NOSTORE      // (variant libraries only, results garbage: the backward launch without its stores -- and, the compiler
        return;                   // dropping what feeds them, without its products: 6.1 of 20.5 us)
#endif
        const f32x4 t = {v.x, v.y, v.z, v.w};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(attn_v4u, t), r, (row * ld + col) * 4, 0, 0);
    }
};

__device__ __forceinline__ float f4comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// one head's operands in their matrix-core lane layouts: row fragments of Q, K, V, dO, O (float4 along the head dim: lane (m, g) holds
// row 16 t + m, dims 4 g .. 4 g + 3), loaded from memory, and the transposed fragments K^T (phase 1), Q^T, dO^T (phase 2): lane (m, g)
// holds row 16 t + 4 g + r, dim m -- made from the row fragments by a 16 x 16 transpose through the wave's LDS scratch (as loads they
// were 48 of a head's 80 vector-memory operations: more than the 63 a wave can have in flight).  Rows past T are zeros.
struct AttnBwdOps {
    float4 kfr[4], vfr[4], qfr[4], dofr[4], ofr[4];
    float2 str[4];
    float kt[4][4], qts[4][4], dots[4][4];
    unsigned long long kw_own;                          // dropout keep word (64 keys) of query row `lane`
    float2 str2[4];                                     // head-dim-8 pairs only (PAIR): the tile's second head
    unsigned long long kw_own2;
};

// Every operand of BOTH phases is requested up front, without a wait, in two parts: what the forward saved (q, k, v, o, row statistics;
// the dropout keep word is drawn here too) and what the backward's predecessor produces (d_o).  A caller with registers to spare requests
// the saved part long before d_o exists (the fused per-sequence backward: under the last product of the chain that computes d_o).
// (A wave has at most 63 vector-memory operations in flight; a head's two parts are 28.)
// (PAIR: h is a 16-column tile = the head-dim-8 heads 2 h and 2 h + 1, see attn_fwd_head)
template <int NT, bool PAIR = false>
__device__ __forceinline__ void attn_bwd_load_saved(AttnBwdOps& o, const AttnArgs& a, int g, int b, long long rowbase, int h) {
    const int T = a.T, D = a.D, H = a.H;
    const int lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int col4 = h * AHD + 4 * gq;
    const int h0 = PAIR ? 2 * h : h;
    const SeqBuf bq(a.q, rowbase, T, D), bk(a.k, rowbase, T, D), bv(a.v, rowbase, T, D), bo(a.o, rowbase, T, D), bst(a.stats, rowbase, T, 2 * H);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        o.kfr[t] = bk.ld4(t * 16 + m, col4);
        o.vfr[t] = bv.ld4(t * 16 + m, col4);
        o.qfr[t] = bq.ld4(t * 16 + m, col4);
        o.ofr[t] = bo.ld4(t * 16 + m, col4);
        // (two dword loads: hipcc 7.2 drops the second half of a raw_buffer_load_b64 here)
        o.str[t] = make_float2(bst.ld1(t * 16 + m, 2 * h0), bst.ld1(t * 16 + m, 2 * h0 + 1));
        if constexpr (PAIR) o.str2[t] = make_float2(bst.ld1(t * 16 + m, 2 * h0 + 2), bst.ld1(t * 16 + m, 2 * h0 + 3));
    }
    o.kw_own = ~0ull;
    o.kw_own2 = ~0ull;
    if (a.train) {
        const int qrow = min(lane, T - 1);
        o.kw_own = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step,
                                 (unsigned long long)(b * H + h0) * T + qrow, T, a.thr16);
        if constexpr (PAIR)
            o.kw_own2 = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step,
                                      (unsigned long long)(b * H + h0 + 1) * T + qrow, T, a.thr16);
    }
}
template <int NT>
__device__ __forceinline__ void attn_bwd_load_dout(AttnBwdOps& o, const AttnArgs& a, long long rowbase, int h) {
    const int T = a.T, D = a.D;
    const int lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int col4 = h * AHD + 4 * gq;
    const SeqBuf bdo(a.d_o, rowbase, T, D);
#pragma unroll
    for (int t = 0; t < NT; ++t) o.dofr[t] = bdo.ld4(t * 16 + m, col4);
}

// the LDS scratch of a wave: two [16][20]-float transpose tiles -- ONE contiguous block per wave
constexpr int ATTN_BWD_TILE_FLOATS = 16 * 20;
constexpr int ATTN_BWD_LDS_PER_WAVE = 2 * ATTN_BWD_TILE_FLOATS * 4;

// this wave's 16 x 16 tile transposed through LDS: lane (m, g) hands in X[m][4 g .. 4 g + 3] and gets X[4 g + r][m], r = 0 .. 3.
// Rows of 20 floats: the 16-byte writes stay aligned, the dword reads of the four lane groups fall on two bank sets.  LDS operations
// of a wave execute in order: reusing the tile needs no wait for the previous reads.
__device__ __forceinline__ void tile_transpose(float* __restrict__ tile, const float4 in, float (&out)[4]) {
    const int lane = lane_id(), m = lane & 15, gq = lane >> 4;
    st4(tile + m * 20 + 4 * gq, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = tile[(4 * gq + r) * 20 + m];
}

// dq, dk, dv of the head's 16 columns in ONE pass over the (query tile, key tile) pairs of the causal triangle.  lds: this wave's
// ATTN_BWD_LDS_PER_WAVE bytes.  NT = ceil(T / 16) key / query tiles, all pairs computed without a branch (rows past T are zeros and
// their results are not stored).  Per pair, lanes = queries: S = Qs K^T and dP~ = dO V^T (8 matrix instructions over independent
// accumulators), the element-wise part with e = 2^(s log2(e) - m log2(e)), c = keep ? dscale / l : 0:  P~ = e c,
// dS = P (dP~ - delta) = e (dP c - delta / l), then dQ += dS K with dS as the second operand as it stands.  dV += P~^T dO and
// dK += dS^T Qs need the contraction index (the query) on the lane groups: the two 16 x 16 tiles are TRANSPOSED through LDS (one
// 16-byte write and four dword reads each) -- the first version recomputed S^T and dP~^T in a second pass with lanes = keys: 8 more
// matrix instructions and a second exponential per element, and the row statistics of every query travelled through LDS.
// 20 matrix instructions per pair instead of 28; dK / dV accumulate per key tile across the query tiles.
// PAIR (head dim 8, see attn_fwd_head): two passes over the pairs, one per head of the tile -- the K and V operands of S and dP~ zeroed
// in the other head's lane groups, that head's row statistics / keep word / delta; dq, dk, dv come out for all 16 dims and the lanes of the
// pass's head store theirs.
// (e_only >= 0, PAIR: only that head's pass -- a caller with a wave per HEAD runs the two passes of a tile on two waves side by side)
template <int NT, bool PAIR = false>
__device__ __forceinline__ void attn_bwd_compute(AttnBwdOps& o, const AttnArgs& a, long long rowbase, int h, float* __restrict__ lds, int e_only = -1) {
    const int T = a.T, D = a.D;
    const int lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int col4 = h * AHD + 4 * gq;
    const SeqBuf bdq(a.dq, rowbase, T, D), bdk(a.dk, rowbase, T, D), bdv(a.dv, rowbase, T, D);
    float* tile = lds;
    float* tile2 = lds + ATTN_BWD_TILE_FLOATS;
    // a query row's share of delta = sum dO . O, as soon as both are here: the O fragments (16 registers) are dead from then on -- with them
    // held through the core the standalone launch, at 256 registers per wave, spilled ten registers to scratch
    // (an explicit fma chain: every kernel this core is compiled into -- the standalone launch, the two fused backward builds -- gets
    // the same bits, whatever its surroundings make of a * b + c)
    float dparts[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 dof = o.dofr[t], of = o.ofr[t];
        dparts[t] = fmaf(dof.w, of.w, fmaf(dof.z, of.z, fmaf(dof.y, of.y, dof.x * of.x)));
        asm volatile("" : "+v"(dparts[t]));
    }
    // the transposed operand fragments K^T, Qs^T, dO^T from the row fragments
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        o.qfr[t] = f4scale(o.qfr[t], a.scale);
        tile_transpose(tile, o.kfr[t], o.kt[t]);
        tile_transpose(tile2, o.qfr[t], o.qts[t]);
        tile_transpose(tile, o.dofr[t], o.dots[t]);
    }
#pragma unroll
  for (int e = 0; e < (PAIR ? 2 : 1); ++e) {
    if (PAIR && e_only >= 0 && e != e_only) continue;      // (wave-uniform)
    const bool mine = !PAIR || (gq >> 1) == e;             // this lane's dims belong to the pass's head
    float4 kfm[4], vfm[4];                                 // K / V row fragments of the pass's head (zeros in the other head's lane groups)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        kfm[t] = mine ? o.kfr[t] : z;
        vfm[t] = mine ? o.vfr[t] : z;
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) { dk[kj] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[kj] = dk[kj]; }
#pragma unroll
    for (int qi = 0; qi < NT; ++qi) {
#ifdef AMID_EXP_ATTN_NOCOMPUTE    // (variant libraries only: loads, transposes and stores, no tile pair: 13.4 of 20.5 us)
        if (a.T > 0) continue;
#endif
        const int q = qi * 16 + m;
        const float4 qf = o.qfr[qi], dof = o.dofr[qi];
        const float dpart = dparts[qi];
        float delta;
        if constexpr (PAIR) {                              // a head's dims sit in two lane groups: sum those, fetch the other head's from across
            const float own = dpart + __shfl_xor(dpart, 16, 64);
            const float other = __shfl_xor(own, 32, 64);
            delta = mine ? own : other;
        } else {
            delta = quad_group_sum(dpart);
        }
        const float2 st = (PAIR && e == 1) ? o.str2[qi] : o.str[qi];
        const float ml = st.x * LOG2E, c1 = st.y * a.dscale, dr = delta * st.y;
        const unsigned long long kw = shfl64((PAIR && e == 1) ? o.kw_own2 : o.kw_own, q);
        const unsigned kwh[2] = {(unsigned)kw, (unsigned)(kw >> 32)};
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kj = 0; kj <= qi; ++kj) { s[kj] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kj] = s[kj]; }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int kj = 0; kj <= qi; ++kj) {
                s[kj] = mfma4(f4comp(kfm[kj], c), f4comp(qf, c), s[kj]);
                dp[kj] = mfma4(f4comp(vfm[kj], c), f4comp(dof, c), dp[kj]);
            }
        f32x4 dqa = f32x4{0.f, 0.f, 0.f, 0.f}, dqb = dqa;
#pragma unroll
        for (int kj = 0; kj <= qi; ++kj) {
            const unsigned bits = kwh[kj >> 1] >> ((kj & 1) * 16 + 4 * gq);              // this lane's keys 16 kj + 4 gq + r: bits 0..3
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = kj * 16 + 4 * gq + r;
                float e = __builtin_amdgcn_exp2f(fmaf(s[kj][r], LOG2E, -ml));
                if (kj == qi) e = (n > q) ? 0.f : e;                                    // the diagonal tile's upper triangle
                const float c = ((bits >> r) & 1u) ? c1 : 0.f;
                // (a query row past T has 1 / row sum = 0, i.e. c = 0 and delta / l = 0: P~ = dS = 0 without a test)
                pd[r] = e * c;
                ds[r] = e * fmaf(dp[kj][r], c, -dr);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (r & 1) dqb = mfma4(o.kt[kj][r], ds[r], dqb); else dqa = mfma4(o.kt[kj][r], ds[r], dqa); }
            float pt[4], dst[4];                     // lane (key m, gq), r <-> query 16 qi + 4 gq + r
            tile_transpose(tile, make_float4(pd[0], pd[1], pd[2], pd[3]), pt);
            tile_transpose(tile2, make_float4(ds[0], ds[1], ds[2], ds[3]), dst);
#pragma unroll
            for (int r = 0; r < 4; ++r) {           // (two chains alternate: a dependent instruction is two issue slots away)
                dv[kj] = mfma4(o.dots[qi][r], pt[r], dv[kj]);
                dk[kj] = mfma4(o.qts[qi][r], dst[r], dk[kj]);
            }
        }
        dqa += dqb;
        // (a lane outside the pass's head stores to row T: past the descriptor's range, i.e. nowhere)
        bdq.st4(mine ? q : T, col4, make_float4(dqa[0] * a.scale, dqa[1] * a.scale, dqa[2] * a.scale, dqa[3] * a.scale));
    }
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
        const int key = mine ? kj * 16 + m : T;
        bdk.st4(key, col4, make_float4(dk[kj][0], dk[kj][1], dk[kj][2], dk[kj][3]));
        bdv.st4(key, col4, make_float4(dv[kj][0], dv[kj][1], dv[kj][2], dv[kj][3]));
    }
  }
}

template <int NT, bool PAIR = false>
__device__ __forceinline__ void attn_bwd_head(const AttnArgs& a, int g, int b, long long rowbase, int h, float* __restrict__ lds, int e_only = -1) {
    AttnBwdOps o;
    STRIP_RSTAMP(16);                                   // (diagnostic builds of sasrec_strip.hip only)
    attn_bwd_load_saved<NT, PAIR>(o, a, g, b, rowbase, h);
    attn_bwd_load_dout<NT>(o, a, rowbase, h);
    STRIP_RSTAMP(17);
    attn_bwd_compute<NT, PAIR>(o, a, rowbase, h, lds, e_only);
}

}  // namespace amid
