// The device side of the fused head (head_fused.hip's kernels; the forward with the head on its tail, sasrec_seqn.hip): arguments, LDS carve and
// the phases of one sample's forward + loss + backward.
#pragma once
#include "common.h"

namespace amid {

// Every expression of the head compiles to the operations as written (fmaf where an fma is meant): the compiler's own choice of which
// product of a sum it contracts depends on the code around it, and this code runs in two places -- head_fused.hip's kernels and the tail of
// the forward's workgroups (sasrec_seqn.hip) -- that must agree bit for bit.  (Restored at the end of this header.)
#pragma clang fp contract(off)
// (common.h's float4 helpers are compiled under the default, contract(fast): their products and sums would stay contractible after inlining)
__device__ __forceinline__ float4 h4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 h4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 h4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float h4hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// workgroup barrier between LDS phases that says what it waits for: the LDS queue only.  (hipcc 7.2 compiles __syncthreads() to the same
// s_waitcnt lgkmcnt(0) + s_barrier for gfx950 -- no vmcnt(0) outside threadgroup-split mode, checked in the ISA -- so this changes no
// timing; it keeps the staged path from depending on that.)
__device__ __forceinline__ void head_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// diagnostic builds only (profiles/tools/head_stamps.py compiles this file with -DAMID_HEAD_STAMPS into its own library): real-time
// (100 MHz) stamps of workgroup 0's thread 0, in a buffer no kernel reads
#ifdef AMID_HEAD_STAMPS
static __device__ unsigned long long amid_head_stamp_buf[32];
#define HEAD_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) amid_head_stamp_buf[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define HEAD_STAMP(i) do { } while (0)
#endif

struct HeadArgs {
    const float* x;            // [2, B, T, D] output of the last encoder layer
    const float* lnw[2]; const float* lnb[2];   // last_layernorm (null: no LN, BERT4Rec)
    const float* items;        // [B, NI, D] gathered item rows
    const float* w1; const float* b1; const float* w2; const float* b2;
    const float* labels;       // [B, NI] or null
    const long long* domain;   // [B] or null
    float* u;                  // [2, B, D]
    float* p1; float* p2; float* dp1; float* dp2;   // [B, NI]
    float* loss_part;          // [B]
    // backward
    float* dx;                 // [2, B, T, D]
    float* ditems;             // [B, NI, D]
    float* ln_part;            // [2B][2][D]
    float* sc_part;            // [B][hid*2D + 2 hid + 1]
    const float* tr_src[32]; float* tr_dst[32]; int n_tr;   // square D x D transposes done by the extra blocks
    int B, T, NI, D, hid;
    float eps;
    float* hidg;               // optional [B][amid_scorer_vec_floats]: instead of the per-sample weight-gradient partials sc_part (32 KB a sample)
                               // only the sample's hidden gradients leave the workgroup -- da [2][hid], dc [NI][hid], dW2's [hid], db2's --
                               // and the gradient tail forms dW1 = sum_b da (x) u + dc (x) items itself (amid_grad_tail_live_f32)
    int own_only;              // fused train step over the live sequences: of row b only the sequence of its OWN domain (domain[b]) was
                               // encoded; the other domain's user vector reads as 0, gets no gradient and its rows are not touched
};

// a group of `n` consecutive threads of the workgroup working on one head (tid = index inside the group); the whole workgroup for
// the one-head kernels, 256-thread thirds in the three-head kernel (all groups run the same control flow: barriers stay aligned)
struct Tg { int tid, n; };
__device__ __forceinline__ Tg whole_block() { return Tg{(int)threadIdx.x, (int)blockDim.x}; }

__device__ __forceinline__ void stage_w1t(float* __restrict__ w1t, const float* __restrict__ w1, int D2, int hid, const Tg tg) {
    // w1t[e][j] = w1[j][e], row stride hid + 1.  A half-wave covers 4 rows j x 8 column quads c: its 32 stores of one component fall on
    // banks (4 c (hid + 1) + j) mod 32 = (4 c' + j') mod 32 (hid = 32 or 64), c' < 8, j' < 4 -- every bank once.  (Round 4 walked the quads
    // of ONE row with consecutive lanes: stride 4 (hid + 1), eight banks, every store four-way conflicted: a third of the kernel's LDS cycles.)
    const int q = D2 >> 2;                            // column quads per row: a multiple of 8
    const int l = tg.tid & 31, hw = tg.tid >> 5, n_hw = tg.n >> 5;
    const int qb = q >> 3;                            // blocks of 8 quads per row
    // (four blocks' loads in flight at once: one block per trip was four L2 latencies in a row at hid 32, 512 threads)
    const int nblk = (hid >> 2) * qb;
    for (int blk0 = hw; blk0 < nblk; blk0 += 4 * n_hw) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int blk = blk0 + k * n_hw;
            const int jb = blk / qb, cb = blk - jb * qb;
            const int j = 4 * jb + (l >> 3), c = 8 * cb + (l & 7);
            v[k] = blk < nblk ? ld4(w1 + (long long)j * D2 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int blk = blk0 + k * n_hw;
            if (blk >= nblk) continue;
            const int jb = blk / qb, cb = blk - jb * qb;
            const int j = 4 * jb + (l >> 3), c = 8 * cb + (l & 7);
            float* o = w1t + (4 * c) * (hid + 1) + j;
            o[0] = v[k].x; o[hid + 1] = v[k].y; o[2 * (hid + 1)] = v[k].z; o[3 * (hid + 1)] = v[k].w;
        }
    }
}

// LN_last + mean over T for (g, b): 8 row groups of 32 lanes; result u_s[D] (LDS) and u (global).
// Rows are fetched in chunks of 64 (8 per row group), every load of a chunk issued before the first use:
// the first version walked the rows one by one through three dependent global passes (mean, variance,
// normalise) and was pure load latency (33 us for a kernel that moves 13 MB).
constexpr int HEAD_CHUNK = 8;      // rows per row group per chunk

// 512 threads: threads 0..255 take domain 0, 256..511 domain 1 (both LayerNorm passes in flight at once: the kernel is one
// latency chain per workgroup, and there is one workgroup per CU); red [2][8][D], u_s [2][D]
__device__ __forceinline__ void lnmean_rows(const HeadArgs& a, int b, float* __restrict__ red_all, float* __restrict__ u_all) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int g = threadIdx.x >> 8;
    const int sub = threadIdx.x & 31, rg = (threadIdx.x >> 5) & 7;
    float* red = red_all + g * 8 * D;
    const bool use_ln = a.lnw[0] != nullptr;
    const float* w = a.lnw[g];
    const float* bb = a.lnb[g];
    const float* xb = a.x + ((long long)g * a.B + b) * T * D;
    {   // D <= 128: one float4 per lane covers the row; lanes past the row (D = 64) take part in the shuffles with zeros
        const int c = sub;
        const bool on = c < q;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (use_ln && on) { ww = ld4(w + 4 * c); b4 = ld4(bb + 4 * c); }
        for (int t0 = 0; t0 < T; t0 += 8 * HEAD_CHUNK) {
            float4 v[HEAD_CHUNK];
#pragma unroll
            for (int i = 0; i < HEAD_CHUNK; ++i) {
                const int t = t0 + rg + 8 * i;
                v[i] = (t < T && on) ? ld4(xb + (long long)t * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < HEAD_CHUNK; ++i) {
                const int t = t0 + rg + 8 * i;
                if (t < T) {                              // uniform over the 32 lanes of the row group
                    float4 y = v[i];
                    if (use_ln) {
                        float mean, rstd;
                        mean = group_sum<32>(h4hsum(y)) / D;
                        float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        rstd = 1.0f / sqrtf(group_sum<32>(h4hsum(h4mul(d4, d4))) / D + a.eps);
                        y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
                    }
                    acc = h4add(acc, y);
                }
            }
        }
        if (on) st4(red + rg * D + 4 * c, acc);
    }
    __syncthreads();
    for (int ge = threadIdx.x; ge < 2 * D; ge += blockDim.x) {
        const int g2 = ge / D, e = ge - g2 * D;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red_all[(g2 * 8 + k) * D + e];
        s /= T;
        u_all[ge] = s;
        a.u[((long long)g2 * a.B + b) * D + e] = s;
    }
    __syncthreads();
}

// own_only: the T rows of (domain[b], b) over all 16 row groups of the workgroup; red [16][D]
__device__ __forceinline__ void lnmean_rows_own(const HeadArgs& a, int b, float* __restrict__ red, float* __restrict__ u_all) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int own = a.domain[b] != 0 ? 1 : 0;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;          // 16 row groups
    const bool use_ln = a.lnw[0] != nullptr;
    const float* xb = a.x + ((long long)own * a.B + b) * T * D;
    constexpr int CH = HEAD_CHUNK / 2;
    {
        const int c = sub;
        const bool on = c < q;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (use_ln && on) { ww = ld4(a.lnw[own] + 4 * c); b4 = ld4(a.lnb[own] + 4 * c); }
        for (int t0 = 0; t0 < T; t0 += 16 * CH) {
            float4 v[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int t = t0 + rg + 16 * i;
                v[i] = (t < T && on) ? ld4(xb + (long long)t * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int t = t0 + rg + 16 * i;
                if (t < T) {
                    float4 y = v[i];
                    if (use_ln) {
                        const float mean = group_sum<32>(h4hsum(y)) / D;
                        float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(h4hsum(h4mul(d4, d4))) / D + a.eps);
                        y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
                    }
                    acc = h4add(acc, y);
                }
            }
        }
        if (on) st4(red + rg * D + 4 * c, acc);
    }
    __syncthreads();
    for (int ge = threadIdx.x; ge < 2 * D; ge += blockDim.x) {
        const int g2 = ge / D, e = ge - g2 * D;
        float s = 0.f;
        if (g2 == own) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * D + e];
            s /= T;
        }
        u_all[ge] = s;
        a.u[((long long)g2 * a.B + b) * D + e] = s;
    }
    __syncthreads();
}

// ---- own_only, forward + backward in one workgroup: the T rows of (domain[b], b) are loaded ONCE.  The loads are issued first (W1^T is
// staged while they fly), the rows are kept normalised in registers (xh = (x - mean) rstd; with 16 row groups a thread holds at most
// HEAD_CHUNK / 2 of them) together with their rstd, and the LayerNorm backward at the end of the workgroup's life reads them from there:
// no second pass over x, no second mean / variance.  Same operations in the same order as lnmean_rows_own / lnmean_rows_bwd_own: same bits.
struct OwnRows { float4 xh[HEAD_CHUNK / 2]; float rstd[HEAD_CHUNK / 2]; };

__device__ __forceinline__ void own_rows_load(const HeadArgs& a, int b, OwnRows& R) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int own = a.domain[b] != 0 ? 1 : 0;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = c < q;
    const float* xb = a.x + ((long long)own * a.B + b) * T * D;
#pragma unroll
    for (int i = 0; i < HEAD_CHUNK / 2; ++i) {
        const int t = rg + 16 * i;
        R.xh[i] = (t < T && on) ? ld4(xb + (long long)t * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        R.rstd[i] = 0.f;
    }
}

// T <= 16 * HEAD_CHUNK / 2 rows (the caller checks); red [16][D]
__device__ __forceinline__ void own_rows_lnmean(const HeadArgs& a, int b, int own, OwnRows& R, float* __restrict__ red, float* __restrict__ u_all) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = c < q;
    const bool use_ln = a.lnw[0] != nullptr;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (use_ln && on) { ww = ld4(a.lnw[own] + 4 * c); b4 = ld4(a.lnb[own] + 4 * c); }
#pragma unroll
    for (int i = 0; i < HEAD_CHUNK / 2; ++i) {
        const int t = rg + 16 * i;
        if (t < T) {                                  // uniform over the 32 lanes of the row group
            float4 y = R.xh[i];
            if (use_ln) {
                const float mean = group_sum<32>(h4hsum(y)) / D;
                float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float rstd = 1.0f / sqrtf(group_sum<32>(h4hsum(h4mul(d4, d4))) / D + a.eps);
                R.rstd[i] = rstd;
                R.xh[i] = h4scale(d4, rstd);
                y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
            }
            acc = h4add(acc, y);
        }
    }
    if (on) st4(red + rg * D + 4 * c, acc);
    head_lds_barrier();
    for (int ge = threadIdx.x; ge < 2 * D; ge += blockDim.x) {
        const int g2 = ge / D, e = ge - g2 * D;
        float s = 0.f;
        if (g2 == own) {
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * D + e];
            s /= T;
        }
        u_all[ge] = s;
        a.u[((long long)g2 * a.B + b) * D + e] = s;
    }
    head_lds_barrier();
}

// red [16][2][D]
__device__ __forceinline__ void own_rows_ln_bwd(const HeadArgs& a, int b, int own, const OwnRows& R, const float* __restrict__ du_all, float* __restrict__ red) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = c < q;
    const float* du_s = du_all + own * D;
    const bool use_ln = a.lnw[0] != nullptr;
    const float invT = 1.0f / T;
    const long long base = ((long long)own * a.B + b) * T * D;
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    const float4 dy = on ? h4scale(ld4(du_s + 4 * c), invT) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gy = dy;
    if (use_ln && on) gy = h4mul(dy, ld4(a.lnw[own] + 4 * c));
    const float c1 = use_ln ? group_sum<32>(h4hsum(gy)) / D : 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_CHUNK / 2; ++i) {
        const int t = rg + 16 * i;
        if (t < T) {
            float4 out = dy;
            if (use_ln) {
                const float4 xh = R.xh[i];
                const float rstd = R.rstd[i];
                const float c2 = group_sum<32>(h4hsum(h4mul(gy, xh))) / D;
                out = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                  rstd * (gy.w - c1 - xh.w * c2));
                dgam = h4add(dgam, h4mul(dy, xh));
                dbet = h4add(dbet, dy);
            }
            if (on) st4(a.dx + base + (long long)t * D + 4 * c, out);
        }
    }
    if (on) {
        st4(red + rg * 2 * D + 4 * c, dgam);
        st4(red + rg * 2 * D + D + 4 * c, dbet);
    }
    head_lds_barrier();
    if (use_ln) {
        for (int ge = threadIdx.x; ge < 4 * D; ge += blockDim.x) {
            const int g2 = ge / (2 * D), e = ge - g2 * 2 * D;
            float sacc = 0.f;
            if (g2 == own) {
#pragma unroll
                for (int k = 0; k < 16; ++k) sacc += red[k * 2 * D + e];
            }
            a.ln_part[((long long)g2 * a.B + b) * 2 * D + e] = sacc;
        }
    }
}

// LDS carve (floats): w1t [2D][hid+1] | u_s [2][D] | au [2][hid] | da [2][hid] | dw2 [hid+4] | ci [64][hid+1] | dc [64][hid+1] | scratch [16][D]
// `chunk` = items whose pre-activations are resident at once (64 for the one-head kernels; the three-head kernel keeps three
// carves in LDS and uses 16)
// LDS copies of a sample's small operands (head_own_rows_body stages them while W1^T is staged: every phase that read one of them from
// global paid an L2 latency of its own -- profiles/tools/head_stamps.py); all null: the phases read global memory
struct HeadStaged {
    const float *b1, *w2, *b2, *labels, *items;       // [hid], [hid], [1], [NI], [NI][D]
    float* pd;                                        // p1 [4] | p2 [4] | dp1 [4] | dp2 [4] of the sample's items (NI <= 4)
    int own;                                          // the sample's domain (with b1 != null)
};
struct HeadLds {
    float *w1t, *u_s, *au, *da, *dw2, *ci, *dc, *scr;
    int chunk;
    HeadStaged st = {};
    __device__ HeadLds(float* base, int D, int hid, int chunk_ = 64) {
        chunk = chunk_;
        w1t = base; u_s = w1t + 2 * D * (hid + 1); au = u_s + 2 * D; da = au + 2 * hid; dw2 = da + 2 * hid;
        ci = dw2 + hid + 4; dc = ci + chunk * (hid + 1); scr = dc + chunk * (hid + 1);
        scr = base + ((scr - base + 3) & ~3);
    }
};
// (head_lds_barrier, above, wherever the sample's operands are staged: nothing crosses the threads through global memory then)
__device__ __forceinline__ void head_sync(const HeadLds& s) {
    if (s.st.b1 != nullptr) head_lds_barrier(); else __syncthreads();
}
__host__ __device__ inline size_t head_carve_floats(int D, int hid, int chunk) {
    size_t f = (size_t)2 * D * (hid + 1) + 2 * D + 4 * hid + hid + 4 + 2 * chunk * (hid + 1);
    return (f + 3) & ~(size_t)3;
}
__host__ __device__ inline size_t head_lds_floats(int D, int hid) {
    return head_carve_floats(D, hid, 64) + 32 * D;          // scratch: [2][8][2][D] partials of the LayerNorm backward (forward uses half)
}

// au[d][j] = b1[j] + sum_e w1t[e][j] u_d[e].  Eight lanes per output, each summing every eighth e (a thread per output walked D
// dependent fmas: 2.2 us of the 22 this kernel's workgroup lives -- profiles/tools/head_stamps.py).
__device__ __forceinline__ void user_half(const HeadArgs& a, const HeadLds& s, const Tg tg) {
    const int D = a.D, hid = a.hid;
    const int part = tg.tid & 7;
    for (int o0 = 0; o0 < 2 * hid; o0 += tg.n >> 3) {             // (uniform trip count: the shuffles below need every lane)
        const int dj = o0 + (tg.tid >> 3);
        const bool on = dj < 2 * hid;
        const int d = on ? dj / hid : 0, j = on ? dj - d * hid : 0;
        float acc = 0.f;
        if (on) {
            const float* ur = s.u_s + d * D;
#pragma unroll 8
            for (int e = part; e < D; e += 8) acc = fmaf(s.w1t[e * (hid + 1) + j], ur[e], acc);
        }
        acc = group_sum<8>(acc);
        if (on && part == 0) s.au[dj] = acc + (s.st.b1 != nullptr ? s.st.b1[j] : a.b1[j]);
    }
}
// ci[n][j] = sum_e w1t[D+e][j] item[n][e] for the chunk's items (items read from global).  Eight lanes per output, each taking every
// eighth column quad of the item row: the eight lanes of an output read 128 consecutive bytes.
__device__ __forceinline__ void item_half(const HeadArgs& a, const HeadLds& s, int b, int n0, int nn, const Tg tg) {
    const int D = a.D, hid = a.hid;
    const int part = tg.tid & 7;
    for (int o0 = 0; o0 < nn * hid; o0 += tg.n >> 3) {
        const int nj = o0 + (tg.tid >> 3);
        const bool on = nj < nn * hid;
        const int n = on ? nj / hid : 0, j = on ? nj - n * hid : 0;
        float acc = 0.f;
        if (on) {
            const float* ir = s.st.items != nullptr ? s.st.items + (n0 + n) * D : a.items + ((long long)b * a.NI + n0 + n) * D;
#pragma unroll 4
            for (int e = 4 * part; e < D; e += 32) {
                const float4 it = ld4(ir + e);
                const float* wp = s.w1t + (D + e) * (hid + 1) + j;
                acc = fmaf(wp[0], it.x, acc); acc = fmaf(wp[hid + 1], it.y, acc);
                acc = fmaf(wp[2 * (hid + 1)], it.z, acc); acc = fmaf(wp[3 * (hid + 1)], it.w, acc);
            }
        }
        acc = group_sum<8>(acc);
        if (on && part == 0) s.ci[n * (hid + 1) + j] = acc;
    }
}

// scorer forward for row b from the user vectors in s.u_s (W1^T staged in s.w1t): p1 / p2, and with labels the masked BCE
// partial + dLoss/dp
__device__ __forceinline__ void scorer_fwd_part(const HeadArgs& a, const HeadLds& s, int b, const Tg tg) {
    const int hid = a.hid, NI = a.NI;
    user_half(a, s, tg);
    HEAD_STAMP(3);
    float lsum = 0.f;
    const int CH = s.chunk;
    for (int n0 = 0; n0 < NI; n0 += CH) {
        const int nn = min(CH, NI - n0);
        if (!(s.st.b1 != nullptr && n0 == 0)) head_sync(s);        // (staged: the first chunk's item half runs beside the user half -- nobody reads s.ci yet)
        item_half(a, s, b, n0, nn, tg);
        head_sync(s);
        HEAD_STAMP(4);
        // 32 lanes per logit, one hidden unit each (a thread per logit walked `hid` dependent loads and fmas)
        for (int nd0 = 0; nd0 < nn * 2; nd0 += tg.n >> 5) {
            const int nd = nd0 + (tg.tid >> 5), j0 = tg.tid & 31;
            const bool on = nd < nn * 2;
            const int n = on ? nd >> 1 : 0, d = nd & 1;
            float zp = 0.f;
            const float* w2p = s.st.w2 != nullptr ? s.st.w2 : a.w2;
            if (on) for (int j = j0; j < hid; j += 32) zp = fmaf(w2p[j], fmaxf(s.au[d * hid + j] + s.ci[n * (hid + 1) + j], 0.f), zp);
            const float z = group_sum<32>(zp) + (s.st.b2 != nullptr ? s.st.b2[0] : a.b2[0]);
            if (!on || j0 != 0) continue;
            const float p = 1.0f / (1.0f + expf(-z));
            const long long o = (long long)b * NI + n0 + n;
            (d ? a.p2 : a.p1)[o] = p;
            if (s.st.pd != nullptr) s.st.pd[4 * d + n] = p;
            if (a.labels) {
                const float y = s.st.labels != nullptr ? s.st.labels[n0 + n] : a.labels[o];
                const bool dom1 = s.st.b1 != nullptr ? s.st.own != 0 : a.domain[b] != 0;
                const float md = dom1 ? (d ? 1.f : 0.f) : (d ? 0.f : 1.f);
                const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.0f - p), -100.f);
                const float inv = 1.0f / ((float)a.B * (float)NI);
                lsum += -(y * lp + (1.f - y) * l1p) * md * inv;
                if (s.st.b1 != nullptr && NI == 2) s.scr[nd] = lsum;      // (the logit's term: summed behind the caller's next barrier, head_loss_terms)
                const float dp = md * inv * (p - y) / fmaxf((1.f - p) * p, 1e-12f);   // torch binary_cross_entropy_backward
                (d ? a.dp2 : a.dp1)[o] = dp;
                if (s.st.pd != nullptr) s.st.pd[8 + 4 * d + n] = dp;
            }
        }
    }
    HEAD_STAMP(5);
    if (a.labels && !(s.st.b1 != nullptr && NI == 2)) {
        head_sync(s);
        lsum = group_sum<64>(lsum);
        if (lane_id() == 0) s.scr[wave_id()] = lsum;
        head_sync(s);
        if (tg.tid == 0) a.loss_part[b] = ((s.scr[0] + s.scr[1]) + (s.scr[2] + s.scr[3])) + ((s.scr[4] + s.scr[5]) + (s.scr[6] + s.scr[7]));
    }
}

// the staged two-item path: the four logits' loss terms (s.scr[nd], nd = 2 item + domain) in the order the workgroup reduction above adds
// them -- wave 0 holds the terms of item 0, wave 1 those of item 1, the other waves zeros: (t0 + t1) + (t2 + t3), the same bits
__device__ __forceinline__ void head_loss_terms(const HeadArgs& a, const HeadLds& s, int b, const Tg tg) {
    if (tg.tid == 0) a.loss_part[b] = (s.scr[0] + s.scr[1]) + (s.scr[2] + s.scr[3]);
}

__device__ __forceinline__ void head_fwd_body(const HeadArgs& a, float* __restrict__ sm, int b) {
    const HeadLds s(sm, a.D, a.hid);
    HEAD_STAMP(0);
    stage_w1t(s.w1t, a.w1, 2 * a.D, a.hid, whole_block());
    HEAD_STAMP(1);
    if (a.own_only) lnmean_rows_own(a, b, s.scr, s.u_s); else lnmean_rows(a, b, s.scr, s.u_s);
    HEAD_STAMP(2);
    scorer_fwd_part(a, s, b, whole_block());
    HEAD_STAMP(6);
}

// dx rows of (g, b) from du_s[D] (LDS): dx = LN_last'(du / T ; x) ; partial d gamma / d beta -> ln_part[(g*B+b)][2][D]
__device__ __forceinline__ void lnmean_rows_bwd(const HeadArgs& a, int b, const float* __restrict__ du_all /* [2][D] */,
                                                float* __restrict__ red_all /* [2][8][2][D] */) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int g = threadIdx.x >> 8;
    const int sub = threadIdx.x & 31, rg = (threadIdx.x >> 5) & 7;
    const float* du_s = du_all + g * D;
    float* red = red_all + g * 16 * D;
    const bool use_ln = a.lnw[0] != nullptr;
    const float* w = a.lnw[g];
    const float invT = 1.0f / T;
    const long long base = ((long long)g * a.B + b) * T * D;
    {
        const int c = sub;
        const bool on = c < q;
        float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
        const float4 dy = on ? h4scale(ld4(du_s + 4 * c), invT) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gy = dy;
        if (use_ln && on) gy = h4mul(dy, ld4(w + 4 * c));
        const float c1 = use_ln ? group_sum<32>(h4hsum(gy)) / D : 0.f;      // same for every row: dy does not depend on t
        for (int t0 = 0; t0 < T; t0 += 8 * HEAD_CHUNK) {
            float4 v[HEAD_CHUNK];
            if (use_ln) {
#pragma unroll
                for (int i = 0; i < HEAD_CHUNK; ++i) {
                    const int t = t0 + rg + 8 * i;
                    v[i] = (t < T && on) ? ld4(a.x + base + (long long)t * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < HEAD_CHUNK; ++i) {
                const int t = t0 + rg + 8 * i;
                if (t < T) {
                    float4 out = dy;
                    if (use_ln) {
                        const float mean = group_sum<32>(h4hsum(v[i])) / D;
                        float4 d4 = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(h4hsum(h4mul(d4, d4))) / D + a.eps);
                        const float4 xh = h4scale(d4, rstd);
                        const float c2 = group_sum<32>(h4hsum(h4mul(gy, xh))) / D;
                        out = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                          rstd * (gy.w - c1 - xh.w * c2));
                        dgam = h4add(dgam, h4mul(dy, xh));
                        dbet = h4add(dbet, dy);
                    }
                    if (on) st4(a.dx + base + (long long)t * D + 4 * c, out);
                }
            }
        }
        if (on) {
            st4(red + rg * 2 * D + 4 * c, dgam);
            st4(red + rg * 2 * D + D + 4 * c, dbet);
        }
    }
    __syncthreads();
    if (use_ln) {
        for (int ge = threadIdx.x; ge < 4 * D; ge += blockDim.x) {
            const int g2 = ge / (2 * D), e = ge - g2 * 2 * D;
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sacc += red_all[g2 * 16 * D + k * 2 * D + e];
            a.ln_part[((long long)g2 * a.B + b) * 2 * D + e] = sacc;
        }
    }
    __syncthreads();
}

// own_only: dx rows of (domain[b], b) only, over all 16 row groups; the other domain's LayerNorm-partial slot is zeroed; red [16][2][D]
__device__ __forceinline__ void lnmean_rows_bwd_own(const HeadArgs& a, int b, const float* __restrict__ du_all, float* __restrict__ red) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int own = a.domain[b] != 0 ? 1 : 0;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const float* du_s = du_all + own * D;
    const bool use_ln = a.lnw[0] != nullptr;
    const float invT = 1.0f / T;
    const long long base = ((long long)own * a.B + b) * T * D;
    constexpr int CH = HEAD_CHUNK / 2;
    {
        const int c = sub;
        const bool on = c < q;
        float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
        const float4 dy = on ? h4scale(ld4(du_s + 4 * c), invT) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gy = dy;
        if (use_ln && on) gy = h4mul(dy, ld4(a.lnw[own] + 4 * c));
        const float c1 = use_ln ? group_sum<32>(h4hsum(gy)) / D : 0.f;
        for (int t0 = 0; t0 < T; t0 += 16 * CH) {
            float4 v[CH];
            if (use_ln) {
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int t = t0 + rg + 16 * i;
                    v[i] = (t < T && on) ? ld4(a.x + base + (long long)t * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int t = t0 + rg + 16 * i;
                if (t < T) {
                    float4 out = dy;
                    if (use_ln) {
                        const float mean = group_sum<32>(h4hsum(v[i])) / D;
                        float4 d4 = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(h4hsum(h4mul(d4, d4))) / D + a.eps);
                        const float4 xh = h4scale(d4, rstd);
                        const float c2 = group_sum<32>(h4hsum(h4mul(gy, xh))) / D;
                        out = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                          rstd * (gy.w - c1 - xh.w * c2));
                        dgam = h4add(dgam, h4mul(dy, xh));
                        dbet = h4add(dbet, dy);
                    }
                    if (on) st4(a.dx + base + (long long)t * D + 4 * c, out);
                }
            }
        }
        if (on) {
            st4(red + rg * 2 * D + 4 * c, dgam);
            st4(red + rg * 2 * D + D + 4 * c, dbet);
        }
    }
    __syncthreads();
    if (use_ln) {
        for (int ge = threadIdx.x; ge < 4 * D; ge += blockDim.x) {
            const int g2 = ge / (2 * D), e = ge - g2 * 2 * D;
            float sacc = 0.f;
            if (g2 == own) {
#pragma unroll
                for (int k = 0; k < 16; ++k) sacc += red[k * 2 * D + e];
            }
            a.ln_part[((long long)g2 * a.B + b) * 2 * D + e] = sacc;
        }
    }
    __syncthreads();
}

// FUSED = true: called right after head_fwd_body in the same workgroup -- W1^T, the user vectors and their hidden pre-activations
// (s.w1t, s.u_s, s.au) are still in LDS, and with NI <= 64 so are the item pre-activations (s.ci)
// extra workgroups (blockIdx >= B): out[j][i] = in[i][j] for the projection weights, 32x32 tiles
__device__ __forceinline__ void transpose_extra(const HeadArgs& a, float* __restrict__ sm) {
    const int D = a.D;
    const int tiles = D / 32, per = tiles * tiles;
    float (*tile)[33] = reinterpret_cast<float (*)[33]>(sm);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int w = (int)blockIdx.x - a.B; w < a.n_tr * per; w += (int)gridDim.x - a.B) {
        const int m = w / per, tt = w - m * per, bx = (tt % tiles) * 32, by = (tt / tiles) * 32;
        const float* __restrict__ src = a.tr_src[m];
        float* __restrict__ dst = a.tr_dst[m];
        __syncthreads();
        for (int r = ty; r < 32; r += 16) tile[r][tx] = src[(long long)(by + r) * D + bx + tx];
        __syncthreads();
        for (int r = ty; r < 32; r += 16) dst[(long long)(bx + r) * D + by + tx] = tile[tx][r];
    }
}

// scorer backward for row b: d items (ACC: added to what another head wrote), the head's weight-gradient partials, and d u[2][D]
// left in LDS (returned pointer; a region of s.ci).
// FUSED = true: called right after scorer_fwd_part in the same workgroup -- W1^T, the user vectors and their hidden pre-activations
// (s.w1t, s.u_s, s.au) are still in LDS, and with NI <= 64 so are the item pre-activations (s.ci)
template <bool FUSED, bool ACC>
__device__ __forceinline__ float* scorer_bwd_part(const HeadArgs& a, const HeadLds& s, int b, const Tg tg, float* __restrict__ dit_lds = nullptr) {
    const int D = a.D, hid = a.hid, NI = a.NI;
    const int P = (hid * 2 * D + 2 * hid + 1 + 3) & ~3;      // amid_scorer_part_floats: rows padded to whole float4s
    float* part = a.sc_part + (long long)b * P;
    if (!FUSED) {
        stage_w1t(s.w1t, a.w1, 2 * D, hid, tg);
        for (int e = tg.tid; e < 2 * D; e += tg.n) s.u_s[e] = a.u[((long long)(e / D) * a.B + b) * D + (e % D)];
    }
    for (int e = tg.tid; e < 2 * hid; e += tg.n) s.da[e] = 0.f;
    for (int e = tg.tid; e < hid + 1; e += tg.n) s.dw2[e] = 0.f;
    head_sync(s);
    if (!FUSED) user_half(a, s, tg);
    // item half of dW1 accumulates over item chunks in registers: thread owns (j, e) pairs je = tid + 256 k
    const int CH = s.chunk;
    for (int n0 = 0; n0 < NI; n0 += CH) {
        const int nn = min(CH, NI - n0);
        const bool one_pass = FUSED && NI <= CH && s.st.b1 != nullptr;       // (staged, one chunk: nothing happens between these barriers)
        if (!one_pass) head_sync(s);
        if (!(FUSED && NI <= CH)) item_half(a, s, b, n0, nn, tg);
        if (!one_pass) head_sync(s);
        if (tg.tid < hid) {                       // hidden unit j walks the chunk's items in order
            const int j = tg.tid;
            float s_da0 = 0.f, s_da1 = 0.f, s_w2 = 0.f;
            const float w2j = s.st.w2 != nullptr ? s.st.w2[j] : a.w2[j];
            const float* pd = s.st.pd;                 // (staged: NI <= 4, one chunk)
            for (int n = 0; n < nn; ++n) {
                const long long o = (long long)b * NI + n0 + n;
                const float p1 = pd != nullptr ? pd[n] : a.p1[o], p2 = pd != nullptr ? pd[4 + n] : a.p2[o];
                const float dz1 = (pd != nullptr ? pd[8 + n] : a.dp1[o]) * p1 * (1.f - p1), dz2 = (pd != nullptr ? pd[12 + n] : a.dp2[o]) * p2 * (1.f - p2);
                const float c = s.ci[n * (hid + 1) + j];
                const float h1 = fmaxf(s.au[j] + c, 0.f), h2 = fmaxf(s.au[hid + j] + c, 0.f);
                const float g1 = h1 > 0.f ? dz1 * w2j : 0.f, g2 = h2 > 0.f ? dz2 * w2j : 0.f;
                s_w2 += dz1 * h1 + dz2 * h2;
                s_da0 += g1; s_da1 += g2;
                s.dc[n * (hid + 1) + j] = g1 + g2;
            }
            s.da[j] += s_da0; s.da[hid + j] += s_da1; s.dw2[j] += s_w2;
        }
        if (tg.tid == 64) {
            float acc = 0.f;
            for (int n = 0; n < nn; ++n) {
                const long long o = (long long)b * NI + n0 + n;
                const float* pd = s.st.pd;
                const float p1 = pd != nullptr ? pd[n] : a.p1[o], p2 = pd != nullptr ? pd[4 + n] : a.p2[o];
                acc += (pd != nullptr ? pd[8 + n] : a.dp1[o]) * p1 * (1.f - p1) + (pd != nullptr ? pd[12 + n] : a.dp2[o]) * p2 * (1.f - p2);
            }
            s.dw2[hid] += acc;
        }
        head_sync(s);
        HEAD_STAMP(8);
        if (one_pass && a.hidg != nullptr && dit_lds == nullptr && !ACC && tg.n == 512 && nn * D <= 256 && 2 * D <= 256) {
            // the training step's shape: d items on the first four waves, d u on the other four (both read only what the barrier above
            // published; d u goes where the item pre-activations were), the hidden gradients leave, one barrier -- the same sums
            float* du_s = s.ci;
            if (tg.tid < 256) {
                const int ne = tg.tid;
                if (ne < nn * D) {
                    const int n = ne / D, e = ne - n * D;
                    float acc = 0.f;
                    const float* wp = s.w1t + (D + e) * (hid + 1);
#pragma unroll 8
                    for (int j = 0; j < hid; ++j) acc = fmaf(s.dc[n * (hid + 1) + j], wp[j], acc);
                    a.ditems[((long long)b * NI + n) * D + e] = acc;
                }
            } else {
                const int de = tg.tid - 256;
                if (de < 2 * D) {
                    const int d = de / D, e = de - d * D;
                    float acc = 0.f;
                    const float* wp = s.w1t + e * (hid + 1);
#pragma unroll 8
                    for (int j = 0; j < hid; ++j) acc = fmaf(s.da[d * hid + j], wp[j], acc);
                    du_s[de] = acc;
                }
            }
            float* hg = a.hidg + (long long)b * (((3 + NI) * hid + 1 + 3) & ~3);
            for (int nj = tg.tid; nj < nn * hid; nj += tg.n) hg[2 * hid + nj] = s.dc[(nj / hid) * (hid + 1) + (nj % hid)];
            for (int j = tg.tid; j < 2 * hid; j += tg.n) hg[j] = s.da[j];
            for (int j = tg.tid; j < hid + 1; j += tg.n) hg[(2 + NI) * hid + j] = s.dw2[j];
            HEAD_STAMP(9);
            head_sync(s);
            return du_s;
        }
        // d item[n][e] = sum_j dc[n][j] w1t[D+e][j]
        for (int ne = tg.tid; ne < nn * D; ne += tg.n) {
            const int n = ne / D, e = ne - n * D;
            float acc = 0.f;
            const float* wp = s.w1t + (D + e) * (hid + 1);
#pragma unroll 8
            for (int j = 0; j < hid; ++j) acc = fmaf(s.dc[n * (hid + 1) + j], wp[j], acc);
            if (dit_lds != nullptr) { dit_lds[ne] = acc; continue; }        // single chunk: the caller sums the heads' shares
            float* dst = a.ditems + ((long long)b * NI + n0 + n) * D + e;
            *dst = ACC ? *dst + acc : acc;
        }
        if (a.hidg != nullptr) {                   // (single chunk: the launcher checks NI <= chunk) dc of every item: the tail multiplies
            float* hg = a.hidg + (long long)b * (((3 + NI) * hid + 1 + 3) & ~3) + 2 * hid;
            for (int nj = tg.tid; nj < nn * hid; nj += tg.n) hg[nj] = s.dc[(nj / hid) * (hid + 1) + (nj % hid)];
        }
        // dW1[j][D+e] (+)= sum_n dc[n][j] item[n][e]
        if (a.hidg == nullptr)
        for (int je = tg.tid; je < hid * D; je += tg.n) {
            const int j = je / D, e = je - j * D;
            float acc = 0.f;
            for (int n = 0; n < nn; ++n) acc = fmaf(s.dc[n * (hid + 1) + j], a.items[((long long)b * NI + n0 + n) * D + e], acc);
            float* dst = part + j * 2 * D + D + e;
            *dst = (n0 == 0) ? acc : *dst + acc;
        }
    }
    head_sync(s);
    HEAD_STAMP(9);
    // user halves: du_d[e] = sum_j da[d][j] w1t[e][j]  -> scratch region [2][D] reused from ci (dead now)
    float* du_s = s.ci;
    for (int de = tg.tid; de < 2 * D; de += tg.n) {
        const int d = de / D, e = de - d * D;
        float acc = 0.f;
        const float* wp = s.w1t + e * (hid + 1);
#pragma unroll 8
        for (int j = 0; j < hid; ++j) acc = fmaf(s.da[d * hid + j], wp[j], acc);
        du_s[de] = acc;
    }
    if (a.hidg != nullptr) {
        float* hg = a.hidg + (long long)b * (((3 + NI) * hid + 1 + 3) & ~3);
        for (int j = tg.tid; j < 2 * hid; j += tg.n) hg[j] = s.da[j];
        for (int j = tg.tid; j < hid + 1; j += tg.n) hg[(2 + NI) * hid + j] = s.dw2[j];
        head_sync(s);
        return du_s;
    }
    for (int je = tg.tid; je < hid * D; je += tg.n) {
        const int j = je / D, e = je - j * D;
        part[j * 2 * D + e] = s.da[j] * s.u_s[e] + s.da[hid + j] * s.u_s[D + e];
    }
    for (int j = tg.tid; j < hid; j += tg.n) {
        part[hid * 2 * D + j] = s.da[j] + s.da[hid + j];            // db1
        part[hid * 2 * D + hid + j] = s.dw2[j];                     // dW2
    }
    if (tg.tid == 0) part[hid * 2 * D + 2 * hid] = s.dw2[hid];   // db2
    head_sync(s);
    return du_s;
}

template <bool FUSED>
__device__ __forceinline__ void head_bwd_body(const HeadArgs& a, float* __restrict__ sm) {
    if ((int)blockIdx.x >= a.B) { transpose_extra(a, sm); return; }
    const HeadLds s(sm, a.D, a.hid);
    const int b = blockIdx.x;
    HEAD_STAMP(7);
    float* du_s = scorer_bwd_part<FUSED, false>(a, s, b, whole_block());
    HEAD_STAMP(10);
    if (a.own_only) lnmean_rows_bwd_own(a, b, du_s, s.scr); else lnmean_rows_bwd(a, b, du_s, s.scr);
    HEAD_STAMP(11);
}

// W1^T's staging split into its loads and its LDS stores for the training shape (D2 256, hid 32, 512 threads: exactly four blocks per
// half-wave), so that another phase runs between them
__device__ __forceinline__ void stage_w1t_load4(float4 (&v)[4], const float* __restrict__ w1, const Tg tg) {
    constexpr int D2 = 256, qb = 8;
    const int l = tg.tid & 31, hw = tg.tid >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int blk = hw + 16 * k, jb = blk / qb, cb = blk - jb * qb;
        const int j = 4 * jb + (l >> 3), c = 8 * cb + (l & 7);
        v[k] = ld4(w1 + (long long)j * D2 + 4 * c);
    }
}
__device__ __forceinline__ void stage_w1t_store4(float* __restrict__ w1t, const float4 (&v)[4], const Tg tg) {
    constexpr int hid = 32, qb = 8;
    const int l = tg.tid & 31, hw = tg.tid >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int blk = hw + 16 * k, jb = blk / qb, cb = blk - jb * qb;
        const int j = 4 * jb + (l >> 3), c = 8 * cb + (l & 7);
        float* o = w1t + (4 * c) * (hid + 1) + j;
        o[0] = v[k].x; o[hid + 1] = v[k].y; o[2 * (hid + 1)] = v[k].z; o[3 * (hid + 1)] = v[k].w;
    }
}

// CONST_SHAPE: D 128, hid 32, NI 2, labels given, 512 threads -- the training shape, its sizes compile-time constants of `a` (see
// head_own_rows_body): every loop of the phases is unrolled with its LDS reads issued together (with run-time bounds each trip waited for
// its own reads: the phases were chains of LDS latencies).
// UNDER_LN (the forward's tail, 256 registers a wave; head_fused.hip's kernel has 128 and its rows' loads in flight at that point anyway):
// the loads of W1 and of the small operands are issued in front of the LayerNorm and stored to LDS behind it.
template <bool CONST_SHAPE, bool UNDER_LN>
__device__ __forceinline__ void head_own_rows_impl(const HeadArgs& a, float* __restrict__ sm, int b, int own, OwnRows& R) {
    HeadLds s(sm, a.D, a.hid);
    const Tg tg = whole_block();
    // the sample's small operands go into the second half of the scratch (the LayerNorm forward uses the first [16][D]; the LayerNorm
    // backward, the last phase, all of it): b1 | w2 | b2 | labels | p / dLoss/dp | item rows
    float* stg = s.scr + 16 * a.D;
    const int hid = a.hid, q = a.D >> 2;
    const int o_lab = (2 * hid + 4 + 3) & ~3;
    float* its = stg + o_lab + 4 + 16;
    const bool staged = a.NI <= 4 && a.labels != nullptr;
    if (staged) {
        s.st.b1 = stg; s.st.w2 = stg + hid; s.st.b2 = stg + 2 * hid; s.st.labels = stg + o_lab;
        s.st.pd = stg + o_lab + 4; s.st.items = its; s.st.own = own;
    }
    if constexpr (CONST_SHAPE && UNDER_LN) {
        float4 wv[4];
        stage_w1t_load4(wv, a.w1, tg);
        const int i = tg.tid;
        float cv = 0.f, lv = 0.f;
        float4 itv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < 2 * hid + 1) cv = i < hid ? a.b1[i] : i < 2 * hid ? a.w2[i - hid] : a.b2[0];
        if (i < a.NI) lv = a.labels[(long long)b * a.NI + i];
        if (i < a.NI * q) itv = ld4(a.items + (long long)b * a.NI * a.D + 4 * i);
        HEAD_STAMP(1);
        own_rows_lnmean(a, b, own, R, s.scr, s.u_s);
        stage_w1t_store4(s.w1t, wv, tg);
        if (i < 2 * hid + 1) stg[i] = cv;
        if (i < a.NI) stg[o_lab + i] = lv;
        if (i < a.NI * q) st4(its + 4 * i, itv);
        head_lds_barrier();
    } else {
        if (staged) {
            for (int i = tg.tid; i < 2 * hid + 1; i += tg.n) stg[i] = i < hid ? a.b1[i] : i < 2 * hid ? a.w2[i - hid] : a.b2[0];
            if (tg.tid < a.NI) stg[o_lab + tg.tid] = a.labels[(long long)b * a.NI + tg.tid];
            for (int i = tg.tid; i < a.NI * q; i += tg.n) st4(its + 4 * i, ld4(a.items + (long long)b * a.NI * a.D + 4 * i));
        }
        stage_w1t(s.w1t, a.w1, 2 * a.D, a.hid, tg);
        HEAD_STAMP(1);
        own_rows_lnmean(a, b, own, R, s.scr, s.u_s);
    }
    HEAD_STAMP(2);
    scorer_fwd_part(a, s, b, tg);
    HEAD_STAMP(6);
    if (s.st.b1 != nullptr) {
        head_lds_barrier();                           // (p and dLoss/dp cross the threads through s.st.pd)
        if (a.NI == 2) head_loss_terms(a, s, b, tg);
    } else {
        __threadfence_block();                        // dLoss/dp written above is read by other threads of this workgroup below
        __syncthreads();
    }
    HEAD_STAMP(7);
    float* du_s = scorer_bwd_part<true, false>(a, s, b, tg);
    HEAD_STAMP(10);
    own_rows_ln_bwd(a, b, own, R, du_s, s.scr);
    HEAD_STAMP(11);
}

// ---- the live-sequence train step's head behind the loading of the sample's rows: W1^T staged, LayerNorm + mean over T, scorer forward + loss,
// scorer backward, LayerNorm backward (dx rows, ln_part).  `sm`: head_lds_floats(D, hid) floats of LDS; 512 threads; T <= 16 HEAD_CHUNK / 2.
template <bool UNDER_LN = false>
__device__ __forceinline__ void head_own_rows_body(const HeadArgs& a_, float* __restrict__ sm, int b, int own, OwnRows& R) {
    if (a_.D == 128 && a_.hid == 32 && a_.NI == 2 && a_.labels != nullptr && blockDim.x == 512) {
        HeadArgs a = a_;
        a.D = 128; a.hid = 32; a.NI = 2;
        head_own_rows_impl<true, UNDER_LN>(a, sm, b, own, R);
    } else {
        head_own_rows_impl<false, false>(a_, sm, b, own, R);
    }
}

// the rows from an LDS image xl [T][ld] (the encoder's last output still in its workgroup: sasrec_seqn.hip's forward with the head on its
// tail) instead of own_rows_load's global loads; the same lane <-> (row, column quad) assignment
__device__ __forceinline__ void own_rows_take_lds(const float* __restrict__ xl, int ld, int T, int D, OwnRows& R) {
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = c < (D >> 2);
#pragma unroll
    for (int i = 0; i < HEAD_CHUNK / 2; ++i) {
        const int t = rg + 16 * i;
        R.xh[i] = (t < T && on) ? *reinterpret_cast<const float4*>(xl + t * ld + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        R.rstd[i] = 0.f;
    }
}

#pragma clang fp contract(fast)

// host side (head_fused.hip): the arguments of amid_head_fwd_bwd_own_vec_f32 without x and without transposes, checked -- for the forward
// launch that carries the head on its tail (sasrec_seq.hip amid_sas_seq_fwd_split_lnstat_head_f32)
int head_own_vec_args(HeadArgs& a, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1, const float* b1,
                      const float* w2, const float* b2, const float* labels, const long long* domain_id, int B, int T, int NI, int D, int hid,
                      float eps, float* u, float* p1, float* p2, float* dp1, float* dp2, float* loss_part, float* dx, float* ditems,
                      float* ln_part, float* hidg);

}  // namespace amid
