// Fused head of the SASRec step, one workgroup per batch row b (both domains):
//   head_fwd : u_g[b] = mean_t LN_last(x[g,b,t,:])  (model_seq.py:385, :432-434)  ->  scorer (predictModule.forward,
//              model_seq.py:40-54)  ->  masked BCE partial + dLoss/dp (train_sr.py:203-212)
//   head_bwd : scorer backward  ->  d u_g[b]  ->  LN_last' + mean' for the T rows of (g, b)
// One launch each instead of lnmean + scorer (+ sum) launches; every short kernel of the step costs ~5 us of
// timeline whatever it does.  The first scorer weight W1 [hid, 2D] is staged TRANSPOSED in LDS (row stride hid+1:
// conflict-free both for "lane = hidden unit" and for "lane = feature" accesses), so no dot product waits on a
// global load.  Extra workgroups (blockIdx >= B) of head_bwd refresh the transposed copies of the encoder's
// projection weights for the backward GEMMs.
#include "common.h"

namespace amid {

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
    for (int blk = hw; blk < (hid >> 2) * qb; blk += n_hw) {
        const int jb = blk / qb, cb = blk - jb * qb;
        const int j = 4 * jb + (l >> 3), c = 8 * cb + (l & 7);
        const float4 v = ld4(w1 + (long long)j * D2 + 4 * c);
        float* o = w1t + (4 * c) * (hid + 1) + j;
        o[0] = v.x; o[hid + 1] = v.y; o[2 * (hid + 1)] = v.z; o[3 * (hid + 1)] = v.w;
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
                        mean = group_sum<32>(f4hsum(y)) / D;
                        float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        rstd = 1.0f / sqrtf(group_sum<32>(f4hsum(f4mul(d4, d4))) / D + a.eps);
                        y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
                    }
                    acc = f4add(acc, y);
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
                        const float mean = group_sum<32>(f4hsum(y)) / D;
                        float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(f4hsum(f4mul(d4, d4))) / D + a.eps);
                        y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
                    }
                    acc = f4add(acc, y);
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
__device__ __forceinline__ void own_rows_lnmean(const HeadArgs& a, int b, OwnRows& R, float* __restrict__ red, float* __restrict__ u_all) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int own = a.domain[b] != 0 ? 1 : 0;
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
                const float mean = group_sum<32>(f4hsum(y)) / D;
                float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float rstd = 1.0f / sqrtf(group_sum<32>(f4hsum(f4mul(d4, d4))) / D + a.eps);
                R.rstd[i] = rstd;
                R.xh[i] = f4scale(d4, rstd);
                y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
            }
            acc = f4add(acc, y);
        }
    }
    if (on) st4(red + rg * D + 4 * c, acc);
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

// red [16][2][D]
__device__ __forceinline__ void own_rows_ln_bwd(const HeadArgs& a, int b, const OwnRows& R, const float* __restrict__ du_all, float* __restrict__ red) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int own = a.domain[b] != 0 ? 1 : 0;
    const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = c < q;
    const float* du_s = du_all + own * D;
    const bool use_ln = a.lnw[0] != nullptr;
    const float invT = 1.0f / T;
    const long long base = ((long long)own * a.B + b) * T * D;
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    const float4 dy = on ? f4scale(ld4(du_s + 4 * c), invT) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gy = dy;
    if (use_ln && on) gy = f4mul(dy, ld4(a.lnw[own] + 4 * c));
    const float c1 = use_ln ? group_sum<32>(f4hsum(gy)) / D : 0.f;
#pragma unroll
    for (int i = 0; i < HEAD_CHUNK / 2; ++i) {
        const int t = rg + 16 * i;
        if (t < T) {
            float4 out = dy;
            if (use_ln) {
                const float4 xh = R.xh[i];
                const float rstd = R.rstd[i];
                const float c2 = group_sum<32>(f4hsum(f4mul(gy, xh))) / D;
                out = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                  rstd * (gy.w - c1 - xh.w * c2));
                dgam = f4add(dgam, f4mul(dy, xh));
                dbet = f4add(dbet, dy);
            }
            if (on) st4(a.dx + base + (long long)t * D + 4 * c, out);
        }
    }
    if (on) {
        st4(red + rg * 2 * D + 4 * c, dgam);
        st4(red + rg * 2 * D + D + 4 * c, dbet);
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
}

// LDS carve (floats): w1t [2D][hid+1] | u_s [2][D] | au [2][hid] | da [2][hid] | dw2 [hid+4] | ci [64][hid+1] | dc [64][hid+1] | scratch [16][D]
// `chunk` = items whose pre-activations are resident at once (64 for the one-head kernels; the three-head kernel keeps three
// carves in LDS and uses 16)
struct HeadLds {
    float *w1t, *u_s, *au, *da, *dw2, *ci, *dc, *scr;
    int chunk;
    __device__ HeadLds(float* base, int D, int hid, int chunk_ = 64) {
        chunk = chunk_;
        w1t = base; u_s = w1t + 2 * D * (hid + 1); au = u_s + 2 * D; da = au + 2 * hid; dw2 = da + 2 * hid;
        ci = dw2 + hid + 4; dc = ci + chunk * (hid + 1); scr = dc + chunk * (hid + 1);
        scr = base + ((scr - base + 3) & ~3);
    }
};
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
            for (int e = part; e < D; e += 8) acc = fmaf(s.w1t[e * (hid + 1) + j], ur[e], acc);
        }
        acc = group_sum<8>(acc);
        if (on && part == 0) s.au[dj] = acc + a.b1[j];
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
            const float* ir = a.items + ((long long)b * a.NI + n0 + n) * D;
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
        __syncthreads();
        item_half(a, s, b, n0, nn, tg);
        __syncthreads();
        HEAD_STAMP(4);
        // 32 lanes per logit, one hidden unit each (a thread per logit walked `hid` dependent loads and fmas)
        for (int nd0 = 0; nd0 < nn * 2; nd0 += tg.n >> 5) {
            const int nd = nd0 + (tg.tid >> 5), j0 = tg.tid & 31;
            const bool on = nd < nn * 2;
            const int n = on ? nd >> 1 : 0, d = nd & 1;
            float zp = 0.f;
            if (on) for (int j = j0; j < hid; j += 32) zp = fmaf(a.w2[j], fmaxf(s.au[d * hid + j] + s.ci[n * (hid + 1) + j], 0.f), zp);
            const float z = group_sum<32>(zp) + a.b2[0];
            if (!on || j0 != 0) continue;
            const float p = 1.0f / (1.0f + expf(-z));
            const long long o = (long long)b * NI + n0 + n;
            (d ? a.p2 : a.p1)[o] = p;
            if (a.labels) {
                const float y = a.labels[o];
                const float md = a.domain[b] ? (d ? 1.f : 0.f) : (d ? 0.f : 1.f);
                const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.0f - p), -100.f);
                const float inv = 1.0f / ((float)a.B * (float)NI);
                lsum += -(y * lp + (1.f - y) * l1p) * md * inv;
                (d ? a.dp2 : a.dp1)[o] = md * inv * (p - y) / fmaxf((1.f - p) * p, 1e-12f);   // torch binary_cross_entropy_backward
            }
        }
    }
    HEAD_STAMP(5);
    if (a.labels) {
        __syncthreads();
        lsum = group_sum<64>(lsum);
        if (lane_id() == 0) s.scr[wave_id()] = lsum;
        __syncthreads();
        if (tg.tid == 0) a.loss_part[b] = ((s.scr[0] + s.scr[1]) + (s.scr[2] + s.scr[3])) + ((s.scr[4] + s.scr[5]) + (s.scr[6] + s.scr[7]));
    }
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
        const float4 dy = on ? f4scale(ld4(du_s + 4 * c), invT) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gy = dy;
        if (use_ln && on) gy = f4mul(dy, ld4(w + 4 * c));
        const float c1 = use_ln ? group_sum<32>(f4hsum(gy)) / D : 0.f;      // same for every row: dy does not depend on t
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
                        const float mean = group_sum<32>(f4hsum(v[i])) / D;
                        float4 d4 = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(f4hsum(f4mul(d4, d4))) / D + a.eps);
                        const float4 xh = f4scale(d4, rstd);
                        const float c2 = group_sum<32>(f4hsum(f4mul(gy, xh))) / D;
                        out = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                          rstd * (gy.w - c1 - xh.w * c2));
                        dgam = f4add(dgam, f4mul(dy, xh));
                        dbet = f4add(dbet, dy);
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
        const float4 dy = on ? f4scale(ld4(du_s + 4 * c), invT) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 gy = dy;
        if (use_ln && on) gy = f4mul(dy, ld4(a.lnw[own] + 4 * c));
        const float c1 = use_ln ? group_sum<32>(f4hsum(gy)) / D : 0.f;
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
                        const float mean = group_sum<32>(f4hsum(v[i])) / D;
                        float4 d4 = make_float4(v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(f4hsum(f4mul(d4, d4))) / D + a.eps);
                        const float4 xh = f4scale(d4, rstd);
                        const float c2 = group_sum<32>(f4hsum(f4mul(gy, xh))) / D;
                        out = make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                          rstd * (gy.w - c1 - xh.w * c2));
                        dgam = f4add(dgam, f4mul(dy, xh));
                        dbet = f4add(dbet, dy);
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
    __syncthreads();
    if (!FUSED) user_half(a, s, tg);
    // item half of dW1 accumulates over item chunks in registers: thread owns (j, e) pairs je = tid + 256 k
    const int CH = s.chunk;
    for (int n0 = 0; n0 < NI; n0 += CH) {
        const int nn = min(CH, NI - n0);
        __syncthreads();
        if (!(FUSED && NI <= CH)) item_half(a, s, b, n0, nn, tg);
        __syncthreads();
        if (tg.tid < hid) {                       // hidden unit j walks the chunk's items in order
            const int j = tg.tid;
            float s_da0 = 0.f, s_da1 = 0.f, s_w2 = 0.f;
            const float w2j = a.w2[j];
            for (int n = 0; n < nn; ++n) {
                const long long o = (long long)b * NI + n0 + n;
                const float p1 = a.p1[o], p2 = a.p2[o];
                const float dz1 = a.dp1[o] * p1 * (1.f - p1), dz2 = a.dp2[o] * p2 * (1.f - p2);
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
                const float p1 = a.p1[o], p2 = a.p2[o];
                acc += a.dp1[o] * p1 * (1.f - p1) + a.dp2[o] * p2 * (1.f - p2);
            }
            s.dw2[hid] += acc;
        }
        __syncthreads();
        HEAD_STAMP(8);
        // d item[n][e] = sum_j dc[n][j] w1t[D+e][j]
        for (int ne = tg.tid; ne < nn * D; ne += tg.n) {
            const int n = ne / D, e = ne - n * D;
            float acc = 0.f;
            const float* wp = s.w1t + (D + e) * (hid + 1);
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
    __syncthreads();
    HEAD_STAMP(9);
    // user halves: du_d[e] = sum_j da[d][j] w1t[e][j]  -> scratch region [2][D] reused from ci (dead now)
    float* du_s = s.ci;
    for (int de = tg.tid; de < 2 * D; de += tg.n) {
        const int d = de / D, e = de - d * D;
        float acc = 0.f;
        const float* wp = s.w1t + e * (hid + 1);
        for (int j = 0; j < hid; ++j) acc = fmaf(s.da[d * hid + j], wp[j], acc);
        du_s[de] = acc;
    }
    if (a.hidg != nullptr) {
        float* hg = a.hidg + (long long)b * (((3 + NI) * hid + 1 + 3) & ~3);
        for (int j = tg.tid; j < 2 * hid; j += tg.n) hg[j] = s.da[j];
        for (int j = tg.tid; j < hid + 1; j += tg.n) hg[(2 + NI) * hid + j] = s.dw2[j];
        __syncthreads();
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
    __syncthreads();
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

__global__ __launch_bounds__(512) void head_fwd_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    head_fwd_body(a, sm, blockIdx.x);
}

__global__ __launch_bounds__(512) void head_bwd_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    head_bwd_body<false>(a, sm);
}

// the training step's head in ONE launch: forward (loss partials, dLoss/dp) then, in the same workgroup with its LDS state
// intact, backward; the extra workgroups (blockIdx >= B) only transpose the projection weights
__global__ __launch_bounds__(512) void head_fwd_bwd_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if ((int)blockIdx.x < a.B && a.own_only && a.T <= 16 * (HEAD_CHUNK / 2)) {
        // the live-sequence train step: the sample's own sequence, loaded once and kept in registers from the forward LayerNorm to its backward
        const HeadLds s(sm, a.D, a.hid);
        const int b = blockIdx.x;
        HEAD_STAMP(0);
        OwnRows R;
        own_rows_load(a, b, R);
        stage_w1t(s.w1t, a.w1, 2 * a.D, a.hid, whole_block());
        HEAD_STAMP(1);
        own_rows_lnmean(a, b, R, s.scr, s.u_s);
        HEAD_STAMP(2);
        scorer_fwd_part(a, s, b, whole_block());
        HEAD_STAMP(6);
        __threadfence_block();                            // dLoss/dp written above is read by other threads of this workgroup below
        __syncthreads();
        HEAD_STAMP(7);
        float* du_s = scorer_bwd_part<true, false>(a, s, b, whole_block());
        HEAD_STAMP(10);
        own_rows_ln_bwd(a, b, R, du_s, s.scr);
        HEAD_STAMP(11);
        return;
    }
    if ((int)blockIdx.x < a.B) {
        head_fwd_body(a, sm, blockIdx.x);
        __threadfence_block();                            // dLoss/dp written above is read by other threads of this workgroup below
        __syncthreads();
    }
    head_bwd_body<true>(a, sm);
}

// scorer forward alone on given user vectors u [2, B, D] (evaluation of the isItC / isDR models, up to 1 000 candidates per row):
// W1^T staged in LDS, items in chunks of 64 -- the same code path as the fused head's forward half
__global__ __launch_bounds__(512) void scorer_fwd_only_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const HeadLds s(sm, a.D, a.hid);
    const int b = blockIdx.x, D = a.D;
    stage_w1t(s.w1t, a.w1, 2 * D, a.hid, whole_block());
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) s.u_s[e] = a.u[((long long)(e / D) * a.B + b) * D + (e % D)];
    __syncthreads();
    scorer_fwd_part(a, s, b, whole_block());
}

// ---------------------------------------------------------------------------------------------
// The head of the isItC / isDR train step after the user vectors are mixed (csrc/intercomp.hip): up to three scorers
// (predictModule, predict_ips, predict_gfunc; model_seq.py:436-440) forward on the same (u, items) of row b, the row's loss
// terms -- masked BCE (one head) or the doubly-robust objective of the current mode (three heads; train_sr_dr.py:216-221,
// :392-394, same arithmetic as dr_loss_kernel in head.hip) --, then every scorer's backward: d items and d u summed over the
// heads, weight-gradient partials per head.  One launch instead of 3 + 1 + 3 (each 20-40 us of mostly latency).
// Every head keeps its own LDS carve (W1^T, hidden pre-activations of the user vectors and of the items) from forward to backward,
// and the three heads run SIDE BY SIDE on three 256-thread groups of the workgroup (run one after the other they cost 89 us).
// ---------------------------------------------------------------------------------------------
struct MultiHeadArgs {
    HeadArgs h[3];             // per head: w1..b2, p1/p2 (outputs), dp1/dp2 (their gradients), sc_part; shared x-less fields
    int n_heads;               // 1: masked BCE on head 0 (h[0].labels set) ; 3: doubly-robust objective
    const float* labels; const long long* domain; const long long* ob;
    int mode; float w;
    float* dr_loss_part;       // [B][3]
    float* du;                 // [2, B, D]
};

__global__ __launch_bounds__(768) void scorer_multi_fwd_bwd_kernel(const MultiHeadArgs m) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const HeadArgs& a0 = m.h[0];
    if ((int)blockIdx.x >= a0.B) { transpose_extra(a0, sm); return; }
    const int b = blockIdx.x, D = a0.D, NI = a0.NI, nh = m.n_heads;
    // three heads: the workgroup's 768 threads are three groups of 256, one head each, side by side (same control flow, so the
    // barriers inside the scorer parts line up); one head: the whole workgroup
    const int gsz = nh == 1 ? (int)blockDim.x : 256;
    const int h = __builtin_amdgcn_readfirstlane((int)threadIdx.x / gsz);
    const Tg tg{(int)threadIdx.x - h * gsz, gsz};
    const int CH = nh == 1 ? 64 : 16;
    const size_t carve = head_carve_floats(D, a0.hid, CH);
    HeadLds s(sm + h * carve, D, a0.hid, CH);
    float* dit_all = sm + nh * carve;                         // three heads: [3][NI <= 16][D] shares of d items
    float* scratch = dit_all + (nh == 1 ? 0 : nh * CH * D);   // [8 D]
    s.scr = scratch;
    const HeadArgs& a = m.h[h];
    stage_w1t(s.w1t, a.w1, 2 * D, a0.hid, tg);
    for (int e = tg.tid; e < 2 * D; e += tg.n) s.u_s[e] = a0.u[((long long)(e / D) * a0.B + b) * D + (e % D)];
    __syncthreads();
    scorer_fwd_part(a, s, b, tg);
    if (nh == 3) {
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x < 64) {                               // wave 0: the row's doubly-robust terms (dr_loss_kernel, head.hip)
            const int lane = threadIdx.x;
            const int d = m.domain[b] != 0;
            const float ob = (m.ob != nullptr) ? (float)m.ob[b] : 0.f;
            const float inv = 1.0f / (float)((long long)a0.B * NI);
            float lc = 0.f, le = 0.f, lr = 0.f;
            for (int n = lane; n < NI; n += 64) {
                const long long o = (long long)b * NI + n;
                const float p = (d ? m.h[0].p2 : m.h[0].p1)[o], ips = (d ? m.h[1].p2 : m.h[1].p1)[o], g = (d ? m.h[2].p2 : m.h[2].p1)[o];
                const float y = m.labels[o];
                const float bce = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
                const float dbce = (p - y) / fmaxf(p * (1.f - p), 1e-12f);
                float dp, dg, dips;
                lc += bce;
                le += (bce - g) * (bce - g) / ips;
                const float q = bce * bce - g * g;
                lr += g * g + ob * q * q / ips;
                if (m.mode == 0) {
                    dp = (1.f + 2.f * m.w * (bce - g) / ips) * dbce;
                    dg = -2.f * m.w * (bce - g) / ips;
                    dips = -m.w * (bce - g) * (bce - g) / (ips * ips);
                } else {
                    dp = ob * 4.f * q * bce / ips * dbce;
                    dg = 2.f * g - ob * 4.f * q * g / ips;
                    dips = -ob * q * q / (ips * ips);
                }
                (d ? m.h[0].dp2 : m.h[0].dp1)[o] = dp * inv; (d ? m.h[0].dp1 : m.h[0].dp2)[o] = 0.f;
                (d ? m.h[1].dp2 : m.h[1].dp1)[o] = dips * inv; (d ? m.h[1].dp1 : m.h[1].dp2)[o] = 0.f;
                (d ? m.h[2].dp2 : m.h[2].dp1)[o] = dg * inv; (d ? m.h[2].dp1 : m.h[2].dp2)[o] = 0.f;
            }
            lc = group_sum<64>(lc); le = group_sum<64>(le); lr = group_sum<64>(lr);
            if (lane == 0) { m.dr_loss_part[b * 3 + 0] = lc * inv; m.dr_loss_part[b * 3 + 1] = le * inv; m.dr_loss_part[b * 3 + 2] = lr * inv; }
        }
    }
    __threadfence_block();                                    // dLoss/dp written above is read by other threads of this workgroup below
    __syncthreads();
    scorer_bwd_part<true, false>(a, s, b, tg, nh == 1 ? nullptr : dit_all + h * CH * D);      // ends with a barrier
    // d u (left by every head at the start of its ci region) and, with three heads, d items: fixed-order sums over the heads
    const size_t ci_off = (size_t)(s.ci - (sm + h * carve));
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
        float t = sm[ci_off + e];
        for (int k = 1; k < nh; ++k) t += sm[k * carve + ci_off + e];
        m.du[((long long)(e / D) * a0.B + b) * D + (e % D)] = t;
    }
    if (nh > 1)
        for (int i = threadIdx.x; i < NI * D; i += blockDim.x) {
            float t = dit_all[i];
            for (int k = 1; k < nh; ++k) t += dit_all[k * CH * D + i];
            a0.ditems[(long long)b * NI * D + i] = t;
        }
}

}  // namespace amid

using namespace amid;

#ifdef AMID_HEAD_STAMPS
extern "C" int amid_head_stamps_read(unsigned long long* host) {       // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_head_stamp_buf), sizeof(unsigned long long) * 32);
}
#endif

static int head_fill(HeadArgs& a, const float* x, const float* const* lnw, const float* const* lnb, const float* items, const float* w1,
                     const float* b1, const float* w2, const float* b2, int B, int T, int NI, int D, int hid, float eps) {
    if (!(x && items && w1 && b1 && w2 && b2) || B <= 0 || T <= 0 || NI <= 0 || D <= 0 || (D % 32) != 0 || D > 128 || hid <= 0 || hid > 64 || (hid % 4) != 0)
        return AMID_ERR_ARG;
    a.x = x; a.items = items; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    for (int g = 0; g < 2; ++g) { a.lnw[g] = lnw ? lnw[g] : nullptr; a.lnb[g] = lnb ? lnb[g] : nullptr; }
    a.B = B; a.T = T; a.NI = NI; a.D = D; a.hid = hid; a.eps = eps; a.n_tr = 0;
    return AMID_OK;
}

static int head_lds_attr(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return AMID_OK;
}

// ln_w / ln_b: host arrays of 2 device pointers (both null arrays: no LayerNorm, plain mean over T)
extern "C" int amid_head_fwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                 const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id, int B,
                                 int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                 float* loss_part, void* stream) {
    HeadArgs a = {};
    if (int e = head_fill(a, x, ln_w, ln_b, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    AMID_CHECK_ARG(u && p1 && p2 && (!labels || (domain_id && dp1 && dp2 && loss_part)));
    a.labels = labels; a.domain = domain_id; a.u = u; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)head_fwd_kernel, lds)) return e;
    head_fwd_kernel<<<B, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// tr_src / tr_dst: optional host arrays of n_tr (<= 32) device pointers of square D x D matrices transposed by extra workgroups
extern "C" int amid_head_bwd_f32(const float* x, const float* const* ln_w, const float* u, const float* items, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* p1, const float* p2, const float* dp1, const float* dp2, int B,
                                 int T, int NI, int D, int hid, float eps, float* dx, float* ditems, float* ln_part, float* sc_part,
                                 const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    HeadArgs a = {};
    if (int e = head_fill(a, x, ln_w, nullptr, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    AMID_CHECK_ARG(u && p1 && p2 && dp1 && dp2 && dx && ditems && sc_part && (!ln_w || ln_part) && n_tr >= 0 && n_tr <= 32);
    a.u = const_cast<float*>(u); a.p1 = const_cast<float*>(p1); a.p2 = const_cast<float*>(p2);
    a.dp1 = const_cast<float*>(dp1); a.dp2 = const_cast<float*>(dp2);
    a.dx = dx; a.ditems = ditems; a.ln_part = ln_part; a.sc_part = sc_part; a.n_tr = n_tr;
    for (int i = 0; i < n_tr; ++i) { AMID_CHECK_ARG(tr_src && tr_dst && tr_src[i] && tr_dst[i]); a.tr_src[i] = tr_src[i]; a.tr_dst[i] = tr_dst[i]; }
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)head_bwd_kernel, lds)) return e;
    const int extra = n_tr > 0 ? min(n_tr * (D / 32) * (D / 32), 96) : 0;
    head_bwd_kernel<<<B + extra, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// amid_head_fwd_f32 (with labels) immediately followed by amid_head_bwd_f32, ONE launch -- the training step's head
static int head_fwd_bwd(int own_only, const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                     int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                     float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part,
                                     const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream, float* hidg = nullptr) {
    HeadArgs a = {};
    if (int e = head_fill(a, x, ln_w, ln_b, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    a.own_only = own_only;
    a.hidg = hidg;
    AMID_CHECK_ARG(labels && domain_id && u && p1 && p2 && dp1 && dp2 && loss_part && dx && ditems && (sc_part || hidg) && (!ln_w || ln_part) &&
                   n_tr >= 0 && n_tr <= 32);
    AMID_CHECK_ARG(hidg == nullptr || NI <= 64);
    a.labels = labels; a.domain = domain_id; a.u = u; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    a.dx = dx; a.ditems = ditems; a.ln_part = ln_part; a.sc_part = sc_part; a.n_tr = n_tr;
    for (int i = 0; i < n_tr; ++i) { AMID_CHECK_ARG(tr_src && tr_dst && tr_src[i] && tr_dst[i]); a.tr_src[i] = tr_src[i]; a.tr_dst[i] = tr_dst[i]; }
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)head_fwd_bwd_kernel, lds)) return e;
    const int extra = n_tr > 0 ? min(n_tr * (D / 32) * (D / 32), 96) : 0;
    head_fwd_bwd_kernel<<<B + extra, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_head_fwd_bwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                     int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                     float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part,
                                     const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    return head_fwd_bwd(0, x, ln_w, ln_b, items, w1, b1, w2, b2, labels, domain_id, B, T, NI, D, hid, eps, u, p1, p2, dp1, dp2, loss_part, dx,
                        ditems, ln_part, sc_part, tr_src, tr_dst, n_tr, stream);
}

// the same when only the sequence (domain_id[b], b) of every sample was encoded (the train step over the live sequences): the other
// domain's user vector reads as zero (its logits are NOT the model's: the masked loss never reads them, train_sr.py:205-211), it
// receives no gradient, and of x / dx only the own sequences' rows are read / written
extern "C" int amid_head_fwd_bwd_own_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                         const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                         int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1,
                                         float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part,
                                         const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    return head_fwd_bwd(1, x, ln_w, ln_b, items, w1, b1, w2, b2, labels, domain_id, B, T, NI, D, hid, eps, u, p1, p2, dp1, dp2, loss_part, dx,
                        ditems, ln_part, sc_part, tr_src, tr_dst, n_tr, stream);
}

// amid_head_fwd_bwd_own_f32 that hands the scorer's weight gradients on as per-sample HIDDEN gradients (hidg [B][amid_scorer_vec_floats(NI, hid)]:
// da [2][hid] | dc [NI][hid] | dW2's [hid] | db2's) instead of per-sample partials of every weight (sc_part: 32 KB a sample, 8.5 MB a step
// written here and read back by the gradient tail): amid_grad_tail_live_f32 forms dW1 = sum_b da_b (x) u_b + dc_b (x) items_b itself.  NI <= 64.
extern "C" long long amid_scorer_vec_floats(int NI, int hid) { return ((long long)(3 + NI) * hid + 1 + 3) & ~3LL; }
extern "C" int amid_head_fwd_bwd_own_vec_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                             const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                             int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1,
                                             float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg,
                                             const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    AMID_CHECK_ARG(hidg != nullptr);
    return head_fwd_bwd(1, x, ln_w, ln_b, items, w1, b1, w2, b2, labels, domain_id, B, T, NI, D, hid, eps, u, p1, p2, dp1, dp2, loss_part, dx,
                        ditems, ln_part, nullptr, tr_src, tr_dst, n_tr, stream, hidg);
}

// Up to three scorers forward + loss + backward in ONE launch on given user vectors u [2, B, D] (the isItC / isDR train step, see
// scorer_multi_fwd_bwd_kernel).  Host arrays of n_heads device pointers: w1, b1, w2, b2, p1, p2 (outputs), dp1, dp2 (gradients of
// the outputs, written here), sc_part (weight-gradient partials).  n_heads == 1: masked BCE of train_sr.py:203-212 (loss_part [B]);
// n_heads == 3 (predictModule, predict_ips, predict_gfunc): the doubly-robust objective `mode` of amid_dr_loss_f32
// (dr_loss_part [B][3]; ob_label needed for mode 1).  du [2, B, D] and ditems [B, NI, D] are sums over the heads.
extern "C" int amid_scorer_multi_fwd_bwd_f32(const float* u, const float* items, const float* const* w1, const float* const* b1,
                                             const float* const* w2, const float* const* b2, int n_heads, const float* labels,
                                             const long long* domain_id, const long long* ob_label, int mode, float dr_e_w, int B, int NI,
                                             int D, int hid, float* const* p1, float* const* p2, float* const* dp1, float* const* dp2,
                                             float* loss_part, float* dr_loss_part, float* du, float* ditems, float* const* sc_part,
                                             const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    AMID_CHECK_ARG(u && items && w1 && b1 && w2 && b2 && labels && domain_id && p1 && p2 && dp1 && dp2 && du && ditems && sc_part);
    AMID_CHECK_ARG((n_heads == 1 && loss_part) || (n_heads == 3 && dr_loss_part && (mode == 0 || (mode == 1 && ob_label))));
    AMID_CHECK_ARG(n_tr >= 0 && n_tr <= 32);
    MultiHeadArgs m = {};
    for (int h = 0; h < n_heads; ++h) {
        HeadArgs& a = m.h[h];
        AMID_CHECK_ARG(p1[h] && p2[h] && dp1[h] && dp2[h] && sc_part[h]);
        if (int e = head_fill(a, u, nullptr, nullptr, items, w1[h], b1[h], w2[h], b2[h], B, 1, NI, D, hid, 0.f)) return e;
        a.u = const_cast<float*>(u); a.p1 = p1[h]; a.p2 = p2[h]; a.dp1 = dp1[h]; a.dp2 = dp2[h]; a.sc_part = sc_part[h]; a.ditems = ditems;
        if (n_heads == 1) { a.labels = labels; a.domain = domain_id; a.loss_part = loss_part; }
    }
    m.h[0].n_tr = n_tr;
    for (int i = 0; i < n_tr; ++i) { AMID_CHECK_ARG(tr_src && tr_dst && tr_src[i] && tr_dst[i]); m.h[0].tr_src[i] = tr_src[i]; m.h[0].tr_dst[i] = tr_dst[i]; }
    m.n_heads = n_heads; m.labels = labels; m.domain = domain_id; m.ob = ob_label; m.mode = mode; m.w = dr_e_w;
    m.dr_loss_part = dr_loss_part; m.du = du;
    if (n_heads == 3 && NI > 16) return AMID_ERR_UNSUPPORTED;          // the three heads' d item shares are summed through LDS
    const size_t lds = ((size_t)n_heads * head_carve_floats(D, hid, n_heads == 1 ? 64 : 16) + (n_heads == 1 ? 0 : (size_t)n_heads * 16 * D) +
                        8 * (size_t)D) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)scorer_multi_fwd_bwd_kernel, lds)) return e;
    const int extra = n_tr > 0 ? min(n_tr * (D / 32) * (D / 32), 96) : 0;
    scorer_multi_fwd_bwd_kernel<<<B + extra, n_heads == 1 ? 512 : 768, lds, (hipStream_t)stream>>>(m);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// replaces: predictModule.forward model_seq.py:40-54 on given user vectors; with labels != NULL also the masked BCE mean of
// train_sr.py:203-212 (per-row loss partials + dLoss/dp)
extern "C" int amid_scorer_fwd_f32(const float* u, const float* items, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* labels, const long long* domain_id, int B, int NI, int D, int hid, float* p1, float* p2,
                                   float* dp1, float* dp2, float* loss_part, void* stream) {
    AMID_CHECK_ARG(u && p1 && p2 && (!labels || (domain_id && dp1 && dp2 && loss_part)));
    HeadArgs a = {};
    if (int e = head_fill(a, u, nullptr, nullptr, items, w1, b1, w2, b2, B, 1, NI, D, hid, 0.f)) return e;
    a.u = const_cast<float*>(u); a.labels = labels; a.domain = domain_id; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)scorer_fwd_only_kernel, lds)) return e;
    scorer_fwd_only_kernel<<<B, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
