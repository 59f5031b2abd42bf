// Everything after the encoder stack:
//   lnmean  : u[g,b,:] = mean_t LN_last(x[g,b,t,:])          (model_seq.py:385, :432-434: mean over ALL T, pads included)
//   scorer  : p_d[b,n] = sigmoid(W2 relu(W1 [u_d[b] ; item[b,n]] + b1) + b2), d = 1,2   (predictModule.forward, model_seq.py:40-54)
//   loss    : mean over B*(1+neg) of BCE(p_1)*(1-domain) + BCE(p_2)*domain, log clamped at -100
//             (nn.BCELoss(reduce=False), train_sr.py:184, :203-212) and its gradient wrt p.
// and the matching backward kernels.  All of it is tiny next to the encoder (B*(1+neg) rows).
#include "common.h"

namespace amid {

// ---- final LayerNorm + mean over time ---------------------------------------------------------
// block = (g, b); 256 threads = 8 row groups of 32 lanes
__global__ __launch_bounds__(256) void lnmean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w0, const float* __restrict__ b0,
                                                         const float* __restrict__ w1, const float* __restrict__ b1, int B, int T, int D,
                                                         float eps, int use_ln, float* __restrict__ u) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [8][D]
    const int seq = blockIdx.x, g = seq / B;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int q = D >> 2;
    const float* w = g ? w1 : w0;
    const float* bb = g ? b1 : b0;
    for (int c = sub; c < q; c += 32) st4(red + rg * D + 4 * c, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int t = rg; t < T; t += 8) {
        const float* row = x + ((long long)seq * T + t) * D;
        float s = 0.f;
        for (int c = sub; c < q; c += 32) s += f4hsum(ld4(row + 4 * c));
        const float mean = use_ln ? group_sum<32>(s) / D : 0.f;
        float vs = 0.f;
        if (use_ln) for (int c = sub; c < q; c += 32) { float4 v = ld4(row + 4 * c); v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean; vs += f4hsum(f4mul(v, v)); }
        const float rstd = use_ln ? 1.0f / sqrtf(group_sum<32>(vs) / D + eps) : 1.f;
        for (int c = sub; c < q; c += 32) {
            float4 v = ld4(row + 4 * c);
            if (use_ln) {
                const float4 ww = ld4(w + 4 * c), b4 = ld4(bb + 4 * c);
                v.x = (v.x - mean) * rstd * ww.x + b4.x; v.y = (v.y - mean) * rstd * ww.y + b4.y;
                v.z = (v.z - mean) * rstd * ww.z + b4.z; v.w = (v.w - mean) * rstd * ww.w + b4.w;
            }
            float* rp = red + rg * D + 4 * c;
            st4(rp, f4add(ld4(rp), v));
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D; e += 256) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k * D + e];
        u[(long long)seq * D + e] = s / T;
    }
}

// dx[g,b,t,:] = LN_last'(du[g,b,:] / T ; x[g,b,t,:]) ; per-block partials of d gamma / d beta -> part[seq][2][D]
__global__ __launch_bounds__(256) void lnmean_bwd_kernel(const float* __restrict__ x, const float* __restrict__ du, const float* __restrict__ w0,
                                                         const float* __restrict__ w1, int B, int T, int D, float eps, int use_ln,
                                                         float* __restrict__ dx, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [8][2][D]
    const int seq = blockIdx.x, g = seq / B;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int q = D >> 2;
    const float* w = g ? w1 : w0;
    const float invT = 1.0f / T;
    for (int c = sub; c < 2 * q; c += 32) st4(red + rg * 2 * D + 4 * c, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int t = rg; t < T; t += 8) {
        const long long ro = ((long long)seq * T + t) * D;
        if (!use_ln) {
            for (int c = sub; c < q; c += 32) st4(dx + ro + 4 * c, f4scale(ld4(du + (long long)seq * D + 4 * c), invT));
            continue;
        }
        float s = 0.f;
        for (int c = sub; c < q; c += 32) s += f4hsum(ld4(x + ro + 4 * c));
        const float mean = group_sum<32>(s) / D;
        float vs = 0.f;
        for (int c = sub; c < q; c += 32) { float4 v = ld4(x + ro + 4 * c); v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean; vs += f4hsum(f4mul(v, v)); }
        const float rstd = 1.0f / sqrtf(group_sum<32>(vs) / D + eps);
        float a1 = 0.f, a2 = 0.f;
        for (int c = sub; c < q; c += 32) {
            const float4 v = ld4(x + ro + 4 * c), dy = f4scale(ld4(du + (long long)seq * D + 4 * c), invT), gy = f4mul(dy, ld4(w + 4 * c));
            const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
            a1 += f4hsum(gy);
            a2 += f4hsum(f4mul(gy, xh));
        }
        const float c1 = group_sum<32>(a1) / D, c2 = group_sum<32>(a2) / D;
        for (int c = sub; c < q; c += 32) {
            const float4 v = ld4(x + ro + 4 * c), dy = f4scale(ld4(du + (long long)seq * D + 4 * c), invT), gy = f4mul(dy, ld4(w + 4 * c));
            const float4 xh = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
            st4(dx + ro + 4 * c, make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                                             rstd * (gy.w - c1 - xh.w * c2)));
            float* rp = red + rg * 2 * D + 4 * c;
            st4(rp, f4add(ld4(rp), f4mul(dy, xh)));
            st4(rp + D, f4add(ld4(rp + D), dy));
        }
    }
    __syncthreads();
    if (part) {
        for (int e = threadIdx.x; e < 2 * D; e += 256) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k * 2 * D + e];
            part[(long long)seq * 2 * D + e] = s;
        }
    }
}

// ---- scorer ------------------------------------------------------------------------------------
// block = batch row b; thread layout: hid units x lanes.  W1 = [hid, 2D] (user half | item half).

struct ScorerBwdArgs {
    const float* u; const float* items; const float* w1; const float* b1; const float* w2; const float* b2;
    const float* p1; const float* p2; const float* dp1; const float* dp2;
    float* du;                 // [2, B, D]
    float* ditems;             // [B, NI, D]
    float* part;               // [B][hid*2D + hid + hid + 1]  (dW1 | db1 | dW2 | db2) per-row partials
    int B, NI, D, hid;
    int accumulate;            // 1: du and ditems are added to (several heads share u and items: the doubly-robust trainer)
};

// block = batch row b.  Recomputes the hidden activations, then back-propagates.
__global__ __launch_bounds__(256) void scorer_bwd_kernel(const ScorerBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x, D = a.D, hid = a.hid, NI = a.NI;
    float* au = sm;                         // [2][hid]
    float* da = au + 2 * hid;               // [2][hid]   sum_n dpre[d][n][:]
    float* dw2 = da + 2 * hid;              // [hid] + [1]
    float* ci = dw2 + hid + 4;              // [64][hid]
    float* dc = ci + 64 * hid;              // [64][hid]  dpre[0][n] + dpre[1][n]
    const int lane = lane_id(), w = wave_id();
    const int P = (hid * 2 * D + 2 * hid + 1 + 3) & ~3;      // amid_scorer_part_floats: rows padded to whole float4s
    float* part = a.part + (long long)b * P;
    for (int dj = w; dj < 2 * hid; dj += 4) {
        const int d = dj / hid, j = dj - d * hid;
        const float* ur = a.u + ((long long)d * a.B + b) * D;
        const float* wr = a.w1 + (long long)j * 2 * D;
        float s = 0.f;
        for (int e = lane; e < D; e += 64) s = fmaf(wr[e], ur[e], s);
        s = group_sum<64>(s);
        if (lane == 0) au[d * hid + j] = s + a.b1[j];
    }
    for (int e = threadIdx.x; e < 2 * hid; e += 256) da[e] = 0.f;
    for (int e = threadIdx.x; e < hid + 1; e += 256) dw2[e] = 0.f;
    for (int e = threadIdx.x; e < hid * D; e += 256) part[(e / D) * 2 * D + D + (e % D)] = 0.f;      // item half of dW1 accumulates over chunks
    for (int n0 = 0; n0 < NI; n0 += 64) {
        const int nn = min(64, NI - n0);
        __syncthreads();
        for (int nj = w; nj < nn * hid; nj += 4) {
            const int n = nj / hid, j = nj - n * hid;
            const float* ir = a.items + ((long long)b * NI + n0 + n) * D;
            const float* wr = a.w1 + (long long)j * 2 * D + D;
            float s = 0.f;
            for (int e = lane; e < D; e += 64) s = fmaf(wr[e], ir[e], s);
            s = group_sum<64>(s);
            if (lane == 0) ci[n * hid + j] = s;
        }
        __syncthreads();
        // thread j < hid: walks the chunk's items in order (fixed summation order)
        if (threadIdx.x < hid) {
            const int j = threadIdx.x;
            float s_da0 = 0.f, s_da1 = 0.f, s_w2 = 0.f;
            for (int n = 0; n < nn; ++n) {
                const long long o = (long long)b * NI + n0 + n;
                const float p1 = a.p1[o], p2 = a.p2[o];
                const float dz1 = a.dp1[o] * p1 * (1.f - p1), dz2 = a.dp2[o] * p2 * (1.f - p2);
                const float h1 = fmaxf(au[j] + ci[n * hid + j], 0.f), h2 = fmaxf(au[hid + j] + ci[n * hid + j], 0.f);
                const float g1 = h1 > 0.f ? dz1 * a.w2[j] : 0.f, g2 = h2 > 0.f ? dz2 * a.w2[j] : 0.f;
                s_w2 += dz1 * h1 + dz2 * h2;
                s_da0 += g1; s_da1 += g2;
                dc[n * hid + j] = g1 + g2;
            }
            da[j] += s_da0; da[hid + j] += s_da1; dw2[j] += s_w2;
        }
        if (threadIdx.x == 64) {
            float s = 0.f;
            for (int n = 0; n < nn; ++n) {
                const long long o = (long long)b * NI + n0 + n;
                const float p1 = a.p1[o], p2 = a.p2[o];
                s += a.dp1[o] * p1 * (1.f - p1) + a.dp2[o] * p2 * (1.f - p2);
            }
            dw2[hid] += s;
        }
        __syncthreads();
        // d item[n][e] = sum_j dc[n][j] W1[j][D+e] ; dW1[j][D+e] += sum_n dc[n][j] item[n][e]
        for (int ne = threadIdx.x; ne < nn * D; ne += 256) {
            const int n = ne / D, e = ne - n * D;
            float s = 0.f;
            for (int j = 0; j < hid; ++j) s = fmaf(dc[n * hid + j], a.w1[(long long)j * 2 * D + D + e], s);
            float* dip = a.ditems + ((long long)b * NI + n0 + n) * D + e;
            *dip = a.accumulate ? *dip + s : s;
        }
        for (int je = threadIdx.x; je < hid * D; je += 256) {
            const int j = je / D, e = je - j * D;
            float s = 0.f;
            for (int n = 0; n < nn; ++n) s = fmaf(dc[n * hid + j], a.items[((long long)b * NI + n0 + n) * D + e], s);
            part[j * 2 * D + D + e] += s;
        }
    }
    __syncthreads();
    // user halves
    for (int de = threadIdx.x; de < 2 * D; de += 256) {
        const int d = de / D, e = de - d * D;
        float s = 0.f;
        for (int j = 0; j < hid; ++j) s = fmaf(da[d * hid + j], a.w1[(long long)j * 2 * D + e], s);
        float* dup = a.du + ((long long)d * a.B + b) * D + e;
        *dup = a.accumulate ? *dup + s : s;
    }
    for (int je = threadIdx.x; je < hid * D; je += 256) {
        const int j = je / D, e = je - j * D;
        part[j * 2 * D + e] = da[j] * a.u[(long long)b * D + e] + da[hid + j] * a.u[((long long)a.B + b) * D + e];
    }
    for (int j = threadIdx.x; j < hid; j += 256) {
        part[hid * 2 * D + j] = da[j] + da[hid + j];            // db1
        part[hid * 2 * D + hid + j] = dw2[j];                   // dW2
    }
    if (threadIdx.x == 0) part[hid * 2 * D + 2 * hid] = dw2[hid];   // db2
}

// plain row LayerNorm (biased variance, eps inside the sqrt): one half-wave per row
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                             long long rows, int D, float eps, float* __restrict__ y) {
    const int sub = threadIdx.x & 31;
    const int q = D >> 2;
    for (long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5); r < rows; r += (long long)gridDim.x * 8) {
        const float* row = x + r * D;
        float s = 0.f;
        for (int c = sub; c < q; c += 32) s += f4hsum(ld4(row + 4 * c));
        const float mean = group_sum<32>(s) / D;
        float vs = 0.f;
        for (int c = sub; c < q; c += 32) { float4 v = ld4(row + 4 * c); v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean; vs += f4hsum(f4mul(v, v)); }
        const float rstd = 1.0f / sqrtf(group_sum<32>(vs) / D + eps);
        for (int c = sub; c < q; c += 32) {
            const float4 v = ld4(row + 4 * c), ww = ld4(w + 4 * c), bb = ld4(b + 4 * c);
            st4(y + r * D + 4 * c, make_float4((v.x - mean) * rstd * ww.x + bb.x, (v.y - mean) * rstd * ww.y + bb.y,
                                               (v.z - mean) * rstd * ww.z + bb.z, (v.w - mean) * rstd * ww.w + bb.w));
        }
    }
}

__global__ void sum_vector_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    // single wave, fixed order
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += v[i];
    s = group_sum<64>(s);
    if (threadIdx.x == 0) *out = s;
}

// Doubly-robust objectives (next-4; reference train_sr_dr.py:216-221 and :392-394), elementwise over [B, NI] on the three heads'
// sigmoid outputs (p: predictModule, ips: predict_ips, g: predict_gfunc), each read from the head of the row's own domain:
//   mode 0:  L = mean( BCE(p, y) + w (BCE(p, y) - g)^2 / ips )           (loss_cls + dr_e_w * loss_dr_e; optimizer)
//   mode 1:  L = mean( g^2 + ob ((BCE(p, y))^2 - g^2)^2 / ips )          (loss_dr_r; optimizer2)
// Writes dL/dp, dL/dips, dL/dg for both domains (zero on the head a row does not use) and per-row partials of
// (loss_cls, loss_dr_e, loss_dr_r).  BCE and its derivative follow torch's BCELoss (log clamped at -100, denominator at 1e-12).
struct DrLossArgs {
    const float* p[2]; const float* ips[2]; const float* g[2];
    const float* labels; const long long* domain; const long long* ob;
    float* dp[2]; float* dips[2]; float* dg[2];
    float* loss_part;          // [B][3]
    int B, NI, mode; float w;
};
__global__ __launch_bounds__(64) void dr_loss_kernel(const DrLossArgs a) {
    const int b = blockIdx.x, lane = lane_id();
    const int d = a.domain[b] != 0;
    const float ob = (a.ob != nullptr) ? (float)a.ob[b] : 0.f;
    const float inv = 1.0f / (float)((long long)a.B * a.NI);
    float lc = 0.f, le = 0.f, lr = 0.f;
    for (int n = lane; n < a.NI; n += 64) {
        const long long o = (long long)b * a.NI + n;
        const float p = a.p[d][o], ips = a.ips[d][o], g = a.g[d][o], y = a.labels[o];
        const float bce = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
        const float dbce = (p - y) / fmaxf(p * (1.f - p), 1e-12f);
        float dp, dg, dips;
        lc += bce;
        le += (bce - g) * (bce - g) / ips;
        const float q = bce * bce - g * g;
        lr += g * g + ob * q * q / ips;
        if (a.mode == 0) {
            dp = (1.f + 2.f * a.w * (bce - g) / ips) * dbce;
            dg = -2.f * a.w * (bce - g) / ips;
            dips = -a.w * (bce - g) * (bce - g) / (ips * ips);
        } else {
            dp = ob * 4.f * q * bce / ips * dbce;
            dg = 2.f * g - ob * 4.f * q * g / ips;
            dips = -ob * q * q / (ips * ips);
        }
        a.dp[d][o] = dp * inv; a.dg[d][o] = dg * inv; a.dips[d][o] = dips * inv;
        a.dp[1 - d][o] = 0.f; a.dg[1 - d][o] = 0.f; a.dips[1 - d][o] = 0.f;
    }
    lc = group_sum<64>(lc); le = group_sum<64>(le); lr = group_sum<64>(lr);
    if (lane == 0) { a.loss_part[b * 3 + 0] = lc * inv; a.loss_part[b * 3 + 1] = le * inv; a.loss_part[b * 3 + 2] = lr * inv; }
}

// rank (0 = best) of the positive (column 0) among a row's NI scores, judged by the head of the row's own domain -- the device
// form of choose_predict + the double argsort of get_sample_scores (reference utils.py:21-40, :296-297; train_sr.py:114-115
// subtracts fix_value from the positive first, so a tie counts against it).  One wave per row.
__global__ __launch_bounds__(256) void positive_rank_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                            const long long* __restrict__ domain, int B, int NI, float fix_value,
                                                            int* __restrict__ rank) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (b >= B) return;
    const float* row = (domain[b] ? p2 : p1) + (long long)b * NI;
    const float pos = row[0] - fix_value;
    int c = 0;
    for (int k = 1 + lane; k < NI; k += 64) c += (row[k] > pos) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) rank[b] = c;
}

}  // namespace amid

using namespace amid;

extern "C" int amid_lnmean_fwd_f32(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, int B, int T, int D,
                                   float eps, float* u, void* stream) {
    AMID_CHECK_ARG(x && u && B > 0 && T > 0 && D > 0 && (D % 4) == 0);
    const int use_ln = (w0 != nullptr);
    AMID_CHECK_ARG(!use_ln || (b0 && w1 && b1));
    lnmean_fwd_kernel<<<2 * B, 256, 8 * D * sizeof(float), (hipStream_t)stream>>>(x, w0, b0, w1, b1, B, T, D, eps, use_ln, u);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_lnmean_bwd_f32(const float* x, const float* du, const float* w0, const float* w1, int B, int T, int D, float eps,
                                   float* dx, float* part, void* stream) {
    AMID_CHECK_ARG(x && du && dx && B > 0 && T > 0 && D > 0 && (D % 4) == 0);
    const int use_ln = (w0 != nullptr);
    AMID_CHECK_ARG(!use_ln || (w1 && part));
    lnmean_bwd_kernel<<<2 * B, 256, 16 * D * sizeof(float), (hipStream_t)stream>>>(x, du, w0, w1, B, T, D, eps, use_ln, dx, part);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// per-row partials of the scorer gradients: [w1 (hid x 2D) | b1 (hid) | w2 (hid) | b2 (1)], padded to whole float4s so that the
// fixed-order reduction (reduce_partials.h) takes its 16-byte path
extern "C" long long amid_scorer_part_floats(int D, int hid) { return ((long long)hid * 2 * D + 2 * hid + 1 + 3) & ~3LL; }

extern "C" int amid_scorer_bwd_f32(const float* u, const float* items, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* p1, const float* p2, const float* dp1, const float* dp2, int B, int NI, int D, int hid,
                                   float* du, float* ditems, float* part, int accumulate, void* stream) {
    AMID_CHECK_ARG(u && items && w1 && b1 && w2 && b2 && p1 && p2 && dp1 && dp2 && du && ditems && part && B > 0 && NI > 0 && hid > 0 &&
                   hid <= 64);
    ScorerBwdArgs a;
    a.u = u; a.items = items; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2;
    a.du = du; a.ditems = ditems; a.part = part; a.B = B; a.NI = NI; a.D = D; a.hid = hid; a.accumulate = accumulate ? 1 : 0;
    const size_t lds = (size_t)(4 * hid + hid + 4 + 128 * hid) * sizeof(float);
    scorer_bwd_kernel<<<B, 256, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_layernorm_rows_f32(const float* x, const float* w, const float* b, long long rows, int D, float eps, float* y,
                                       void* stream) {
    AMID_CHECK_ARG(x && w && b && y && rows > 0 && D > 0 && (D % 4) == 0);
    long long blocks = (rows + 7) / 8;
    if (blocks > 4096) blocks = 4096;
    layernorm_rows_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(x, w, b, rows, D, eps, y);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_sum_vector_f32(const float* v, int n, float* out, void* stream) {
    AMID_CHECK_ARG(v && out && n > 0);
    sum_vector_kernel<<<1, 64, 0, (hipStream_t)stream>>>(v, n, out);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_positive_rank_f32(const float* p1, const float* p2, const long long* domain_id, int B, int NI, float fix_value,
                                      int* rank, void* stream) {
    AMID_CHECK_ARG(p1 && p2 && domain_id && rank && B > 0 && NI > 0);
    positive_rank_kernel<<<(B + 3) / 4, 256, 0, (hipStream_t)stream>>>(p1, p2, domain_id, B, NI, fix_value, rank);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// pointer arrays: HOST arrays of 2 device pointers (domain-1 head output, domain-2 head output)
extern "C" int amid_dr_loss_f32(const float* const* p, const float* const* ips, const float* const* g, const float* labels,
                                const long long* domain_id, const long long* ob_label, int mode, float dr_e_w, int B, int NI,
                                float* const* dp, float* const* dips, float* const* dg, float* loss_part, void* stream) {
    AMID_CHECK_ARG(p && ips && g && labels && domain_id && dp && dips && dg && loss_part && B > 0 && NI > 0 && (mode == 0 || mode == 1));
    AMID_CHECK_ARG(mode == 0 || ob_label != nullptr);
    DrLossArgs a;
    for (int d = 0; d < 2; ++d) {
        AMID_CHECK_ARG(p[d] && ips[d] && g[d] && dp[d] && dips[d] && dg[d]);
        a.p[d] = p[d]; a.ips[d] = ips[d]; a.g[d] = g[d]; a.dp[d] = dp[d]; a.dips[d] = dips[d]; a.dg[d] = dg[d];
    }
    a.labels = labels; a.domain = domain_id; a.ob = ob_label; a.loss_part = loss_part; a.B = B; a.NI = NI; a.mode = mode; a.w = dr_e_w;
    dr_loss_kernel<<<B, 64, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
