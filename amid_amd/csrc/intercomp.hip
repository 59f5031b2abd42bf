// InterComp on the SASRec path (reference: InterComp.forward model_seq.py:483-497, used at model_seq.py:426-431 when
// isItC -- the configuration run.sh trains).  SURVEY.md A.4: the reference repeats the other domain's features `bs` times
// along a new axis and every slice computes the same thing, so the block it appends to each row is ONE [T, D] token group
// shared by the whole batch; and since SASRec only takes the mean over the (now 2T) time axis afterwards (:432-434), the
// whole module collapses onto the per-row means:
//     s_j     = max_{a,c} f_self[j,a] . f_other[j,c]                        (pair-max; symmetric in the two domains)
//     gate_j  = [ softmax_j(s)_j > threshold ]                              (softmax over the BATCH; no gradient)
//     z_g     = sum_j w_bs_g[j] gate_j mean_t f_other[j,t]                  [D]
//     c_g     = W_nn_g z_g + b_nn_g sum_j w_bs_g[j] + b_bs_g                = mean_t of the appended token group
//     u_g[b]  = 0.5 mean_t f_g[b,t] + 0.5 c_g
// Three small kernels: the pair-max per batch row (the only part that needs the full features), and a single-workgroup
// mix forward / backward over [B, D] means.  Batch-coupled by construction (trans_bs is Linear(bs, 1) over the batch).
#include "common.h"

namespace amid {

// ---- s_j: one workgroup per batch row; LN_last of both domains' rows staged in LDS, then all T x T dot products ----
struct PairMaxArgs {
    const float* x;                       // [2, B, T, D] output of the last encoder layer
    const float* lnw[2]; const float* lnb[2];
    float* s;                             // [B]
    float* u_raw;                         // optional [2, B, D]: mean_t LN_last(x[g, b]) (what amid_lnmean_fwd_f32 computes)
    int B, T, D; float eps;
};

__global__ __launch_bounds__(256) void itc_pairmax_kernel(const PairMaxArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, LD = D + 4, q = D >> 2;
    float* F[2] = {smem, smem + T * LD};
    __shared__ float red[4];
    const int b = blockIdx.x;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;          // 8 rows at a time, 32 lanes per row (D <= 128)
    for (int g = 0; g < 2; ++g) {
        const float* xb = a.x + ((long long)g * a.B + b) * T * D;
        const bool on = sub < q;
        float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) { ww = ld4(a.lnw[g] + 4 * sub); bb = ld4(a.lnb[g] + 4 * sub); }
        for (int t = rg; t < T; t += 8) {
            float4 y = on ? ld4(xb + (long long)t * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float mean = group_sum<32>(f4hsum(y)) / D;
            float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
            if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
            const float rstd = 1.0f / sqrtf(group_sum<32>(f4hsum(f4mul(d4, d4))) / D + a.eps);
            if (on) st4(F[g] + t * LD + 4 * sub, make_float4(d4.x * rstd * ww.x + bb.x, d4.y * rstd * ww.y + bb.y,
                                                             d4.z * rstd * ww.z + bb.z, d4.w * rstd * ww.w + bb.w));
        }
    }
    __syncthreads();
    if (a.u_raw != nullptr)
        for (int e = threadIdx.x; e < 2 * D; e += 256) {
            const int g = e / D, d = e - g * D;
            float acc = 0.f;
            for (int t = 0; t < T; ++t) acc += F[g][t * LD + d];
            a.u_raw[((long long)g * a.B + b) * D + d] = acc / T;
        }
    float best = -INFINITY;
    for (int p = threadIdx.x; p < T * T; p += 256) {
        const int i = p / T, j = p - i * T;
        const float* fa = F[0] + i * LD;
        const float* fc = F[1] + j * LD;
        float acc = 0.f;
        for (int c = 0; c < q; ++c) {
            const float4 u = ld4(fa + 4 * c), v = ld4(fc + 4 * c);
            acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
        }
        best = fmaxf(best, acc);
    }
    best = group_max<64>(best);
    if (lane_id() == 0) red[wave_id()] = best;
    __syncthreads();
    if (threadIdx.x == 0) a.s[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- mix, forward: one workgroup -------------------------------------------------------------------------------------
struct MixArgs {
    const float* u_raw;                   // [2, B, D] mean_t LN_last(x)
    const float* s;                       // [B]
    const float* wnn[2]; const float* bnn[2]; const float* wbs[2]; const float* bbs[2];   // itc_d{1,2}: [D,D], [D], [B], [1]
    float threshold;
    float* gate;                          // [B]
    float* z;                             // [2, D]
    float* sw;                            // [2]
    float* u_mix;                         // [2, B, D]
    // backward
    const float* du_mix;                  // [2, B, D]
    float* du_raw;                        // [2, B, D]
    float* dwnn[2]; float* dbnn[2]; float* dwbs[2]; float* dbbs[2];
    int B, D;
};

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {      // 1024 threads max; all threads get the result
    v = group_sum<64>(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;
}
__device__ __forceinline__ float block_reduce_max(float v, float* red) {
    v = group_max<64>(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    float s = -INFINITY;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s = fmaxf(s, red[k]);
    return s;
}

// Both mix kernels run on a grid of `nb` workgroups: the small shared quantities (gate, z, c / dc, dz: a few hundred values that
// need the whole batch) are recomputed by every workgroup from L2-resident inputs, then each workgroup handles its slice of the
// [2, B, D] elementwise part and of the parameter gradients.  (As single workgroups they were 58 us forward / 139 us backward of
// serial latency on the configuration run.sh trains.)
__global__ __launch_bounds__(1024) void itc_mix_fwd_kernel(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // gate [B rounded to 4] | z [2][D] | c [2][D] | part [8][2 D]
    __shared__ float red[16];
    const int B = a.B, D = a.D;
    float* gate_s = smem;
    float* z_s = smem + ((B + 3) & ~3);
    float* c_s = z_s + 2 * D;
    const bool first = blockIdx.x == 0;
    // softmax over the batch, thresholded (model_seq.py:490-491)
    float m = -INFINITY;
    for (int j = threadIdx.x; j < B; j += blockDim.x) m = fmaxf(m, a.s[j]);
    m = block_reduce_max(m, red);
    float l = 0.f, sw[2] = {0.f, 0.f};
    for (int j = threadIdx.x; j < B; j += blockDim.x) { l += expf(a.s[j] - m); sw[0] += a.wbs[0][j]; sw[1] += a.wbs[1][j]; }
    l = block_reduce_sum(l, red);
    sw[0] = block_reduce_sum(sw[0], red);
    sw[1] = block_reduce_sum(sw[1], red);
    for (int j = threadIdx.x; j < B; j += blockDim.x) {
        const float gt = (expf(a.s[j] - m) / l > a.threshold) ? 1.f : 0.f;
        gate_s[j] = gt;
        if (first) a.gate[j] = gt;
    }
    if (first && threadIdx.x < 2) a.sw[threadIdx.x] = sw[threadIdx.x];
    __syncthreads();
    // z_g[d] = sum_j w_g[j] gate_j u_raw[other(g)][j][d]: 8 row groups of 32 lanes (float4 per lane) stride over j with every load
    // independent, then a fixed-order sum of the 8 partials
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5, q = D >> 2, nrg = blockDim.x >> 5;
    const bool on = sub < q;
    float* part = c_s + 2 * D;                                        // [row groups][2 D]
    for (int g = 0; g < 2; ++g) {
        const float* uo = a.u_raw + (long long)(1 - g) * B * D;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int j = rg; j < B; j += nrg) {
            const float w = a.wbs[g][j] * gate_s[j];
            const float4 v = on ? ld4(uo + (long long)j * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
        if (on) st4(part + rg * 2 * D + g * D + 4 * sub, acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nrg; ++k) t += part[k * 2 * D + e];
        z_s[e] = t;
        if (first) a.z[e] = t;
    }
    __syncthreads();
    // c_g[o] = W_nn_g[o, :] . z_g + b_nn_g[o] sum_j w_g[j] + b_bs_g : one row per row group and pass
    for (int r = rg; r < 2 * D; r += nrg) {
        const int g = r / D, o = r - g * D;
        float t = 0.f;
        if (on) {
            const float4 w4 = ld4(a.wnn[g] + (long long)o * D + 4 * sub), z4 = ld4(z_s + g * D + 4 * sub);
            t = fmaf(w4.x, z4.x, fmaf(w4.y, z4.y, fmaf(w4.z, z4.z, w4.w * z4.w)));
        }
        t = group_sum<32>(t);
        if (sub == 0) c_s[r] = t + a.bnn[g][o] * sw[g] + a.bbs[g][0];
    }
    __syncthreads();
    const int per = (B + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    for (int g = 0; g < 2; ++g)
        for (int i = b0 * D + threadIdx.x; i < b1 * D; i += blockDim.x) {
            const long long o = (long long)g * B * D + i;
            a.u_mix[o] = 0.5f * a.u_raw[o] + 0.5f * c_s[g * D + i % D];      // mean over the 2T rows of cat(f, group)  (:432-434, :495)
        }
}

__global__ __launch_bounds__(1024) void itc_mix_bwd_kernel(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // dc [2][D] | dz [2][D]
    __shared__ float red[16];
    const int B = a.B, D = a.D;
    float* dc_s = smem;
    float* dz_s = smem + 2 * D;
    const bool first = blockIdx.x == 0;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5, q = D >> 2, nrg = blockDim.x >> 5;
    const bool on = sub < q;
    float* part = dz_s + 2 * D;                                       // [row groups][2 D]
    for (int g = 0; g < 2; ++g) {                                     // dc_g = 0.5 sum_b du_mix[g][b]: 8 row groups over b, then a fixed-order sum
        const float* p = a.du_mix + (long long)g * B * D;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int b = rg; b < B; b += nrg)
            if (on) acc = f4add(acc, ld4(p + (long long)b * D + 4 * sub));
        if (on) st4(part + rg * 2 * D + g * D + 4 * sub, acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nrg; ++k) t += part[k * 2 * D + e];
        dc_s[e] = 0.5f * t;
    }
    __syncthreads();
    for (int g = 0; g < 2; ++g) {                                     // dz_g = W_nn_g^T dc_g : row groups over the rows o of W_nn
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int o = rg; o < D; o += nrg) {
            const float c = dc_s[g * D + o];
            const float4 w4 = on ? ld4(a.wnn[g] + (long long)o * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc.x = fmaf(c, w4.x, acc.x); acc.y = fmaf(c, w4.y, acc.y); acc.z = fmaf(c, w4.z, acc.z); acc.w = fmaf(c, w4.w, acc.w);
        }
        __syncthreads();                                              // part is reused
        if (on) st4(part + rg * 2 * D + g * D + 4 * sub, acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {          // d b_nn_g = dc_g * sum_j w_g[j]
        float t = 0.f;
        for (int k = 0; k < nrg; ++k) t += part[k * 2 * D + e];
        dz_s[e] = t;
        if (first) a.dbnn[e / D][e % D] = dc_s[e] * a.sw[e / D];
    }
    __syncthreads();
    float bdc[2];
    for (int g = 0; g < 2; ++g) {                                     // d b_bs_g = sum_e dc_g[e] ; bdc_g = b_nn_g . dc_g
        float t = 0.f, u = 0.f;
        for (int e = threadIdx.x; e < D; e += blockDim.x) { t += dc_s[g * D + e]; u += a.bnn[g][e] * dc_s[g * D + e]; }
        t = block_reduce_sum(t, red);
        bdc[g] = block_reduce_sum(u, red);
        if (first && threadIdx.x == 0) a.dbbs[g][0] = t;
    }
    __syncthreads();
    {   // d W_nn_g = dc_g (x) z_g : this workgroup's rows
        const int per = (D + gridDim.x - 1) / gridDim.x;
        const int o0 = blockIdx.x * per, o1 = min(D, o0 + per);
        for (int g = 0; g < 2; ++g)
            for (int i = o0 * D + threadIdx.x; i < o1 * D; i += blockDim.x) a.dwnn[g][i] = dc_s[g * D + i / D] * a.z[g * D + i % D];
    }
    const int per = (B + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    // d w_bs_g[j] = gate_j (u_raw[other(g)][j] . dz_g) + b_nn_g . dc_g : one wave per (g, j)
    const int nw = blockDim.x >> 6, w = wave_id(), lane = lane_id();
    for (int p = w; p < 2 * (b1 - b0); p += nw) {
        const int g = p / (b1 - b0), j = b0 + p - g * (b1 - b0);
        const float* uo = a.u_raw + ((long long)(1 - g) * B + j) * D;
        float t = 0.f;
        for (int d = lane; d < D; d += 64) t = fmaf(uo[d], dz_s[g * D + d], t);
        t = group_sum<64>(t);
        if (lane == 0) a.dwbs[g][j] = a.gate[j] * t + bdc[g];
    }
    // d u_raw[g][b] = 0.5 d u_mix[g][b] + gate_b w_bs_{g'}[b] dz_{g'},  g' = the domain whose group is built from g
    for (int g = 0; g < 2; ++g)
        for (int i = b0 * D + threadIdx.x; i < b1 * D; i += blockDim.x) {
            const int b = i / D, d = i - b * D;
            const long long o = (long long)g * B * D + i;
            a.du_raw[o] = 0.5f * a.du_mix[o] + a.gate[b] * a.wbs[1 - g][b] * dz_s[(1 - g) * D + d];
        }
}

// ---- fast forms for B <= 256, D <= 128 (the shapes run.sh trains): 512 threads = 16 row groups of 32 lanes, and EVERY global
// operand of the kernel is requested in its first instructions -- u_raw rows (16 per row group and domain), the W_nn rows, the
// scores, w_bs -- so the kernel pays one memory latency instead of one per phase (the looped forms above: ~7 dependent phases,
// 19 / 16 us at B = 256).  Same arithmetic order as the looped forms with 16 row groups.
constexpr int MIXF_RG = 16, MIXF_K = 16;          // rows per row group: B <= MIXF_RG * MIXF_K; W_nn rows per row group: 2 D / 16 <= 16

__global__ __launch_bounds__(512) void itc_mix_fwd_fast_kernel(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // s [B4] | wb [2][B4] | gate [B4] | z [2][D] | c [2][D] | part [16][2 D]
    __shared__ float red[16];
    const int B = a.B, D = a.D, B4 = (B + 3) & ~3, q = D >> 2;
    float* s_s = smem;
    float* wb_s = s_s + B4;
    float* gate_s = wb_s + 2 * B4;
    float* z_s = gate_s + B4;
    float* c_s = z_s + 2 * D;
    float* part = c_s + 2 * D;
    const bool first = blockIdx.x == 0;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = sub < q;
    // ---- every global load, up front ----
    float4 uv[2][MIXF_K], wv[MIXF_K];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* uo = a.u_raw + (long long)(1 - g) * B * D;
#pragma unroll
        for (int k = 0; k < MIXF_K; ++k) {
            const int j = rg + k * MIXF_RG;
            uv[g][k] = (on && j < B) ? ld4(uo + (long long)j * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // (every __syncthreads() waits for ALL outstanding global loads, so a load issued in front of a barrier is not prefetched at all)
#pragma unroll
    for (int k = 0; k < MIXF_K; ++k) {
        const int r = rg + k * MIXF_RG;
        const int g = (r >= D) ? 1 : 0, o = r - g * D;
        wv[k] = (on && r < 2 * D) ? ld4(a.wnn[g] + (long long)o * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int j = threadIdx.x; j < B; j += 512) { s_s[j] = a.s[j]; wb_s[j] = a.wbs[0][j]; wb_s[B4 + j] = a.wbs[1][j]; }
    const int per = (B + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    float4 own[2];                                                    // this workgroup's slice of u_raw (<= 8 rows x 2 domains x 32 quads)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int i = threadIdx.x;                                    // (row b0 + i / q, quad i % q)
        const int b = b0 + i / q;
        own[g] = (b < b1) ? ld4(a.u_raw + ((long long)g * B + b) * D + 4 * (i % q)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float bias_bs0 = a.bbs[0][0], bias_bs1 = a.bbs[1][0];
    __syncthreads();
    // softmax over the batch, thresholded (model_seq.py:490-491), from LDS
    float m = -INFINITY;
    for (int j = threadIdx.x; j < B; j += 512) m = fmaxf(m, s_s[j]);
    m = block_reduce_max(m, red);
    float l = 0.f, sw[2] = {0.f, 0.f};
    for (int j = threadIdx.x; j < B; j += 512) { l += expf(s_s[j] - m); sw[0] += wb_s[j]; sw[1] += wb_s[B4 + j]; }
    {   // the three sums through ONE pair of barriers
        __shared__ float red3[3][8];
        l = group_sum<64>(l); sw[0] = group_sum<64>(sw[0]); sw[1] = group_sum<64>(sw[1]);
        if (lane_id() == 0) { red3[0][wave_id()] = l; red3[1][wave_id()] = sw[0]; red3[2][wave_id()] = sw[1]; }
        __syncthreads();
        l = 0.f; sw[0] = 0.f; sw[1] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { l += red3[0][k]; sw[0] += red3[1][k]; sw[1] += red3[2][k]; }
    }
    for (int j = threadIdx.x; j < B; j += 512) {
        const float gt = (expf(s_s[j] - m) / l > a.threshold) ? 1.f : 0.f;
        gate_s[j] = gt;
        if (first) a.gate[j] = gt;
    }
    if (first && threadIdx.x < 2) a.sw[threadIdx.x] = sw[threadIdx.x];
    __syncthreads();
    // z_g[d] = sum_j w_g[j] gate_j u_raw[other(g)][j][d]
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < MIXF_K; ++k) {
            const int j = rg + k * MIXF_RG;
            const float w = (j < B) ? wb_s[g * B4 + j] * gate_s[j] : 0.f;
            acc.x = fmaf(w, uv[g][k].x, acc.x); acc.y = fmaf(w, uv[g][k].y, acc.y);
            acc.z = fmaf(w, uv[g][k].z, acc.z); acc.w = fmaf(w, uv[g][k].w, acc.w);
        }
        if (on) st4(part + rg * 2 * D + g * D + 4 * sub, acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += 512) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < MIXF_RG; ++k) t += part[k * 2 * D + e];
        z_s[e] = t;
        if (first) a.z[e] = t;
    }
    __syncthreads();
    // c_g[o] = W_nn_g[o, :] . z_g + b_nn_g[o] sum_j w_g[j] + b_bs_g.  The row group's 16 rows are reduced over its 32 lanes TOGETHER:
    // a butterfly in which the lane halves swap the rows they do not keep (16 shuffles in 5 dependent levels; one
    // group_sum<32> per row was 80 shuffles in 80 levels -- 7 us of ds_bpermute latency)
    {
        float v[MIXF_K];
#pragma unroll
        for (int k = 0; k < MIXF_K; ++k) {
            const int r = rg + k * MIXF_RG;
            const float4 z4 = on ? ld4(z_s + ((r >= D) ? D : 0) + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            v[k] = fmaf(wv[k].x, z4.x, fmaf(wv[k].y, z4.y, fmaf(wv[k].z, z4.z, wv[k].w * z4.w)));
        }
#pragma unroll
        for (int half = 8, bit = 16; half >= 1; half >>= 1, bit >>= 1) {      // xor 16, 8, 4, 2: keep `half` rows, send the other half
            const bool hi = (sub & bit) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float keep = hi ? v[i + half] : v[i], send = hi ? v[i] : v[i + half];
                v[i] = keep + __shfl_xor(send, bit, 64);
            }
        }
        v[0] += __shfl_xor(v[0], 1, 64);
        // lane `sub` now holds row k = 8 b4 + 4 b3 + 2 b2 + b1 of its row group (both lanes of a pair hold it)
        const int kk = ((sub >> 4) & 1) * 8 + ((sub >> 3) & 1) * 4 + ((sub >> 2) & 1) * 2 + ((sub >> 1) & 1);
        const int r = rg + kk * MIXF_RG;
        if ((sub & 1) == 0 && r < 2 * D) {
            const int g = (r >= D) ? 1 : 0, o = r - g * D;
            c_s[r] = v[0] + a.bnn[g][o] * sw[g] + (g ? bias_bs1 : bias_bs0);
        }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int i = threadIdx.x, b = b0 + i / q, c4 = 4 * (i % q);
        if (b < b1) {
            const float4 cc = ld4(c_s + g * D + c4);
            st4(a.u_mix + ((long long)g * B + b) * D + c4, make_float4(0.5f * own[g].x + 0.5f * cc.x, 0.5f * own[g].y + 0.5f * cc.y,
                                                                     0.5f * own[g].z + 0.5f * cc.z, 0.5f * own[g].w + 0.5f * cc.w));
        }
    }
}

__global__ __launch_bounds__(512) void itc_mix_bwd_fast_kernel(const MixArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // dc [2][D] | dz [2][D] | part [16][2 D]
    __shared__ float red[16];
    const int B = a.B, D = a.D, q = D >> 2;
    float* dc_s = smem;
    float* dz_s = smem + 2 * D;
    float* part = dz_s + 2 * D;
    const bool first = blockIdx.x == 0;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const bool on = sub < q;
    const int per = (B + gridDim.x - 1) / gridDim.x;
    const int b0 = blockIdx.x * per, b1 = min(B, b0 + per);
    // ---- every global load, up front ----
    float4 dv[2][MIXF_K], wv[2][MIXF_K / 2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const float* p = a.du_mix + (long long)g * B * D;
#pragma unroll
        for (int k = 0; k < MIXF_K; ++k) {
            const int b = rg + k * MIXF_RG;
            dv[g][k] = (on && b < B) ? ld4(p + (long long)b * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < MIXF_K / 2; ++k) {
            const int o = rg + k * MIXF_RG;
            wv[g][k] = (on && o < D) ? ld4(a.wnn[g] + (long long)o * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 own_u[2], own_d[2];                                        // slices of u_raw (for d w_bs) and of du_mix (for du_raw)
    float own_gw[2];                                                  // gate_b * w_bs_{1-g}[b] of the slice rows
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int i = threadIdx.x, b = b0 + i / q, c4 = 4 * (i % q);
        const bool ok = b < b1;
        own_u[g] = ok ? ld4(a.u_raw + ((long long)(1 - g) * B + b) * D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        own_d[g] = ok ? ld4(a.du_mix + ((long long)g * B + b) * D + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        own_gw[g] = ok ? a.gate[b] * a.wbs[1 - g][b] : 0.f;
    }
    const float sw0 = a.sw[0], sw1 = a.sw[1];
    // dc_g = 0.5 sum_b du_mix[g][b]
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < MIXF_K; ++k) acc = f4add(acc, dv[g][k]);
        if (on) st4(part + rg * 2 * D + g * D + 4 * sub, acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += 512) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < MIXF_RG; ++k) t += part[k * 2 * D + e];
        dc_s[e] = 0.5f * t;
    }
    __syncthreads();
    // dz_g = W_nn_g^T dc_g
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < MIXF_K / 2; ++k) {
            const int o = rg + k * MIXF_RG;
            const float c = (o < D) ? dc_s[g * D + o] : 0.f;
            acc.x = fmaf(c, wv[g][k].x, acc.x); acc.y = fmaf(c, wv[g][k].y, acc.y); acc.z = fmaf(c, wv[g][k].z, acc.z); acc.w = fmaf(c, wv[g][k].w, acc.w);
        }
        if (on) st4(part + rg * 2 * D + g * D + 4 * sub, acc);       // part's dc partials are all read (barrier above)
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += 512) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < MIXF_RG; ++k) t += part[k * 2 * D + e];
        dz_s[e] = t;
        if (first) a.dbnn[e / D][e % D] = dc_s[e] * (e < D ? sw0 : sw1);
    }
    __syncthreads();
    float bdc[2];
    for (int g = 0; g < 2; ++g) {                                     // d b_bs_g = sum_e dc_g[e] ; bdc_g = b_nn_g . dc_g
        float t = 0.f, u = 0.f;
        for (int e = threadIdx.x; e < D; e += 512) { t += dc_s[g * D + e]; u += a.bnn[g][e] * dc_s[g * D + e]; }
        t = block_reduce_sum(t, red);
        bdc[g] = block_reduce_sum(u, red);
        if (first && threadIdx.x == 0) a.dbbs[g][0] = t;
    }
    {   // d W_nn_g = dc_g (x) z_g : this workgroup's rows
        const int perw = (D + gridDim.x - 1) / gridDim.x;
        const int o0 = blockIdx.x * perw, o1 = min(D, o0 + perw);
        for (int g = 0; g < 2; ++g)
            for (int i = o0 * D + threadIdx.x; i < o1 * D; i += 512) a.dwnn[g][i] = dc_s[g * D + i / D] * a.z[g * D + i % D];
    }
    // slice rows: d w_bs_g[j] = gate_j (u_raw[other(g)][j] . dz_g) + b_nn_g . dc_g ; d u_raw[g][b] = 0.5 d u_mix[g][b] + gate_b w_bs_{g'}[b] dz_{g'}
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int i = threadIdx.x, b = b0 + i / q, c4 = 4 * (i % q);
        const float4 z4 = ld4(dz_s + g * D + c4);
        float t = fmaf(own_u[g].x, z4.x, fmaf(own_u[g].y, z4.y, fmaf(own_u[g].z, z4.z, own_u[g].w * z4.w)));
        // the q lanes of a row are consecutive threads: q = 32 -> one half-wave per row, q = 16 -> a quarter
        if (q == 32) t = group_sum<32>(t); else t = group_sum<16>(t);
        if (b < b1) {
            if ((i % q) == 0) a.dwbs[g][b] = a.gate[b] * t + bdc[g];
            const float4 zo = ld4(dz_s + (1 - g) * D + c4);
            st4(a.du_raw + ((long long)g * B + b) * D + c4, make_float4(fmaf(own_gw[g], zo.x, 0.5f * own_d[g].x), fmaf(own_gw[g], zo.y, 0.5f * own_d[g].y),
                                                                      fmaf(own_gw[g], zo.z, 0.5f * own_d[g].z), fmaf(own_gw[g], zo.w, 0.5f * own_d[g].w)));
        }
    }
}

}  // namespace amid

using namespace amid;

// Pointer-array parameters are HOST arrays of 2 device pointers (itc_d1, itc_d2 / sac1, sac2).
extern "C" int amid_itc_pairmax_f32(const float* x, const float* const* ln_w, const float* const* ln_b, int B, int T, int D, float eps,
                                    float* s, float* u_raw, void* stream) {
    AMID_CHECK_ARG(x && ln_w && ln_b && ln_w[0] && ln_w[1] && ln_b[0] && ln_b[1] && s && B > 0 && T > 0 && D > 0 && (D % 4) == 0);
    if (D > 128) return AMID_ERR_UNSUPPORTED;
    const size_t lds = (size_t)2 * T * (D + 4) * sizeof(float);
    if (lds > 160 * 1024 - 256) return AMID_ERR_UNSUPPORTED;       // T <= 151 at D = 128 (both domains' LayerNorm'd rows of a batch row)
    PairMaxArgs a;
    a.x = x; a.s = s; a.u_raw = u_raw; a.B = B; a.T = T; a.D = D; a.eps = eps;
    for (int g = 0; g < 2; ++g) { a.lnw[g] = ln_w[g]; a.lnb[g] = ln_b[g]; }
    hipError_t e = hipFuncSetAttribute((const void*)itc_pairmax_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    itc_pairmax_kernel<<<B, 256, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int mix_fill(MixArgs& a, const float* u_raw, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs,
                    const float* const* b_bs, int B, int D) {
    AMID_CHECK_ARG(u_raw && w_nn && b_nn && w_bs && b_bs && B > 0 && D > 0 && (D % 4) == 0);
    a.u_raw = u_raw; a.B = B; a.D = D;
    for (int g = 0; g < 2; ++g) {
        AMID_CHECK_ARG(w_nn[g] && b_nn[g] && w_bs[g] && b_bs[g]);
        a.wnn[g] = w_nn[g]; a.bnn[g] = b_nn[g]; a.wbs[g] = w_bs[g]; a.bbs[g] = b_bs[g];
    }
    return AMID_OK;
}

extern "C" int amid_itc_mix_fwd_f32(const float* u_raw, const float* s, const float* const* w_nn, const float* const* b_nn,
                                    const float* const* w_bs, const float* const* b_bs, float threshold, int B, int D, float* gate,
                                    float* z, float* sw, float* u_mix, void* stream) {
    MixArgs a = {};
    if (int e = mix_fill(a, u_raw, w_nn, b_nn, w_bs, b_bs, B, D)) return e;
    AMID_CHECK_ARG(s && gate && z && sw && u_mix);
    a.s = s; a.threshold = threshold; a.gate = gate; a.z = z; a.sw = sw; a.u_mix = u_mix;
    if (B <= MIXF_RG * MIXF_K && (D == 64 || D == 128) && B >= 32 && (B / 32) * (D / 4) <= 512 && ((B + 31) / 32) * (D / 4) <= 512) {
        // slice per workgroup: ceil(B / 32) rows x D / 4 quads <= 512 threads
        const size_t ldsf = (size_t)(4 * ((B + 3) & ~3) + 4 * D + MIXF_RG * 2 * D) * sizeof(float);
        itc_mix_fwd_fast_kernel<<<32, 512, ldsf, (hipStream_t)stream>>>(a);
        AMID_LAUNCH_CHECK();
        return AMID_OK;
    }
    const size_t lds = (size_t)(((B + 3) & ~3) + 4 * D + 32 * 2 * D) * sizeof(float);
    if (lds > 60 * 1024) return AMID_ERR_UNSUPPORTED;
    itc_mix_fwd_kernel<<<B < 32 ? B : 32, 1024, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_itc_mix_bwd_f32(const float* du_mix, const float* u_raw, const float* gate, const float* z, const float* sw,
                                    const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B, int D,
                                    float* du_raw, float* const* dw_nn, float* const* db_nn, float* const* dw_bs, float* const* db_bs,
                                    void* stream) {
    MixArgs a = {};
    const float* dummy[2] = {sw, sw};
    if (int e = mix_fill(a, u_raw, w_nn, b_nn, w_bs, dummy, B, D)) return e;
    AMID_CHECK_ARG(du_mix && gate && z && sw && du_raw && dw_nn && db_nn && dw_bs && db_bs);
    a.du_mix = du_mix; a.gate = const_cast<float*>(gate); a.z = const_cast<float*>(z); a.sw = const_cast<float*>(sw); a.du_raw = du_raw;
    for (int g = 0; g < 2; ++g) {
        AMID_CHECK_ARG(dw_nn[g] && db_nn[g] && dw_bs[g] && db_bs[g]);
        a.dwnn[g] = dw_nn[g]; a.dbnn[g] = db_nn[g]; a.dwbs[g] = dw_bs[g]; a.dbbs[g] = db_bs[g];
    }
    if (B <= MIXF_RG * MIXF_K && (D == 64 || D == 128) && B >= 32 && ((B + 31) / 32) * (D / 4) <= 512) {
        itc_mix_bwd_fast_kernel<<<32, 512, (size_t)(4 * D + MIXF_RG * 2 * D) * sizeof(float), (hipStream_t)stream>>>(a);
        AMID_LAUNCH_CHECK();
        return AMID_OK;
    }
    itc_mix_bwd_kernel<<<B < 32 ? B : 32, 1024, (size_t)(4 * D + 32 * 2 * D) * sizeof(float), (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
