// InnerComp on the SASRec path (reference: InnerComp.forward model_seq.py:459-472, used at model_seq.py:422-424 when isInC,
// BEFORE the encoders, on the raw gathered item rows; the encoders then run over 2T tokens with 2T-row pos_emb tables,
// model_seq.py:398-401).  As for InterComp (csrc/intercomp.hip) the reference repeats the batch `bs` times along a
// new axis and every slice computes the same thing, so the block appended to each row is ONE [T, D] token group per domain:
//     s_j     = max_{a,c} e[j,a] . e[j,c]                                   (self pair-max of row j's gathered rows, :463-464)
//     gate_j  = [ softmax_j(s)_j > threshold ]                              (softmax over the BATCH, :465-466; no gradient)
//     S[t]    = sum_j w_bs[j] gate_j e[j,t]                                 [T, D]
//     Z[t]    = W_nn S[t] + b_nn sum_j w_bs[j] + b_bs                       (:467-469) the appended tokens, same for every row
//     x[b]    = [ e[b,0..T) | Z[0..T) ] + P[0..2T), dropout, (==0) mask     (:470 then Log2feats :361-366)
// Unlike InterComp the tokens pass through the encoder, so the full [T, D] group is kept.  Backward: the encoder-input gradient
// of the appended half summed over the batch IS the pos_emb gradient of rows T..2T-1 (both are sum_b of the same rows), so dZ is
// read from the embedding backward's pos_emb partials; then
//     dS[t]   = dZ[t] W_nn            dW_nn = sum_t dZ[t]^T S[t]      db_nn = (sum_j w_bs[j]) sum_t dZ[t]     db_bs = sum_{t,d} dZ
//     dw_bs[j]= gate_j sum_t dS[t] . e[j,t] + sum_t dZ[t] . b_nn
//     de[j,t] = (encoder-input gradient of the row's own half) + w_bs[j] gate_j dS[t]     -> the table-row gradient buffer
// Batch-coupled by construction (trans_bs is Linear(bs, 1) over the batch).  None of this is on the cfg 2 hot path.
#include "common.h"
#include "rng.h"

namespace amid {

// ---- s[g][j]: one workgroup per (row j, domain g); the row's T gathered rows in LDS, all pairs a <= c.  cross (InterComp in front
// of BERT4Rec's encoders, model_seq.py:289-293): a runs over domain g's rows and c over the OTHER domain's, all T x T pairs ----
__global__ __launch_bounds__(256) void inc_score_kernel(const float* __restrict__ xg, int B, int T, int D, int cross, float* __restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4];
    const int LD = D + 4, q = D >> 2;
    const int j = blockIdx.x, g = blockIdx.y;
    const float* e = xg + ((long long)g * B + j) * T * D;
    for (int i = threadIdx.x; i < T * q; i += 256) {
        const int t = i / q, c = i - t * q;
        st4(smem + t * LD + 4 * c, ld4(e + (long long)t * D + 4 * c));
    }
    const float* other = smem;
    if (cross) {
        const float* e2 = xg + ((long long)(1 - g) * B + j) * T * D;
        float* o2 = smem + T * LD;
        for (int i = threadIdx.x; i < T * q; i += 256) {
            const int t = i / q, c = i - t * q;
            st4(o2 + t * LD + 4 * c, ld4(e2 + (long long)t * D + 4 * c));
        }
        other = o2;
    }
    __syncthreads();
    float best = -INFINITY;
    for (int p = threadIdx.x; p < T * T; p += 256) {
        const int a = p / T, c = p - a * T;
        if (!cross && c < a) continue;
        const float* fa = smem + a * LD;
        const float* fc = other + c * LD;
        float acc = 0.f;
        for (int k = 0; k < q; ++k) {
            const float4 u = ld4(fa + 4 * k), v = ld4(fc + 4 * k);
            acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
        }
        best = fmaxf(best, acc);
    }
    best = group_max<64>(best);
    if (lane_id() == 0) red[wave_id()] = best;
    __syncthreads();
    if (threadIdx.x == 0) s[g * B + j] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

struct IncFwdArgs {
    const float* xg;                     // [2, B, T, D] gathered rows
    const float* s;                      // [2, B]
    const float* w_nn[2]; const float* b_nn[2]; const float* w_bs[2]; const float* b_bs[2];
    float threshold;
    int B, T, D;
    int cross;                           // 1: module g mixes the OTHER domain's rows (InterComp), 0: its own (InnerComp)
    float* gate;                         // [2, B]
    float* S;                            // [2, T, D]
    float* Z;                            // [2, T, D]
    float* sw;                           // [2]
    // data-parallel shard (amid_inc_embed_fwd_shard_f32): the B rows are samples j0 .. j0 + B - 1 of a global batch of Bg; s then holds
    // the GLOBAL scores [2, Bg], w_bs spans Bg, and the launch is cut at the all-reduce of S: phase 1 = gates + this shard's partial S,
    // phase 2 = Z from the summed S.  One GPU: Bg = B, j0 = 0, phase 0 (both).
    int Bg, j0, phase;
};

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = group_sum<64>(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float block_max256(float v, float* red) {
    v = group_max<64>(v);
    __syncthreads();
    if (lane_id() == 0) red[wave_id()] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- one workgroup per (t, g): batch softmax + gate (recomputed by every workgroup: B values), S[t], Z[t] ----
__global__ __launch_bounds__(256) void inc_mix_fwd_kernel(const IncFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // c[B] | part[8][D] | S[D]
    __shared__ float red[4];
    const int B = a.B, T = a.T, D = a.D, q = D >> 2;
    const int t = blockIdx.x, g = blockIdx.y;
    const int sd = a.cross ? 1 - g : g;
    float* cj = smem;
    float* part = smem + ((B + 3) & ~3);
    float* Srow = part + 8 * D;
    const int Bg = a.Bg, j0 = a.j0;
    const float* sg = a.s + g * Bg;                      // the softmax runs over the GLOBAL batch
    float wsum = 0.f;
    if (a.phase != 2) {
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < Bg; j += 256) mx = fmaxf(mx, sg[j]);
    mx = block_max256(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < Bg; j += 256) { sum += expf(sg[j] - mx); wsum += a.w_bs[g][j]; }
    sum = block_sum256(sum, red);
    wsum = block_sum256(wsum, red);
    for (int j = threadIdx.x; j < B; j += 256) {
        const float gt = (expf(sg[j0 + j] - mx) / sum > a.threshold) ? 1.f : 0.f;
        cj[j] = a.w_bs[g][j0 + j] * gt;
        if (t == 0) a.gate[g * B + j] = gt;
    }
    if (t == 0 && threadIdx.x == 0) a.sw[g] = wsum;
    __syncthreads();
    }
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;
    if (a.phase != 2)
    for (int c = sub; c < q; c += 32) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = rg; j < B; j += 8) {
            const float w = cj[j];
            if (w != 0.f) {
                const float4 v = ld4(a.xg + (((long long)sd * B + j) * T + t) * D + 4 * c);
                acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
            }
        }
        st4(part + rg * D + 4 * c, acc);
    }
    __syncthreads();
    if (a.phase != 2) {
        for (int d = threadIdx.x; d < D; d += 256) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += part[k * D + d];
            Srow[d] = v;
            a.S[((long long)g * T + t) * D + d] = v;
        }
        if (a.phase == 1) return;                        // the shards' partial sums are added by the caller's all-reduce
    } else {
        for (int d = threadIdx.x; d < D; d += 256) Srow[d] = a.S[((long long)g * T + t) * D + d];
        wsum = a.sw[g];
    }
    __syncthreads();
    const float bias_bs = a.b_bs[g][0];
    for (int d = threadIdx.x; d < D; d += 256) {
        const float* wr = a.w_nn[g] + (long long)d * D;
        float acc = 0.f;
        for (int k = 0; k < D; ++k) acc = fmaf(wr[k], Srow[k], acc);
        a.Z[((long long)g * T + t) * D + d] = acc + a.b_nn[g][d] * wsum + bias_bs;
    }
}

// ---- encoder input: x[g,b,t'] = (t' < T ? e[g,b,t'] : Z[g,t'-T]) + P_g[t'], dropout, (==0) mask; half-wave per row ----
__global__ __launch_bounds__(256) void inc_embed_fwd_kernel(const float* __restrict__ xg, const float* __restrict__ Z,
                                                            const float* __restrict__ pos0, const float* __restrict__ pos1, int B, int T,
                                                            int D, float* __restrict__ x0, unsigned char* __restrict__ tmq,
                                                            const RngState* __restrict__ rng, int train, unsigned thr16, float scale) {
    const int sub = threadIdx.x & 31;
    const int hw = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    const int n_hw = gridDim.x * (blockDim.x >> 5);
    const int q = D >> 2, Te = 2 * T, Me = B * Te;
    unsigned long long seed = 0;
    unsigned step = 0;
    if (train) { seed = rng->seed; step = (unsigned)rng->step; }
    for (int r = hw; r < 2 * Me; r += n_hw) {
        const int g = r >= Me;
        const int local = r - g * Me;
        const int b = local / Te, t = local - b * Te;
        const float* src = (t < T) ? xg + (((long long)g * B + b) * T + t) * D : Z + ((long long)g * T + (t - T)) * D;
        const float* pp = pos0 ? (g ? pos1 : pos0) + (long long)t * D : nullptr;        // BERT4Rec: no positional rows
        for (int c = sub; c < q; c += 32) {
            float4 x = ld4(src + 4 * c);
            if (pp) x = f4add(x, ld4(pp + 4 * c));
            const unsigned bits = (x.x == 0.f ? 1u : 0u) | (x.y == 0.f ? 2u : 0u) | (x.z == 0.f ? 4u : 0u) | (x.w == 0.f ? 8u : 0u);
            if (train) x = f4mul(x, dropout_mult4(seed, site_id(g, 0, SITE_EMB), step, (unsigned long long)local * D + 4 * c, thr16, scale));
            if (bits) {
                if (bits & 1u) x.x = 0.f;
                if (bits & 2u) x.y = 0.f;
                if (bits & 4u) x.z = 0.f;
                if (bits & 8u) x.w = 0.f;
            }
            if (tmq) tmq[(long long)r * q + c] = (unsigned char)bits;
            st4(x0 + (long long)r * D + 4 * c, x);
        }
    }
}

struct IncBwdArgs {
    const float* dpos_part;              // [nsplit][2][2T][D] partials of the embedding backward (rows T.. are dZ's partials);
    int nsplit;                          // nullptr: no embedding backward ran (BERT4Rec) -- dZ[t] = sum_b dx0[b, T + t] is formed here
    int cross;                           // as IncFwdArgs
    const float* xg;                     // [2, B, T, D]
    const float* dx0;                    // [2, B, 2T, D] encoder-input gradient after the dropout / mask backward
    const float* gate; const float* S; const float* sw;
    const float* w_nn[2]; const float* b_nn[2]; const float* w_bs[2];
    int B, T, D;
    float* dZ;                           // [2, T, D]
    float* dS;                           // [2, T, D]
    float* rows;                         // [2, T, 2]: (sum_d dZ[t][d], sum_d dZ[t][d] b_nn[d])
    float* dw_nn[2]; float* db_nn[2]; float* dw_bs[2]; float* db_bs[2];
    float* dxg;                          // [2, B, T, D] table-row gradients (seq rows)
    // data-parallel shard (amid_inc_bwd_shard_f32; as IncFwdArgs): phase 1 = this shard's dZ only, phase 2 = everything behind the
    // all-reduce of dZ; gscale = 1 / world on the gradients every rank computes alike from global operands (W_nn, b_nn, b_bs -- the
    // step's dense exchange sums the ranks); w_bs's gradient is written for the shard's own Bg-slice (the rest zeroed by the host side)
    int Bg, j0, phase;
    float gscale;
};

// ---- grid (T, 2): dZ[t] (fixed-order sum of the partials), dS[t] = dZ[t] W_nn, the two row sums ----
__global__ __launch_bounds__(256) void inc_dz_kernel(const IncBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // dZ row [D] | column partials [256 / D][D]
    __shared__ float red[4];
    const int B = a.B, T = a.T, D = a.D;
    const int t = blockIdx.x, g = blockIdx.y;
    if (a.phase == 2) {                  // dZ is already the world's sum
        for (int d = threadIdx.x; d < D; d += 256) smem[d] = a.dZ[((long long)g * T + t) * D + d];
    } else
    if (!a.dpos_part) {                  // fixed-order column sum over the batch, 256 / D interleaved partial sums per column
        const int nh = 256 / D > 0 ? 256 / D : 1;
        float* cp = smem + D;
        for (int i = threadIdx.x; i < nh * D; i += 256) {
            const int d = i % D, h = i / D;
            float v = 0.f;
            for (int j = h; j < B; j += nh) v += a.dx0[(((long long)g * B + j) * 2 * T + T + t) * D + d];
            cp[h * D + d] = v;
        }
        __syncthreads();
    }
    float rs = 0.f, rb = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        if (a.phase == 2) {
            v = smem[d];
        } else {
            if (a.dpos_part) {
                for (int z = 0; z < a.nsplit; ++z) v += a.dpos_part[(((long long)z * 2 + g) * 2 * T + T + t) * D + d];
            } else {
                const int nh = 256 / D > 0 ? 256 / D : 1;
                for (int h = 0; h < nh; ++h) v += smem[D + h * D + d];
            }
            smem[d] = v;
            a.dZ[((long long)g * T + t) * D + d] = v;
        }
        rs += v;
        rb = fmaf(v, a.b_nn[g][d], rb);
    }
    if (a.phase == 1) return;            // (uniform over the workgroup)
    rs = block_sum256(rs, red);
    rb = block_sum256(rb, red);
    if (threadIdx.x == 0) { a.rows[(g * T + t) * 2] = rs; a.rows[(g * T + t) * 2 + 1] = rb; }
    __syncthreads();
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc = 0.f;
        for (int d = 0; d < D; ++d) acc = fmaf(smem[d], a.w_nn[g][(long long)d * D + k], acc);
        a.dS[((long long)g * T + t) * D + k] = acc;
    }
}

// ---- grid (D, 2): dW_nn[d][:], db_nn[d]; workgroup d == 0 also db_bs ----
__global__ __launch_bounds__(256) void inc_wgrad_kernel(const IncBwdArgs a) {
    __shared__ float red[4];
    const int T = a.T, D = a.D;
    const int d = blockIdx.x, g = blockIdx.y;
    for (int k = threadIdx.x; k < D; k += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc = fmaf(a.dZ[((long long)g * T + t) * D + d], a.S[((long long)g * T + t) * D + k], acc);
        a.dw_nn[g][(long long)d * D + k] = acc * a.gscale;
    }
    float cs = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) cs += a.dZ[((long long)g * T + t) * D + d];
    cs = block_sum256(cs, red);
    if (threadIdx.x == 0) a.db_nn[g][d] = cs * a.sw[g] * a.gscale;
    if (d == 0) {
        float all = 0.f;
        for (int t = threadIdx.x; t < T; t += 256) all += a.rows[(g * T + t) * 2];
        all = block_sum256(all, red);
        if (threadIdx.x == 0) a.db_bs[g][0] = all * a.gscale;
    }
}

// ---- grid (B, 2): dw_bs[j] of module g and the table-row gradients of the rows it mixed (its own domain's, or with cross the
// other domain's: every (j, domain) is the source of exactly one module either way, so the output is covered once) ----
__global__ __launch_bounds__(256) void inc_scatter_bwd_kernel(const IncBwdArgs a) {
    __shared__ float red[4];
    const int B = a.B, T = a.T, D = a.D, q = D >> 2;
    const int j = blockIdx.x, g = blockIdx.y;
    const int sd = a.cross ? 1 - g : g;
    const float gt = a.gate[g * B + j];
    const float cj = a.w_bs[g][a.j0 + j] * gt;
    const float* e = a.xg + ((long long)sd * B + j) * T * D;
    const float* dx = a.dx0 + ((long long)sd * B + j) * 2 * T * D;     // the row's own half: tokens 0..T-1 of its 2T
    float* out = a.dxg + ((long long)sd * B + j) * T * D;
    const float* dS = a.dS + (long long)g * T * D;
    float dot = 0.f;
    for (int i = threadIdx.x; i < T * q; i += 256) {
        const float4 s4 = ld4(dS + 4 * i), e4 = ld4(e + 4 * i), d4 = ld4(dx + 4 * i);
        dot = fmaf(s4.x, e4.x, dot); dot = fmaf(s4.y, e4.y, dot); dot = fmaf(s4.z, e4.z, dot); dot = fmaf(s4.w, e4.w, dot);
        st4(out + 4 * i, make_float4(fmaf(cj, s4.x, d4.x), fmaf(cj, s4.y, d4.y), fmaf(cj, s4.z, d4.z), fmaf(cj, s4.w, d4.w)));
    }
    dot = block_sum256(dot, red);
    float cb = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) cb += a.rows[(g * T + t) * 2 + 1];
    cb = block_sum256(cb, red);
    if (threadIdx.x == 0) a.dw_bs[g][a.j0 + j] = gt * dot + cb;
}

}  // namespace amid

using namespace amid;

static int comp_score(const float* xg, int B, int T, int D, int cross, float* s, void* stream) {
    AMID_CHECK_ARG(xg && s && B > 0 && T > 0 && D > 0 && (D % 4) == 0);
    const size_t lds = (size_t)(cross ? 2 : 1) * T * (D + 4) * sizeof(float);
    if (lds > 160 * 1024 - 256) return AMID_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)inc_score_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    inc_score_kernel<<<dim3(B, 2), 256, lds, (hipStream_t)stream>>>(xg, B, T, D, cross, s);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int comp_tokens_fwd(const float* xg, const float* s, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs,
                           const float* const* b_bs, float threshold, int cross, const float* pos0, const float* pos1, int B, int T, int D,
                           float* gate, float* S, float* Z, float* sw, float* x0, unsigned char* tmq, const void* step_state, int train,
                           float p_drop, void* stream, int Bg = 0, int j0 = 0, int phase = 0) {
    AMID_CHECK_ARG(xg && s && w_nn && b_nn && w_bs && b_bs && gate && S && Z && sw && x0 && (!pos0 == !pos1));
    AMID_CHECK_ARG(B > 0 && T > 0 && D > 0 && (D % 4) == 0 && D <= 256 && (!train || step_state));
    if (Bg == 0) Bg = B;
    AMID_CHECK_ARG(phase >= 0 && phase <= 2 && j0 >= 0 && j0 + B <= Bg);
    IncFwdArgs a;
    a.Bg = Bg; a.j0 = j0; a.phase = phase;
    a.xg = xg; a.s = s; a.threshold = threshold; a.B = B; a.T = T; a.D = D; a.cross = cross ? 1 : 0; a.gate = gate; a.S = S; a.Z = Z; a.sw = sw;
    for (int g = 0; g < 2; ++g) {
        AMID_CHECK_ARG(w_nn[g] && b_nn[g] && w_bs[g] && b_bs[g]);
        a.w_nn[g] = w_nn[g]; a.b_nn[g] = b_nn[g]; a.w_bs[g] = w_bs[g]; a.b_bs[g] = b_bs[g];
    }
    const size_t lds = ((size_t)((B + 3) & ~3) + 9 * (size_t)D) * sizeof(float);
    if (lds > 64 * 1024) return AMID_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    inc_mix_fwd_kernel<<<dim3(T, 2), 256, lds, st>>>(a);
    AMID_LAUNCH_CHECK();
    if (phase == 1) return AMID_OK;
    const int tr = (train && p_drop > 0.f) ? 1 : 0;
    long long blocks = ((long long)4 * B * T + 7) / 8;
    if (blocks > 16384) blocks = 16384;
    inc_embed_fwd_kernel<<<(int)blocks, 256, 0, st>>>(xg, Z, pos0, pos1, B, T, D, x0, tmq, (const RngState*)step_state, tr, keep_thr16(p_drop),
                                                     tr ? 1.0f / (1.0f - p_drop) : 1.0f);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int comp_tokens_bwd(const float* dpos_part, int nsplit, int cross, const float* xg, const float* dx0, const float* gate, const float* S,
                           const float* sw, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B, int T,
                           int D, float* dZ, float* dS, float* rows, float* const* dw_nn, float* const* db_nn, float* const* dw_bs,
                           float* const* db_bs, float* dxg, void* stream, int Bg = 0, int j0 = 0, int phase = 0, float gscale = 1.f) {
    AMID_CHECK_ARG((!dpos_part || nsplit > 0) && xg && dx0 && gate && S && sw && w_nn && b_nn && w_bs && dZ && dS && rows && dw_nn && db_nn &&
                   dw_bs && db_bs && dxg && B > 0 && T > 0 && D > 0 && (D % 4) == 0 && D <= 256);
    if (Bg == 0) Bg = B;
    AMID_CHECK_ARG(phase >= 0 && phase <= 2 && j0 >= 0 && j0 + B <= Bg);
    IncBwdArgs a;
    a.Bg = Bg; a.j0 = j0; a.phase = phase; a.gscale = gscale;
    a.dpos_part = dpos_part; a.nsplit = nsplit; a.cross = cross ? 1 : 0; a.xg = xg; a.dx0 = dx0; a.gate = gate; a.S = S; a.sw = sw;
    a.B = B; a.T = T; a.D = D; a.dZ = dZ; a.dS = dS; a.rows = rows; a.dxg = dxg;
    for (int g = 0; g < 2; ++g) {
        AMID_CHECK_ARG(w_nn[g] && b_nn[g] && w_bs[g] && dw_nn[g] && db_nn[g] && dw_bs[g] && db_bs[g]);
        a.w_nn[g] = w_nn[g]; a.b_nn[g] = b_nn[g]; a.w_bs[g] = w_bs[g];
        a.dw_nn[g] = dw_nn[g]; a.db_nn[g] = db_nn[g]; a.dw_bs[g] = dw_bs[g]; a.db_bs[g] = db_bs[g];
    }
    hipStream_t st = (hipStream_t)stream;
    const int nh = 256 / D > 0 ? 256 / D : 1;
    inc_dz_kernel<<<dim3(T, 2), 256, (size_t)(1 + nh) * D * sizeof(float), st>>>(a);
    AMID_LAUNCH_CHECK();
    if (phase == 1) return AMID_OK;
    if (Bg > B)              // the other shards' entries of d w_bs are theirs to fill: zeros here (the dense exchange sums the ranks)
        for (int g = 0; g < 2; ++g) {
            hipError_t e = hipMemsetAsync(dw_bs[g], 0, (size_t)Bg * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
    inc_wgrad_kernel<<<dim3(D, 2), 256, 0, st>>>(a);
    AMID_LAUNCH_CHECK();
    inc_scatter_bwd_kernel<<<dim3(B, 2), 256, 0, st>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_inc_score_f32(const float* xg, int B, int T, int D, float* s, void* stream) {
    return comp_score(xg, B, T, D, 0, s, stream);
}

extern "C" int amid_inc_embed_fwd_f32(const float* xg, const float* s, const float* const* w_nn, const float* const* b_nn,
                                      const float* const* w_bs, const float* const* b_bs, float threshold, const float* pos0,
                                      const float* pos1, int B, int T, int D, float* gate, float* S, float* Z, float* sw, float* x0,
                                      unsigned char* tmq, const void* step_state, int train, float p_drop, void* stream) {
    AMID_CHECK_ARG(pos0 && pos1 && tmq);
    return comp_tokens_fwd(xg, s, w_nn, b_nn, w_bs, b_bs, threshold, 0, pos0, pos1, B, T, D, gate, S, Z, sw, x0, tmq, step_state, train, p_drop,
                           stream);
}

extern "C" int amid_inc_bwd_f32(const float* dpos_part, int nsplit, const float* xg, const float* dx0, const float* gate, const float* S,
                                const float* sw, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B, int T,
                                int D, float* dZ, float* dS, float* rows, float* const* dw_nn, float* const* db_nn, float* const* dw_bs,
                                float* const* db_bs, float* dxg, void* stream) {
    AMID_CHECK_ARG(dpos_part);
    return comp_tokens_bwd(dpos_part, nsplit, 0, xg, dx0, gate, S, sw, w_nn, b_nn, w_bs, B, T, D, dZ, dS, rows, dw_nn, db_nn, dw_bs, db_bs, dxg,
                           stream);
}

// ---- a data-parallel shard of the batch (B rows = samples j0 .. j0 + B - 1 of a global batch of Bg = len(trans_bs.weight)): the softmax
// over the batch takes the GLOBAL scores s_all [2, Bg] (the ranks' amid_inc_score_f32 outputs, all-gathered per domain), S is summed
// over the ranks between phase 1 (gates, this shard's partial S, sw) and phase 2 (Z, the encoder input); backward: phase 1 = this
// shard's dZ, the caller all-reduces it, phase 2 = dS, the parameter gradients (W_nn, b_nn, b_bs scaled by gscale = 1 / world -- every
// rank computes them alike and the dense exchange sums the ranks --, w_bs's own slice, zeros elsewhere) and the table-row gradients.
extern "C" int amid_inc_embed_fwd_shard_f32(const float* xg, const float* s_all, const float* const* w_nn, const float* const* b_nn,
                                            const float* const* w_bs, const float* const* b_bs, float threshold, const float* pos0,
                                            const float* pos1, int B, int T, int D, int Bg, int j0, int phase, float* gate, float* S, float* Z,
                                            float* sw, float* x0, unsigned char* tmq, const void* step_state, int train, float p_drop,
                                            void* stream) {
    AMID_CHECK_ARG(pos0 && pos1 && tmq && (phase == 1 || phase == 2));
    return comp_tokens_fwd(xg, s_all, w_nn, b_nn, w_bs, b_bs, threshold, 0, pos0, pos1, B, T, D, gate, S, Z, sw, x0, tmq, step_state, train,
                           p_drop, stream, Bg, j0, phase);
}

extern "C" int amid_inc_bwd_shard_f32(const float* dpos_part, int nsplit, const float* xg, const float* dx0, const float* gate, const float* S,
                                      const float* sw, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B,
                                      int T, int D, int Bg, int j0, int phase, float gscale, float* dZ, float* dS, float* rows,
                                      float* const* dw_nn, float* const* db_nn, float* const* dw_bs, float* const* db_bs, float* dxg,
                                      void* stream) {
    AMID_CHECK_ARG(dpos_part && (phase == 1 || phase == 2));
    return comp_tokens_bwd(dpos_part, nsplit, 0, xg, dx0, gate, S, sw, w_nn, b_nn, w_bs, B, T, D, dZ, dS, rows, dw_nn, db_nn, dw_bs, db_bs, dxg,
                           stream, Bg, j0, phase, gscale);
}

// ---- the same token group in front of BERT4Rec's encoders (model_seq.py:283-294): no positional rows, no input dropout ----
extern "C" int amid_bert_comp_score_f32(const float* xg, int B, int T, int D, int cross, float* s, void* stream) {
    return comp_score(xg, B, T, D, cross ? 1 : 0, s, stream);
}

extern "C" int amid_bert_comp_fwd_f32(const float* xg, const float* s, const float* const* w_nn, const float* const* b_nn,
                                      const float* const* w_bs, const float* const* b_bs, float threshold, int cross, int B, int T, int D,
                                      float* gate, float* S, float* Z, float* sw, float* x0, void* stream) {
    return comp_tokens_fwd(xg, s, w_nn, b_nn, w_bs, b_bs, threshold, cross, nullptr, nullptr, B, T, D, gate, S, Z, sw, x0, nullptr, nullptr, 0, 0.f,
                           stream);
}

extern "C" int amid_bert_comp_bwd_f32(const float* xg, const float* dx0, const float* gate, const float* S, const float* sw,
                                      const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int cross, int B, int T,
                                      int D, float* dZ, float* dS, float* rows, float* const* dw_nn, float* const* db_nn, float* const* dw_bs,
                                      float* const* db_bs, float* dxg, void* stream) {
    return comp_tokens_bwd(nullptr, 0, cross, xg, dx0, gate, S, sw, w_nn, b_nn, w_bs, B, T, D, dZ, dS, rows, dw_nn, db_nn, dw_bs, db_bs, dxg,
                           stream);
}

// ... as data-parallel shards (amid_inc_embed_fwd_shard_f32 / amid_inc_bwd_shard_f32 without the positional rows and the input dropout):
// B rows = samples j0 .. j0 + B - 1 of a global batch of Bg; s_all [2, Bg] = the ranks' amid_bert_comp_score_f32 outputs all-gathered per
// domain; forward phase 1 (gates, this shard's partial S, sw) | all-reduce S | phase 2 (Z, the encoder input); backward phase 1 (this
// shard's dZ) | all-reduce dZ | phase 2 (dS, parameter gradients scaled by gscale = 1 / world, w_bs's own slice, table-row gradients).
extern "C" int amid_bert_comp_fwd_shard_f32(const float* xg, const float* s_all, const float* const* w_nn, const float* const* b_nn,
                                            const float* const* w_bs, const float* const* b_bs, float threshold, int cross, int B, int T, int D,
                                            int Bg, int j0, int phase, float* gate, float* S, float* Z, float* sw, float* x0, void* stream) {
    AMID_CHECK_ARG(phase == 1 || phase == 2);
    return comp_tokens_fwd(xg, s_all, w_nn, b_nn, w_bs, b_bs, threshold, cross, nullptr, nullptr, B, T, D, gate, S, Z, sw, x0, nullptr, nullptr, 0, 0.f,
                           stream, Bg, j0, phase);
}

extern "C" int amid_bert_comp_bwd_shard_f32(const float* xg, const float* dx0, const float* gate, const float* S, const float* sw,
                                            const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int cross, int B, int T,
                                            int D, int Bg, int j0, int phase, float gscale, float* dZ, float* dS, float* rows,
                                            float* const* dw_nn, float* const* db_nn, float* const* dw_bs, float* const* db_bs, float* dxg,
                                            void* stream) {
    AMID_CHECK_ARG(phase == 1 || phase == 2);
    return comp_tokens_bwd(nullptr, 0, cross, xg, dx0, gate, S, sw, w_nn, b_nn, w_bs, B, T, D, dZ, dS, rows, dw_nn, db_nn, dw_bs, db_bs, dxg,
                           stream, Bg, j0, phase, gscale);
}
