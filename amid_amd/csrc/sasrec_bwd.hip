// Fused backward kernels of one SASRec encoder layer (autograd of Log2feats.forward model_seq.py:371-383;
// reference: loss.backward(), train_sr.py:214).  Mirrors sasrec_fwd.hip; weights arrive TRANSPOSED
// (wT[in][out], refreshed once per step by amid_transpose_weights) so that every data gradient is again
// C[rows, N] = A[rows, K] * W'[N, K]^T on the same row-tile machinery.
//
//   ffn_bwd : dz = dx' * ~tm ; dpre2 = dz * drop2 ; dpre1 = (dpre2 C2) * relu'(h) ; dy = dpre1 C1 + dz ;
//             dr = LN2'(dy ; r) ; d_o = dr Wo           (+ per-tile partial sums for d gamma2 / d beta2)
//   qkv_bwd : dx = LN1'(dq Wq + dr ; x) + dk Wk + dv Wv  (+ per-tile partials for d gamma1 / d beta1)
//   wgrad   : dW = dY^T X and db = colsum(dY) for the six projections of a layer, split over row
//             ranges; partials are summed in fixed order by amid_reduce_partials (no atomics).
#include "common.h"
#include "rng.h"
#include "tile_gemm.h"
#include "reduce_partials.h"
#include "wgrad_split.h"
#include "wgrad_args.h"
#include "sort_phases.h"

namespace amid {

struct TileGeomB {
    int M; int rows_per_tile; int tiles_per_group;
    // optional hint (the *_rows entry points): of domain g only the sequences b with (row_domain[b] != 0) == g carry a gradient (the
    // step's own loss masks the other domain of every sample, train_sr.py:205-211).  The tiles then walk those sequences' rows only --
    // "virtual" rows v = 0 .. n_live * T - 1 of a domain, live sequences back to back, rows_per_tile of them per tile -- and the
    // launch still has 2 * tiles_per_group workgroups (the worst case: every sample in one domain): the first tiles0 + tiles1 are
    // the live tiles, the rest only zero the LayerNorm partial slot they would have filled.
    const long long* row_domain; int B, T;
};

constexpr int MAP_LIVE_MAX = 1024;                         // live sequences a tile's window may hold (LDS ints); any batch size
constexpr int MAP_INTS = MAP_LIVE_MAX + 128;               // live list + the tile's row map

struct TileRowsB {
    int g, nrows, slot, local0;        // slot: the tile's LayerNorm-partial slot (g * tiles_per_group + index within the domain)
    long long base;                    // g * M
    const int* rmap;                   // hint: LDS table tile row -> row of the domain; nullptr: local0 + r
    bool live;
    __device__ __forceinline__ int lrow(int r) const { return rmap ? rmap[r] : local0 + r; }
    __device__ __forceinline__ long long grow(int r) const { return base + lrow(r); }
};

__device__ __forceinline__ TileRowsB tile_rows_b(const TileGeomB& tg, int tile, int* __restrict__ ints) {
    TileRowsB tr;
    tr.rmap = nullptr; tr.live = true;
    if (tg.row_domain == nullptr) {
        tr.g = tile / tg.tiles_per_group;
        const int tl = tile - tr.g * tg.tiles_per_group;
        tr.local0 = tl * tg.rows_per_tile;
        tr.nrows = min(tg.rows_per_tile, tg.M - tr.local0);
        tr.base = (long long)tr.g * tg.M;
        tr.slot = tile;
        return tr;
    }
    __shared__ int hdr[5];
    int* live = ints;                  // the tile's window of its domain's live batch rows, in batch order
    int* rmap = ints + MAP_LIVE_MAX;
    const int lane = lane_id();
    const int B = tg.B, T = tg.T, rpt = tg.rows_per_tile, tpg = tg.tiles_per_group;
    if (wave_id() == 0) {
        int n0 = 0;
        for (int c = 0; c < B; c += 64) n0 += __popcll(__ballot(c + lane < B && tg.row_domain[c + lane] == 0));
        const int tiles0 = (n0 * T + rpt - 1) / rpt, tiles1 = ((B - n0) * T + rpt - 1) / rpt;
        int g, tl, lv = 1;
        if (tile < tiles0) { g = 0; tl = tile; }
        else if (tile < tiles0 + tiles1) { g = 1; tl = tile - tiles0; }
        else {                                                       // a slot no live tile fills
            lv = 0;
            const int d = tile - tiles0 - tiles1;
            if (d < tpg - tiles0) { g = 0; tl = tiles0 + d; } else { g = 1; tl = tiles1 + d - (tpg - tiles0); }
        }
        const int ng = g == 0 ? n0 : B - n0;
        int sq0 = 0;
        if (lv) {                                                    // the tile's window of the domain's live sequences (<= rpt / T + 2 of them)
            const int v0 = tl * rpt, nr = min(rpt, ng * T - v0);
            sq0 = v0 / T;
            const int sq1 = (v0 + nr - 1) / T;
            int n = 0;
            for (int c = 0; c < B && n <= sq1; c += 64) {
                const int b = c + lane;
                const bool f = b < B && ((tg.row_domain[b] != 0 ? 1 : 0) == g);
                const unsigned long long m = __ballot(f);
                const int k = n + __popcll(m & ((1ull << lane) - 1ull));
                if (f && k >= sq0 && k <= sq1) live[k - sq0] = b;
                n += __popcll(m);
            }
        }
        if (lane == 0) { hdr[0] = g; hdr[1] = tl; hdr[2] = lv; hdr[3] = ng * T; hdr[4] = sq0; }
    }
    __syncthreads();
    tr.g = hdr[0];
    const int tl = hdr[1];
    tr.live = hdr[2] != 0;
    tr.slot = tr.g * tpg + tl;
    tr.base = (long long)tr.g * tg.M;
    tr.local0 = 0;
    tr.nrows = 0;
    if (!tr.live) return tr;
    const int v0 = tl * rpt, sq0 = hdr[4];
    tr.nrows = min(rpt, hdr[3] - v0);
    for (int r = threadIdx.x; r < tr.nrows; r += blockDim.x) {
        const int v = v0 + r, sq = v / T;
        rmap[r] = live[sq - sq0] * T + (v - sq * T);
    }
    __syncthreads();
    tr.rmap = rmap;
    return tr;
}

// the live list / row map sit behind the two operand regions of the row-tile kernels' LDS
template <int D>
__device__ __forceinline__ int* map_ints(float* __restrict__ smem) {
    return reinterpret_cast<int*>(smem + TileCfg<D>::A_FLOATS + TileCfg<D>::W_FLOATS);
}

template <int D>
__device__ __forceinline__ void zero_ln_slot(float* __restrict__ part, int slot) {
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) part[(long long)slot * 2 * D + e] = 0.f;
}

__device__ __forceinline__ float4 apply_tm(float4 v, unsigned bits) {
    if (bits) {
        if (bits & 1u) v.x = 0.f;
        if (bits & 2u) v.y = 0.f;
        if (bits & 4u) v.z = 0.f;
        if (bits & 8u) v.w = 0.f;
    }
    return v;
}

// LayerNorm backward for one row held as a float4 per lane (QPR lanes): given dy, the LN input x and gamma,
// returns dx; accumulates the per-column d gamma / d beta contributions.
template <int QPR>
__device__ __forceinline__ float4 ln_bwd_row(float4 dy, float4 x, float4 gam, int n, float eps, float4& dgam, float4& dbet) {
    float mean, rstd;
    row_stats<QPR>(x, n, eps, mean, rstd);
    const float4 xh = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
    const float4 gy = f4mul(gam, dy);
    const float c1 = group_sum<QPR>(f4hsum(gy)) * (1.0f / n);
    const float c2 = group_sum<QPR>(f4hsum(f4mul(gy, xh))) * (1.0f / n);
    dgam = f4add(dgam, f4mul(dy, xh));
    dbet = f4add(dbet, dy);
    return make_float4(rstd * (gy.x - c1 - xh.x * c2), rstd * (gy.y - c1 - xh.y * c2), rstd * (gy.z - c1 - xh.z * c2),
                       rstd * (gy.w - c1 - xh.w * c2));
}

// block-level reduction of the per-thread (d gamma, d beta) column sums -> part[2][D]
template <int D>
__device__ __forceinline__ void ln_partials_out(float* __restrict__ scratch, float4 dgam, float4 dbet, float* __restrict__ part) {
    using RP = RowPass<D>;
    const int sub = RP::sub(), slot = RP::first_row();            // slot in [0, RPP)
    st4(scratch + (slot * 2 + 0) * D + 4 * sub, dgam);
    st4(scratch + (slot * 2 + 1) * D + 4 * sub, dbet);
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * D; e += GEMM_THREADS) {
        float s = 0.f;
#pragma unroll 4
        for (int k = 0; k < RP::RPP; ++k) s += scratch[k * 2 * D + e];
        part[e] = s;
    }
}

struct FfnBwdArgs {
    const float* dxo;                  // [2M, D] grad of the layer output
    const unsigned char* tmq;
    const float* h; const float* r;    // saved relu output, saved pre-LN2 sum
    const float* ln_w[2];              // LN2 gamma
    const float* w1T[2]; const float* w2T[2]; const float* woT[2];   // [D][D] transposed conv1 / conv2 / out_proj weights
    float* dpre2; float* dpre1; float* dr; float* d_o;               // [2M, D]
    float* ln_part;                    // [ntiles][2][D]
    float ln_eps;
    const StepState* st; int train; unsigned thr16; float scale; int layer;
    TileGeomB tg;
};

// dzin != nullptr (fused behind the next layer's q / k / v backward): the incoming gradient rows arrive in registers
template <int D, bool BF>
__device__ __forceinline__ void ffn_bwd_body(const FfnBwdArgs& a, float* __restrict__ smem, const TileRowsB& tr, const TileRegs<D>* dzin = nullptr) {
    using RP = RowPass<D>;
    constexpr int LDC = D + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<D>::A_FLOATS;
    float* Cs = Ws;
    const int g = tr.g, nrows = tr.nrows;
    auto rowf = [&](int r) { return tr.grow(r); };
    const int sub = RP::sub();
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    TileRegs<D> dz, aux;                                  // dz = dxo * ~tm is needed again for the residual path
    WRegs<D, D> wr;
    if (dzin != nullptr) dz = *dzin; else load_tile_rows<D>(dz, a.dxo, rowf, nrows, D);
    load_w<D, D>(wr, a.w2T[g], D);
    load_tile_rows<D>(aux, a.h, rowf, nrows, D);               // relu output: consumed by the first epilogue
    // 1. dpre2 = (dxo * ~tm) * drop2  -> A image + global
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < TileCfg<D>::ROWS) {
            float4 v = dz.v[i];
            if (r < nrows) {
                if (a.tmq) v = apply_tm(v, a.tmq[tr.grow(r) * (D / 4) + sub]);
                dz.v[i] = v;
                if (a.train) v = f4mul(v, dropout_mult4(seed, site_id(g, a.layer, SITE_FFN2), step, (unsigned long long)tr.lrow(r) * D + 4 * sub,
                                                        a.thr16, a.scale));
                st4(a.dpre2 + tr.grow(r) * D + 4 * sub, v);
            }
            store_a4<D, BF>(As, r, sub, v);
        }
    }
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_w<D, D>(wr, a.w1T[g], D);
    f32x4 acc[WaveMap<D>::ACC];
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
    // 2. dpre1 = dh * relu'(h) * drop1   (h > 0 implies the unit was kept by drop1)
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < TileCfg<D>::ROWS) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrows) {
                const float4 dh = ld4(Cs + r * LDC + 4 * sub), hv = aux.v[i];
                v.x = hv.x > 0.f ? dh.x * a.scale : 0.f; v.y = hv.y > 0.f ? dh.y * a.scale : 0.f;
                v.z = hv.z > 0.f ? dh.z * a.scale : 0.f; v.w = hv.w > 0.f ? dh.w * a.scale : 0.f;
                st4(a.dpre1 + tr.grow(r) * D + 4 * sub, v);
            }
            store_a4<D, BF>(As, r, sub, v);
        }
    }
    __syncthreads();
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_w<D, D>(wr, a.woT[g], D);
    load_tile_rows<D>(aux, a.r, rowf, nrows, D);               // LN2 input rows for the next epilogue
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
    // 3. dy = C + dz ; dr = LN2'(dy ; r) -> global + A image of the out-proj data gradient
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    {
        const float4 gam = ld4(a.ln_w[g] + 4 * sub);
#pragma unroll
        for (int i = 0; i < RP::NR; ++i) {
            const int r = RP::first_row() + i * RP::RPP;
            if (r < TileCfg<D>::ROWS) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrows) {                               // nrows is uniform over the QPR lanes of a row
                    const float4 dy = f4add(ld4(Cs + r * LDC + 4 * sub), dz.v[i]);
                    v = ln_bwd_row<RP::QPR>(dy, aux.v[i], gam, D, a.ln_eps, dgam, dbet);
                    st4(a.dr + tr.grow(r) * D + 4 * sub, v);
                }
                store_a4<D, BF>(As, r, sub, v);
            }
        }
    }
    __syncthreads();
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    acc_to_global_rows<D>(a.d_o, rowf, nrows, D, nullptr, acc);
    __syncthreads();                                      // As is free now: reduction scratch for the LN partials
    ln_partials_out<D>(As, dgam, dbet, a.ln_part + (long long)tr.slot * 2 * D);
}

struct QkvBwdArgs {
    const float* dq; const float* dk; const float* dv;    // [2M, D]
    const float* dr;                                      // residual-path grad of the normed query
    const float* x;                                       // layer input (LN1 input)
    const float* ln_w[2];
    const float* wqT[2]; const float* wkT[2]; const float* wvT[2];
    float* dx;                                            // [2M, D] grad of the layer input
    float* ln_part;                                       // [ntiles][2][D]
    float ln_eps;
    TileGeomB tg;
};

// keep != nullptr: the input-gradient rows are left in registers for a fused successor and NOT stored (nothing else reads them)
template <int D, bool BF>
__device__ __forceinline__ void qkv_bwd_body(const QkvBwdArgs& a, float* __restrict__ smem, const TileRowsB& tr, TileRegs<D>* keep = nullptr) {
    using RP = RowPass<D>;
    constexpr int LDC = D + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<D>::A_FLOATS;
    const int g = tr.g, nrows = tr.nrows;
    auto rowf = [&](int r) { return tr.grow(r); };
    const int sub = RP::sub();
    f32x4 acc_kv[WaveMap<D>::ACC], acc_q[WaveMap<D>::ACC];
    zero_acc<D>(acc_kv);
    zero_acc<D>(acc_q);
    TileRegs<D> ar, xr;
    WRegs<D, D> wr;
    load_tile_rows<D>(ar, a.dk, rowf, nrows, D);
    load_w<D, D>(wr, a.wkT[g], D);
    tile_to_lds<D, BF>(As, ar);
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_tile_rows<D>(ar, a.dv, rowf, nrows, D);               // next operand pair flies under the MFMAs
    load_w<D, D>(wr, a.wvT[g], D);
    mma_tile<D, D, BF>(As, Ws, acc_kv);
    __syncthreads();
    tile_to_lds<D, BF>(As, ar);
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_tile_rows<D>(ar, a.dq, rowf, nrows, D);
    load_w<D, D>(wr, a.wqT[g], D);
    mma_tile<D, D, BF>(As, Ws, acc_kv);
    __syncthreads();
    tile_to_lds<D, BF>(As, ar);
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_tile_rows<D>(ar, a.dr, rowf, nrows, D);               // epilogue inputs: residual-path grad and the LN1 input rows
    load_tile_rows<D>(xr, a.x, rowf, nrows, D);
    mma_tile<D, D, BF>(As, Ws, acc_q);
    __syncthreads();
    float* Ckv = As;                    // both operand images are dead: reuse them as the two C images
    float* Cq = Ws;
    acc_to_lds<D>(Ckv, LDC, acc_kv);
    acc_to_lds<D>(Cq, LDC, acc_q);
    __syncthreads();
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    const float4 gam = ld4(a.ln_w[g] + 4 * sub);
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        float4 dxv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows) {
            const float4 dqn = f4add(ld4(Cq + r * LDC + 4 * sub), ar.v[i]);
            const float4 dxl = ln_bwd_row<RP::QPR>(dqn, xr.v[i], gam, D, a.ln_eps, dgam, dbet);
            dxv = f4add(dxl, ld4(Ckv + r * LDC + 4 * sub));
            if (keep == nullptr) st4(a.dx + tr.grow(r) * D + 4 * sub, dxv);
        }
        if (keep != nullptr) keep->v[i] = dxv;
    }
    __syncthreads();
    ln_partials_out<D>(Ws, dgam, dbet, a.ln_part + (long long)tr.slot * 2 * D);
}

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_ffn_bwd_kernel(const FfnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileRowsB tr = tile_rows_b(a.tg, blockIdx.x, map_ints<D>(smem));
    if (!tr.live) { zero_ln_slot<D>(a.ln_part, tr.slot); return; }
    ffn_bwd_body<D, BF>(a, smem, tr);
}

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_qkv_bwd_kernel(const QkvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileRowsB tr = tile_rows_b(a.tg, blockIdx.x, map_ints<D>(smem));
    if (!tr.live) { zero_ln_slot<D>(a.ln_part, tr.slot); return; }
    qkv_bwd_body<D, BF>(a, smem, tr);
}

// layer l + 1's q / k / v + LayerNorm1 backward followed by layer l's feed-forward / out-projection backward on the same row
// tile, one launch: the tile's rows of d x[l + 1] pass between the two bodies in registers (no HBM copy at all)
struct QkvFfnBwdArgs { QkvBwdArgs qkv; FfnBwdArgs ffn; TileGeomB tg; };

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_qkv_ffn_bwd_kernel(const QkvFfnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TileRowsB tr = tile_rows_b(a.tg, blockIdx.x, map_ints<D>(smem));
    if (!tr.live) { zero_ln_slot<D>(a.qkv.ln_part, tr.slot); zero_ln_slot<D>(a.ffn.ln_part, tr.slot); return; }
    TileRegs<D> dxr;            // d x[l + 1] of the tile goes from one body to the other in registers and never to HBM
    qkv_bwd_body<D, BF>(a.qkv, smem, tr, &dxr);
    __syncthreads();            // the first body's LDS scratch (LayerNorm partials) is read; the second restages both regions
    ffn_bwd_body<D, BF>(a.ffn, smem, tr, &dxr);
}

// ---------------------------------------------------------------------------------------------
// weight gradients of up to two layers in one launch: blockIdx = (split, layer * 6 + weight 0..5, domain)
// ---------------------------------------------------------------------------------------------
// (WG_MAX, WgradArgs: wgrad_args.h)

constexpr int WG_ROWS = 64;    // rows staged per step
// diagnostic builds only (profiles/tools/wgrad_stamps.py compiles this file with -DAMID_WGRAD_STAMPS into its own library): real-time
// (100 MHz) stamps of workgroup 0's phases and every workgroup's start / end, in buffers no kernel reads
#if defined(AMID_WGRAD_STAMPS) && AMID_TILE_RT == 7
static __device__ unsigned long long amid_wgrad_stamp_buf[64];
static __device__ unsigned long long amid_wgrad_sched_buf[1024 * 2];     // per workgroup: start, end
#define WG_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && (i) < 64) amid_wgrad_stamp_buf[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WG_SCHED(slot) do { const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && wg_ < 1024) amid_wgrad_sched_buf[wg_ * 2 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#define WG_SCHED(slot) do { } while (0)
#endif
// (WG_LIVE_MAX: wgrad_args.h)

// One workgroup = one (domain, weight, row split): it streams its rows of (dY, X) in 64-row chunks through LDS and
// accumulates dW = dY^T X on the matrix cores (one wave per 16 output rows at D = 128).  Single LDS buffer (74 KB at
// D = 128) and <= 128 VGPRs, so that TWO workgroups share a CU: while one stores its next chunk and waits at its barriers
// the other issues MFMAs.  (History: one workgroup per CU with a double-buffered 147 KB image and two chunks of register
// prefetch measured, by s_memtime, 10 % in the prologue -- all workgroups requesting their first chunks at once --, 8 % in
// stores + barriers and 40 cycles per MFMA against the 32-cycle issue rate; with both layers' 504 workgroups in one launch,
// two per CU, those phases of one workgroup hide behind the other's.)
template <int D>
__global__ __launch_bounds__(GEMM_THREADS, 2) void sas_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = D + 16;                         // (4*LD) % 32 words == 16: the four m-rows of an MFMA hit disjoint banks
    constexpr int NTn = D / 16;
    constexpr int WPN = 8 / NTn > 0 ? 8 / NTn : 1;     // waves per n tile (1 at D=128, 2 at D=64)
    constexpr int KTW = NTn / WPN;                     // k tiles per wave (8 at D=128, 2 at D=64)
    float* Ys = smem;
    float* Xs = smem + WG_ROWS * LD;
    int* live = reinterpret_cast<int*>(smem + 2 * WG_ROWS * LD);       // [B] batch rows of this domain's live sequences (hint only)
    WG_STAMP(0);
    WG_SCHED(0);
    const int split = blockIdx.x, wsel = blockIdx.y, g = blockIdx.z;
    const int layer = wsel / 6, wi = wsel - layer * 6;
    const float* __restrict__ dy = a.dy[wsel];
    const float* __restrict__ xin = a.xin[wsel];
    const int w = wave_id(), lane = lane_id();
    const bool hint = a.row_domain != nullptr;
    int local_beg = split * a.rows_per_split;
    int local_end = min(a.M, local_beg + a.rows_per_split);
    int sq0 = 0;
    if (hint) {                // wave 0 counts the domain's live sequences, then lists the window of them this split walks
        __shared__ int hd[3];
        if (w == 0) {
            int nl = 0;
            for (int c = 0; c < a.B; c += 64) nl += __popcll(__ballot(c + lane < a.B && ((a.row_domain[c + lane] != 0 ? 1 : 0) == g)));
            const int mv = nl * a.T, rps = (mv + a.splits - 1) / a.splits;
            const int lb = min(mv, split * rps), le = min(mv, lb + rps);
            const int s0 = lb / a.T, s1 = le > lb ? (le - 1) / a.T : s0 - 1;
            int n = 0;
            for (int c = 0; c < a.B && n <= s1; c += 64) {
                const int b = c + lane;
                const bool f = b < a.B && ((a.row_domain[b] != 0 ? 1 : 0) == g);
                const unsigned long long m = __ballot(f);
                const int k = n + __popcll(m & ((1ull << lane) - 1ull));
                if (f && k >= s0 && k <= s1) live[k - s0] = b;
                n += __popcll(m);
            }
            if (lane == 0) { hd[0] = lb; hd[1] = le; hd[2] = s0; }
        }
        __syncthreads();
        local_beg = hd[0]; local_end = hd[1]; sq0 = hd[2];
    }
    WG_STAMP(1);
    const int nt = w / WPN, kt0 = (w % WPN) * KTW;
    const int i = lane & 15, gq = lane >> 4;
    constexpr int QPR = D / 4, RPP = GEMM_THREADS / QPR;
    const int sub = threadIdx.x % QPR, rl = threadIdx.x / QPR;
    f32x4 acc[KTW];
#pragma unroll
    for (int t = 0; t < KTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int NRW = WG_ROWS / RPP;                 // rows of a chunk per thread (4 at D=128, 2 at D=64)
    float4 py[NRW], px[NRW];
    auto fetch = [&](int c0) {                         // issue every load of chunk c0 (zeros beyond the split's range)
        const int nr = min(WG_ROWS, local_end - c0);
        const long long grow = (long long)g * a.M + c0;
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int r = rl + i * RPP;
            const bool ok = r < nr;
            long long row = grow + r;
            if (hint && ok) {                          // virtual row -> (live sequence, position) -> row of the domain
                const int v = c0 + r, sq = v / a.T;
                row = (long long)g * a.M + (long long)live[sq - sq0] * a.T + (v - sq * a.T);
            }
            py[i] = ok ? ld4(dy + row * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            px[i] = ok ? ld4(xin + row * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (local_beg < local_end) fetch(local_beg);
    WG_STAMP(2);
    for (int c0 = local_beg; c0 < local_end; c0 += WG_ROWS) {
        __syncthreads();                               // previous chunk fully consumed
#pragma unroll
        for (int i = 0; i < NRW; ++i) {
            const int r = rl + i * RPP;
            bsum = f4add(bsum, py[i]);
            st4(Ys + r * LD + 4 * sub, py[i]);
            st4(Xs + r * LD + 4 * sub, px[i]);
        }
        __syncthreads();
        if (c0 + WG_ROWS < local_end) fetch(c0 + WG_ROWS);     // next chunk flies under this chunk's MFMAs
        // all 16 m-steps, always (rows past the split's range are zeros in LDS): branch-free, with the operands of step
        // ms + 1 read while the MFMAs of step ms issue
        const float* yp = Ys + gq * LD + nt * 16 + i;
        const float* xp = Xs + gq * LD + kt0 * 16 + i;
        float a_cur = yp[0], x_cur[KTW];
#pragma unroll
        for (int t = 0; t < KTW; ++t) x_cur[t] = xp[t * 16];
#pragma unroll
        for (int ms = 0; ms < WG_ROWS / 4; ++ms) {
            float a_nxt = a_cur, x_nxt[KTW];
#pragma unroll
            for (int t = 0; t < KTW; ++t) x_nxt[t] = x_cur[t];
            if (ms + 1 < WG_ROWS / 4) {
                a_nxt = yp[(ms + 1) * 4 * LD];
#pragma unroll
                for (int t = 0; t < KTW; ++t) x_nxt[t] = xp[(ms + 1) * 4 * LD + t * 16];
            }
#pragma unroll
            for (int t = 0; t < KTW; ++t) acc[t] = mfma16(a_cur, x_cur[t], acc[t]);
            // pin the issue order: one LDS read of the next step behind every MFMA of this one (left alone, the scheduler
            // sinks each read to just before its first use and waits on it)
#pragma unroll
            for (int t = 0; t < KTW; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < KTW; ++t) x_cur[t] = x_nxt[t];
        }
    }
    WG_STAMP(40);
    float* wp = a.w_part[layer] + (((long long)g * 6 + wi) * a.splits + split) * D * D;
    if constexpr (D == 128) {
        // the wave's 16 x 128 block of the partial goes through LDS (its own 16 x 132 floats: no barrier beyond the one that frees the
        // chunk image) and leaves as whole 512-byte rows, two per store instruction, instead of four 64-byte pieces (step 0.3703 -> 0.3690 ms)
        __syncthreads();
        constexpr int LDP = D + 4;
        float* mine = smem + w * 16 * LDP;
#pragma unroll
        for (int t = 0; t < KTW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(gq * 4 + r) * LDP + (kt0 + t) * 16 + i] = acc[t][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q4 = lane + 64 * k, row = q4 >> 5, c4 = q4 & 31;
            st4(wp + (long long)(nt * 16 + row) * D + 4 * c4, ld4(mine + row * LDP + 4 * c4));
        }
    } else {
#pragma unroll
    for (int t = 0; t < KTW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[(long long)(nt * 16 + gq * 4 + r) * D + (kt0 + t) * 16 + i] = acc[t][r];
    }
    __syncthreads();
    st4(Ys + rl * D + 4 * sub, bsum);                  // [RPP][D] scratch
    __syncthreads();
    float* bp = a.b_part[layer] + (((long long)g * 6 + wi) * a.splits + split) * D;
    for (int e = threadIdx.x; e < D; e += GEMM_THREADS) {
        float s = 0.f;
#pragma unroll 4
        for (int k = 0; k < RPP; ++k) s += Ys[k * D + e];
        bp[e] = s;
    }
    WG_STAMP(41);
    WG_SCHED(1);
}

// The same weight gradients with bf16 operands (compute = "bf16": BASELINE.json configs[2]; fp32 accumulation, fp32 partial sums, fp32
// bias sums from the unrounded rows): dW = dY^T X contracts over ROWS, and v_mfma_f32_16x16x32_bf16 wants eight consecutive k per lane,
// so a chunk is staged TRANSPOSED -- Yt / Xt [column][64 rows] bf16, 144 bytes per column (16-byte chunks of eight rows, chunk c of column
// col at position c ^ ((col >> 2) & 7): a thread packs its four consecutive rows of a column into one 8-byte write, an operand fragment
// is one 16-byte read).  16 matrix instructions per wave and chunk instead of 128: the launch is then bound by its 157 MB of operands.
constexpr int WG16_COL_BYTES = 144;                    // 64 rows x 2 bytes + 16: a column of the transposed image

template <int D>
__global__ __launch_bounds__(GEMM_THREADS, 2) void sas_wgrad16_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(D == 128, "eight waves = eight 16-row tiles of dW");
    constexpr int NTn = D / 16;
    char* const Yt = reinterpret_cast<char*>(smem);
    char* const Xt = Yt + D * WG16_COL_BYTES;
    float* const scratch = reinterpret_cast<float*>(Xt + D * WG16_COL_BYTES);             // [16][D] bias-sum scratch, then the live window
    int* live = reinterpret_cast<int*>(scratch + 16 * D);
    const int split = blockIdx.x, wsel = blockIdx.y, g = blockIdx.z;
    const int layer = wsel / 6, wi = wsel - layer * 6;
    const float* __restrict__ dy = a.dy[wsel];
    const float* __restrict__ xin = a.xin[wsel];
    const int w = wave_id(), lane = lane_id();
    const bool hint = a.row_domain != nullptr;
    int local_beg = split * a.rows_per_split;
    int local_end = min(a.M, local_beg + a.rows_per_split);
    int sq0 = 0;
    if (hint) {                // wave 0 counts the domain's live sequences, then lists the window of them this split walks (as sas_wgrad_kernel)
        __shared__ int hd[3];
        if (w == 0) {
            int nl = 0;
            for (int c = 0; c < a.B; c += 64) nl += __popcll(__ballot(c + lane < a.B && ((a.row_domain[c + lane] != 0 ? 1 : 0) == g)));
            const int mv = nl * a.T, rps = (mv + a.splits - 1) / a.splits;
            const int lb = min(mv, split * rps), le = min(mv, lb + rps);
            const int s0 = lb / a.T, s1 = le > lb ? (le - 1) / a.T : s0 - 1;
            int n = 0;
            for (int c = 0; c < a.B && n <= s1; c += 64) {
                const int b = c + lane;
                const bool f = b < a.B && ((a.row_domain[b] != 0 ? 1 : 0) == g);
                const unsigned long long m = __ballot(f);
                const int k = n + __popcll(m & ((1ull << lane) - 1ull));
                if (f && k >= s0 && k <= s1) live[k - s0] = b;
                n += __popcll(m);
            }
            if (lane == 0) { hd[0] = lb; hd[1] = le; hd[2] = s0; }
        }
        __syncthreads();
        local_beg = hd[0]; local_end = hd[1]; sq0 = hd[2];
    }
    const int i = lane & 15, gq = lane >> 4;
    const int sub = threadIdx.x & 31, rl = threadIdx.x >> 5;       // column quad 0..31; rows 4 rl .. 4 rl + 3 of the chunk
    f32x4 acc[NTn];
#pragma unroll
    for (int t = 0; t < NTn; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 py[4], px[4];
    auto fetch = [&](int c0) {                         // issue every load of chunk c0 (zeros beyond the split's range)
        const int nr = min(WG_ROWS, local_end - c0);
        const long long grow = (long long)g * a.M + c0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 4 * rl + k;
            const bool ok = r < nr;
            long long row = grow + r;
            if (hint && ok) {                          // virtual row -> (live sequence, position) -> row of the domain
                const int v = c0 + r, sq = v / a.T;
                row = (long long)g * a.M + (long long)live[sq - sq0] * a.T + (v - sq * a.T);
            }
            py[k] = ok ? ld4(dy + row * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            px[k] = ok ? ld4(xin + row * D + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // this thread's 8-byte slot in column 4 sub + j: rows 4 rl .. 4 rl + 3 = half (rl & 1) of chunk rl >> 1
    const int wr_off = (((rl >> 1) ^ (sub & 7)) << 4) + ((rl & 1) << 3);
    if (local_beg < local_end) fetch(local_beg);
    for (int c0 = local_beg; c0 < local_end; c0 += WG_ROWS) {
        __syncthreads();                               // previous chunk fully consumed
#pragma unroll
        for (int k = 0; k < 4; ++k) bsum = f4add(bsum, py[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 4 * sub + j;
            const float y0 = f4comp_w(py[0], j), y1 = f4comp_w(py[1], j), y2 = f4comp_w(py[2], j), y3 = f4comp_w(py[3], j);
            const float x0 = f4comp_w(px[0], j), x1 = f4comp_w(px[1], j), x2 = f4comp_w(px[2], j), x3 = f4comp_w(px[3], j);
            *reinterpret_cast<uint2*>(Yt + col * WG16_COL_BYTES + wr_off) = make_uint2(wg_pack2(y0, y1), wg_pack2(y2, y3));
            *reinterpret_cast<uint2*>(Xt + col * WG16_COL_BYTES + wr_off) = make_uint2(wg_pack2(x0, x1), wg_pack2(x2, x3));
        }
        __syncthreads();
        if (c0 + WG_ROWS < local_end) fetch(c0 + WG_ROWS);     // next chunk flies under this chunk's MFMAs
#pragma unroll
        for (int s = 0; s < WG_ROWS / 32; ++s) {       // two k-steps of 32 rows: lane group gq supplies rows 32 s + 8 gq .. + 7
            const int ycol = w * 16 + i;
            const wg_v4u af = *reinterpret_cast<const wg_v4u*>(Yt + ycol * WG16_COL_BYTES + (((4 * s + gq) ^ ((ycol >> 2) & 7)) << 4));
#pragma unroll
            for (int t = 0; t < NTn; ++t) {
                const int xcol = t * 16 + i;
                const wg_v4u bf = *reinterpret_cast<const wg_v4u*>(Xt + xcol * WG16_COL_BYTES + (((4 * s + gq) ^ ((xcol >> 2) & 7)) << 4));
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, af), __builtin_bit_cast(wg_bf16x8, bf), acc[t], 0, 0, 0);
            }
        }
    }
    float* wp = a.w_part[layer] + (((long long)g * 6 + wi) * a.splits + split) * D * D;
#pragma unroll
    for (int t = 0; t < NTn; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[(long long)(w * 16 + gq * 4 + r) * D + t * 16 + i] = acc[t][r];
    __syncthreads();
    st4(scratch + rl * D + 4 * sub, bsum);             // [16][D]
    __syncthreads();
    float* bp = a.b_part[layer] + (((long long)g * 6 + wi) * a.splits + split) * D;
    for (int e = threadIdx.x; e < D; e += GEMM_THREADS) {
        float sum = 0.f;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) sum += scratch[k * D + e];
        bp[e] = sum;
    }
}

// (sas_wgrad_split_kernel -- the same on the bf16 matrix cores at fp32 accuracy, mma mode 2 / 3 -- lives in sasrec_wgrad_split.hip: built
// without the SLP vectorizer, DESIGN.md section 5.0 "The wgrad finding"; this file keeps the default flags)

// ---------------------------------------------------------------------------------------------
// out[j][i] = in[i][j] for a batch of square D x D matrices (per-step refresh of the transposed weights)
// ---------------------------------------------------------------------------------------------
struct TransposeArgs { const float* src[32]; float* dst[32]; int n; };

__global__ __launch_bounds__(256) void transpose_sq_kernel(const TransposeArgs a, int D) {
    __shared__ float tile[32][33];
    const float* __restrict__ s = a.src[blockIdx.z];
    float* __restrict__ d = a.dst[blockIdx.z];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8) tile[r][tx] = s[(long long)(by + r) * D + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) d[(long long)(bx + r) * D + by + tx] = tile[tx][r];
}

// ---------------------------------------------------------------------------------------------
// fixed-order sum of partial buffers: dst[e] = sum_k src[k * stride + e]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reduce_partials_kernel(const ReduceEntry* __restrict__ entries) {
    reduce_partials_block(entries[blockIdx.y], blockIdx.x, gridDim.x);
}

}  // namespace amid

using namespace amid;

template <int D> static constexpr size_t fused_lds_bytes_b() {
    return (size_t)(TileCfg<D>::A_FLOATS + TileCfg<D>::W_FLOATS) * sizeof(float) + (size_t)MAP_INTS * sizeof(int);
}

static int make_geom_b(int M, int rows_per_tile, TileGeomB* tg, const long long* row_domain = nullptr, int B = 0, int T = 0) {
    if (M <= 0 || rows_per_tile <= 0 || rows_per_tile > TILE_ROWS) return AMID_ERR_ARG;
    tg->M = M;
    tg->rows_per_tile = rows_per_tile;
    tg->tiles_per_group = (M + rows_per_tile - 1) / rows_per_tile;
    tg->row_domain = nullptr; tg->B = B; tg->T = T;
    if (row_domain) {
        if (B <= 0 || T <= 0 || (long long)B * T != M || rows_per_tile / T + 2 > MAP_LIVE_MAX) return AMID_ERR_ARG;
        tg->row_domain = row_domain;
    }
    return AMID_OK;
}

#define AMID_LAUNCH_FUSED_B(KERNEL, ARGS, DVAL, BFVAL)                                                                        \
    do {                                                                                                                   \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL<DVAL, BFVAL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)fused_lds_bytes_b<DVAL>());                                              \
            if (e != hipSuccess) return (int)e;                                                                            \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        KERNEL<DVAL, BFVAL><<<2 * ARGS.tg.tiles_per_group, GEMM_THREADS, fused_lds_bytes_b<DVAL>(), (hipStream_t)stream>>>(ARGS); \
    } while (0)

static int amid_sas_ffn_bwd_impl(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                                    const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int M, int D,
                                    int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* dpre2,
                                    float* dpre1, float* dr, float* d_o, float* ln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream) {
    AMID_CHECK_ARG(dxo && h && r && ln_w && w1T && w2T && woT && dpre2 && dpre1 && dr && d_o && ln_part && (!train || step_state));
    FfnBwdArgs a;
    a.dxo = dxo; a.tmq = tmq; a.h = h; a.r = r; a.dpre2 = dpre2; a.dpre1 = dpre1; a.dr = dr; a.d_o = d_o; a.ln_part = ln_part;
    a.ln_eps = ln_eps; a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.thr16 = keep_thr16(p_drop);
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) { a.ln_w[g] = ln_w[g]; a.w1T[g] = w1T[g]; a.w2T[g] = w2T[g]; a.woT[g] = woT[g]; }
    if (int e = make_geom_b(M, rows_per_tile, &a.tg, row_domain, B, T)) return e;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED_B(sas_ffn_bwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED_B(sas_ffn_bwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED_B(sas_ffn_bwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_ffn_bwd_f32)(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                                    const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int M, int D,
                                    int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* dpre2,
                                    float* dpre1, float* dr, float* d_o, float* ln_part, int mma_bf16, void* stream) {
    return amid_sas_ffn_bwd_impl(dxo, tmq, h, r, ln_w, w1T, w2T, woT, ln_eps, M, D, rows_per_tile, layer, step_state, train, p_drop, dpre2, dpre1, dr, d_o, ln_part, mma_bf16, nullptr, 0, 0, stream);
}

// amid_sas_ffn_bwd_f32 over the live sequences only (TileGeomB::row_domain): M = B * T, row_domain [B] = the batch's domain ids; rows_per_tile
// counts live rows, ln_part holds 2 * ceil(M / rows_per_tile) slots
extern "C" int AMID_ENTRY(amid_sas_ffn_bwd_rows_f32)(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                                    const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int M, int D,
                                    int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* dpre2,
                                    float* dpre1, float* dr, float* d_o, float* ln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream) {
    AMID_CHECK_ARG(row_domain != nullptr);
    return amid_sas_ffn_bwd_impl(dxo, tmq, h, r, ln_w, w1T, w2T, woT, ln_eps, M, D, rows_per_tile, layer, step_state, train, p_drop, dpre2, dpre1, dr, d_o, ln_part, mma_bf16, row_domain, B, T, stream);
}

static int amid_sas_qkv_bwd_impl(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                    const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                    float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream) {
    AMID_CHECK_ARG(dq && dk && dv && dr && x && ln_w && wqT && wkT && wvT && dx && ln_part);
    QkvBwdArgs a;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dr = dr; a.x = x; a.dx = dx; a.ln_part = ln_part; a.ln_eps = ln_eps;
    for (int g = 0; g < 2; ++g) { a.ln_w[g] = ln_w[g]; a.wqT[g] = wqT[g]; a.wkT[g] = wkT[g]; a.wvT[g] = wvT[g]; }
    if (int e = make_geom_b(M, rows_per_tile, &a.tg, row_domain, B, T)) return e;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED_B(sas_qkv_bwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED_B(sas_qkv_bwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED_B(sas_qkv_bwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_qkv_bwd_f32)(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                    const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                    float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part, int mma_bf16, void* stream) {
    return amid_sas_qkv_bwd_impl(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, M, D, rows_per_tile, dx, ln_part, mma_bf16, nullptr, 0, 0, stream);
}

// amid_sas_qkv_bwd_f32 over the live sequences only (TileGeomB::row_domain): M = B * T, row_domain [B] = the batch's domain ids; rows_per_tile
// counts live rows, ln_part holds 2 * ceil(M / rows_per_tile) slots
extern "C" int AMID_ENTRY(amid_sas_qkv_bwd_rows_f32)(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                    const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                    float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream) {
    AMID_CHECK_ARG(row_domain != nullptr);
    return amid_sas_qkv_bwd_impl(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, M, D, rows_per_tile, dx, ln_part, mma_bf16, row_domain, B, T, stream);
}

// amid_sas_qkv_bwd_f32 of layer l + 1 followed by amid_sas_ffn_bwd_f32 of layer l (f* arguments; its dxo is the dx just produced)
static int amid_sas_qkv_ffn_bwd_impl(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                        const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                        float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part,
                                        const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w,
                                        const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer,
                                        const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr, float* fd_o,
                                        float* fln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream) {
    AMID_CHECK_ARG(dq && dk && dv && dr && x && ln_w && wqT && wkT && wvT && dx && ln_part);
    AMID_CHECK_ARG(fh && fr && fln_w && fw1T && fw2T && fwoT && fdpre2 && fdpre1 && fdr && fd_o && fln_part && (!train || step_state));
    QkvFfnBwdArgs a;
    a.qkv.dq = dq; a.qkv.dk = dk; a.qkv.dv = dv; a.qkv.dr = dr; a.qkv.x = x; a.qkv.dx = dx; a.qkv.ln_part = ln_part; a.qkv.ln_eps = ln_eps;
    a.ffn.dxo = dx; a.ffn.tmq = tmq; a.ffn.h = fh; a.ffn.r = fr; a.ffn.dpre2 = fdpre2; a.ffn.dpre1 = fdpre1; a.ffn.dr = fdr; a.ffn.d_o = fd_o;
    a.ffn.ln_part = fln_part; a.ffn.ln_eps = ln_eps; a.ffn.st = (const StepState*)step_state; a.ffn.layer = flayer;
    a.ffn.train = (train && p_drop > 0.f) ? 1 : 0;
    a.ffn.thr16 = keep_thr16(p_drop);
    a.ffn.scale = a.ffn.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) {
        a.qkv.ln_w[g] = ln_w[g]; a.qkv.wqT[g] = wqT[g]; a.qkv.wkT[g] = wkT[g]; a.qkv.wvT[g] = wvT[g];
        a.ffn.ln_w[g] = fln_w[g]; a.ffn.w1T[g] = fw1T[g]; a.ffn.w2T[g] = fw2T[g]; a.ffn.woT[g] = fwoT[g];
    }
    if (int e = make_geom_b(M, rows_per_tile, &a.tg, row_domain, B, T)) return e;
    a.qkv.tg = a.tg; a.ffn.tg = a.tg;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED_B(sas_qkv_ffn_bwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED_B(sas_qkv_ffn_bwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED_B(sas_qkv_ffn_bwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_qkv_ffn_bwd_f32)(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                        const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                        float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part,
                                        const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w,
                                        const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer,
                                        const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr, float* fd_o,
                                        float* fln_part, int mma_bf16, void* stream) {
    return amid_sas_qkv_ffn_bwd_impl(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, M, D, rows_per_tile, dx, ln_part, tmq, fh, fr, fln_w, fw1T, fw2T, fwoT, flayer, step_state, train, p_drop, fdpre2, fdpre1, fdr, fd_o, fln_part, mma_bf16, nullptr, 0, 0, stream);
}

// amid_sas_qkv_ffn_bwd_f32 over the live sequences only (TileGeomB::row_domain): M = B * T, row_domain [B] = the batch's domain ids; rows_per_tile
// counts live rows, ln_part holds 2 * ceil(M / rows_per_tile) slots
extern "C" int AMID_ENTRY(amid_sas_qkv_ffn_bwd_rows_f32)(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                        const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                        float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part,
                                        const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w,
                                        const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer,
                                        const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr, float* fd_o,
                                        float* fln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream) {
    AMID_CHECK_ARG(row_domain != nullptr);
    return amid_sas_qkv_ffn_bwd_impl(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, M, D, rows_per_tile, dx, ln_part, tmq, fh, fr, fln_w, fw1T, fw2T, fwoT, flayer, step_state, train, p_drop, fdpre2, fdpre1, fdr, fd_o, fln_part, mma_bf16, row_domain, B, T, stream);
}

#if AMID_TILE_RT == 7      // everything below is independent of the row-tile height: one copy only
static int sas_wgrad(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits, float* const* w_part,
                     float* const* b_part, const long long* row_domain, int B, int T, int mma_bf16, void* stream,
                     const void* sort_plan = nullptr, const float* const* ln_stat = nullptr, const float* const* ln1_w = nullptr,
                     const float* const* ln1_b = nullptr, const float* const* ln2_w = nullptr, const float* const* ln2_b = nullptr) {
    AMID_CHECK_ARG(dy && x && w_part && b_part && (n_layers == 1 || n_layers == 2) && M > 0 && splits > 0);
    AMID_CHECK_ARG(!row_domain || (B > 0 && T > 0 && (long long)B * T == M));
    WgradArgs a = {};
    if (ln_stat != nullptr) {          // (the six-pair build with the rider: checked below)
        AMID_CHECK_ARG(ln1_w && ln1_b && ln2_w && ln2_b && sort_plan != nullptr);
        for (int l = 0; l < n_layers; ++l) {
            AMID_CHECK_ARG(ln_stat[l] != nullptr);
            a.ln_stat[l] = ln_stat[l];
            for (int g = 0; g < 2; ++g) {
                AMID_CHECK_ARG(ln1_w[2 * l + g] && ln1_b[2 * l + g] && ln2_w[2 * l + g] && ln2_b[2 * l + g]);
                a.ln1_w[l][g] = ln1_w[2 * l + g]; a.ln1_b[l][g] = ln1_b[2 * l + g]; a.ln2_w[l][g] = ln2_w[2 * l + g]; a.ln2_b[l][g] = ln2_b[2 * l + g];
            }
        }
    }
    for (int i = 0; i < 6 * n_layers; ++i) { AMID_CHECK_ARG(dy[i] && x[i]); a.dy[i] = dy[i]; a.xin[i] = x[i]; }
    for (int l = 0; l < n_layers; ++l) { AMID_CHECK_ARG(w_part[l] && b_part[l]); a.w_part[l] = w_part[l]; a.b_part[l] = b_part[l]; }
    a.M = M; a.splits = splits;
    a.rows_per_split = (M + splits - 1) / splits;
    // the live list lives in LDS: beyond WG_LIVE_MAX sequences the hint is dropped (every row is walked; same result)
    // a split's window of live sequences lives in LDS: rows_per_split / T + 2 entries at most (any batch size)
    const int win = row_domain ? a.rows_per_split / T + 2 : 0;
    a.row_domain = (row_domain && win <= WG_LIVE_MAX) ? row_domain : nullptr; a.B = B; a.T = T;
    const size_t live_bytes = a.row_domain ? (size_t)((win + 3) & ~3) * sizeof(int) : 0;
    const dim3 grid(splits, 6 * n_layers, 2);
    if (sort_plan != nullptr) {      // the last phase of a sort plan rides in a third z-slice: the six-pair build with the live-row hint only
        SortRider rd;
        rd.plan = *(const SortPlan*)sort_plan;
        rd.phase = 5;
        // (mma_bf16 = 4: the same launch on ONE piece per operand -- bf16 products, the folded bf16 step)
        if (!((mma_bf16 == 3 || mma_bf16 == 4) && D == 128 && a.row_domain != nullptr && splits * 6 * n_layers >= rd.plan.nblk)) return AMID_ERR_UNSUPPORTED;
        return launch_sas_wgrad_split(a, &rd, n_layers, mma_bf16, live_bytes, stream);
    }
    if (mma_bf16 >= 2) {      // fp32 operands as three bf16 pieces each (sasrec_wgrad_split.hip): D = 128 only
        if (D != 128) return AMID_ERR_UNSUPPORTED;
        if (mma_bf16 > 3) return AMID_ERR_ARG;
        return launch_sas_wgrad_split(a, nullptr, n_layers, mma_bf16, live_bytes, stream);
    }
    if (mma_bf16) {                 // bf16 operands (csrc: sas_wgrad16_kernel): D = 128 only
        if (D != 128) return AMID_ERR_UNSUPPORTED;
        const size_t lds = (size_t)2 * 128 * WG16_COL_BYTES + (size_t)16 * 128 * sizeof(float) + live_bytes;
        sas_wgrad16_kernel<128><<<grid, GEMM_THREADS, lds, (hipStream_t)stream>>>(a);
        AMID_LAUNCH_CHECK();
        return AMID_OK;
    }
    if (D == 128) {
        const size_t lds = (size_t)2 * WG_ROWS * (128 + 16) * sizeof(float) + live_bytes;
        static size_t set128 = 0;
        const size_t want = (size_t)2 * WG_ROWS * (128 + 16) * sizeof(float) + WG_LIVE_MAX * sizeof(int);
        if (set128 < want) { hipError_t e = hipFuncSetAttribute((const void*)sas_wgrad_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want); if (e != hipSuccess) return (int)e; set128 = want; }
        sas_wgrad_kernel<128><<<grid, GEMM_THREADS, lds, (hipStream_t)stream>>>(a);
    } else if (D == 64) {
        const size_t lds = (size_t)2 * WG_ROWS * (64 + 16) * sizeof(float) + live_bytes;
        sas_wgrad_kernel<64><<<grid, GEMM_THREADS, lds, (hipStream_t)stream>>>(a);
    } else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

#ifdef AMID_WGRAD_STAMPS
extern "C" int amid_wgrad_sched_read(unsigned long long* host) {        // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_wgrad_sched_buf), sizeof(unsigned long long) * 2048);
}
extern "C" int amid_wgrad_stamps_read(unsigned long long* host) {       // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_wgrad_stamp_buf), sizeof(unsigned long long) * 64);
}
#endif

extern "C" int amid_sas_wgrad_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits,
                                  float* const* w_part, float* const* b_part, int mma_bf16, void* stream) {
    return sas_wgrad(dy, x, n_layers, M, D, splits, w_part, b_part, nullptr, 0, 0, mma_bf16, stream);
}

// the same with the loss structure as a hint (see WgradArgs::row_domain): M = B * T rows per domain, row_domain [B] = the batch's
// domain ids; the rows of the sequences whose dY is zero by construction are skipped -- same partial sums up to the zeros left out
extern "C" int amid_sas_wgrad_rows_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits,
                                       float* const* w_part, float* const* b_part, const long long* row_domain, int B, int T,
                                       int mma_bf16, void* stream) {
    return sas_wgrad(dy, x, n_layers, M, D, splits, w_part, b_part, row_domain, B, T, mma_bf16, stream);
}

// ... carrying the LAST phase (5: run heads) of a sort plan (amid_sort_plan_pack) as extra workgroups; mma_bf16 = 3, D = 128
extern "C" int amid_sas_wgrad_rows_sort_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits,
                                            float* const* w_part, float* const* b_part, const long long* row_domain, int B, int T,
                                            int mma_bf16, const void* sort_plan, void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr && row_domain != nullptr);
    return sas_wgrad(dy, x, n_layers, M, D, splits, w_part, b_part, row_domain, B, T, mma_bf16, stream, sort_plan);
}

// ... behind amid_sas_seq_fwd_split_lnstat_f32 (a forward that stored row statistics instead of qn = LN1(x) and y = LN2(r)): x[6 l + 0] is the
// layer's x, x[6 l + 4] its r, and the two operands are rebuilt as (row - mean) rstd gamma + beta while they are staged; ln_stat: n_layers
// pointers to [2 M][4] (mean1, rstd1, mean2, rstd2); ln1_w / ln1_b / ln2_w / ln2_b: 2 n_layers pointers ordered [layer][domain]
extern "C" int amid_sas_wgrad_rows_sort_ln_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits,
                                               float* const* w_part, float* const* b_part, const long long* row_domain, int B, int T,
                                               int mma_bf16, const void* sort_plan, const float* const* ln_stat, const float* const* ln1_w,
                                               const float* const* ln1_b, const float* const* ln2_w, const float* const* ln2_b, void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr && row_domain != nullptr && ln_stat != nullptr);
    return sas_wgrad(dy, x, n_layers, M, D, splits, w_part, b_part, row_domain, B, T, mma_bf16, stream, sort_plan, ln_stat, ln1_w, ln1_b, ln2_w, ln2_b);
}

extern "C" int amid_transpose_weights_f32(const float* const* src, float* const* dst, int n, int D, void* stream) {
    AMID_CHECK_ARG(src && dst && n > 0 && n <= 32 && D > 0 && (D % 32) == 0);
    TransposeArgs a;
    a.n = n;
    for (int i = 0; i < n; ++i) { AMID_CHECK_ARG(src[i] && dst[i]); a.src[i] = src[i]; a.dst[i] = dst[i]; }
    transpose_sq_kernel<<<dim3(D / 32, D / 32, n), 256, 0, (hipStream_t)stream>>>(a, D);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_reduce_entry_bytes(void) { return (int)sizeof(ReduceEntry); }

extern "C" int amid_reduce_entry_pack(void* host_buf, int index, const float* src, float* dst, long long stride, int n_part, int count) {
    AMID_CHECK_ARG(host_buf && src && dst && n_part > 0 && count > 0 && index >= 0);
    ReduceEntry e;
    e.src = src; e.dst = dst; e.stride = stride; e.n_part = n_part; e.count = count;
    ((ReduceEntry*)host_buf)[index] = e;
    return AMID_OK;
}

extern "C" int amid_reduce_partials_f32(const void* entries_dev, int n_entries, int max_count, void* stream) {
    AMID_CHECK_ARG(entries_dev && n_entries > 0 && max_count > 0);
    int bx = (max_count + 127) / 128;      // aligned entries move 128 elements per block pass; the (small) others loop
    if (bx > 512) bx = 512;
    reduce_partials_kernel<<<dim3(bx, n_entries), 256, 0, (hipStream_t)stream>>>((const ReduceEntry*)entries_dev);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
#endif
