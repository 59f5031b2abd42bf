// Fused forward kernels of one SASRec encoder layer (reference: Log2feats.forward model_seq.py:371-383,
// nn.MultiheadAttention projections as called at :374, PointWiseFeedForward model_seq.py:322-326).
// Both domains (sac1 / sac2) run in the same launch: rows [0, M) belong to domain 0, [M, 2M) to
// domain 1, each with its own weights; a row tile never straddles the two.
//
//   qkv_fwd   : Qn = LN1(x) ; q = Qn Wq^T + bq ; k = x Wk^T + bk ; v = x Wv^T + bv      (k, v from the UN-normed x)
//   oproj_fwd : r = Qn + (o Wo^T + bo)  (residual on the NORMED query, :378) ; y = LN2(r)
//   ffn_fwd   : h = relu(drop1(y C1^T + c1)) ; x' = (drop2(h C2^T + c2) + y) * ~tm        (:323-325, :383)
// MFMA-bound (exact fp32 MFMA, 64 FLOP/clk/SIMD); algorithmic FLOPs per row: 2*D*D per projection.
#include "common.h"
#include "rng.h"
#include "tile_gemm.h"

namespace amid {

struct TileGeom {           // how the 2*M activation rows are cut into row tiles
    int M;                  // rows per domain (B*T)
    int rows_per_tile;      // <= TILE_ROWS
    int tiles_per_group;
};

__device__ __forceinline__ void tile_rows(const TileGeom& tg, int tile, int& g, long long& row0, int& nrows, int& local0) {
    g = tile / tg.tiles_per_group;
    const int tl = tile - g * tg.tiles_per_group;
    local0 = tl * tg.rows_per_tile;
    nrows = min(tg.rows_per_tile, tg.M - local0);
    row0 = (long long)g * tg.M + local0;
}

struct QkvFwdArgs {
    const float* x;                 // [2M, D] layer input
    const float* ln_w[2]; const float* ln_b[2];
    const float* w_in[2];           // [3D, D] in_proj_weight
    const float* b_in[2];           // [3D]
    float* qn; float* q; float* k; float* v;   // [2M, D] each
    float ln_eps;
    TileGeom tg;
};

// xin != nullptr (fused after the previous layer's feed-forward, fp32): the tile's rows of x arrive in registers from that body --
// no global round trip -- and their global copy (the saved layer input) is stored from the A image inside the k slab's MFMA loop
template <int D, bool BF>
__device__ __forceinline__ void qkv_fwd_body(const QkvFwdArgs& a, float* __restrict__ smem, int tile, const TileRegs<D>* xin = nullptr) {
    using RP = RowPass<D>;
    float* As = smem;
    float* Ws = smem + TileCfg<D>::A_FLOATS;
    int g, nrows, local0; long long row0;
    tile_rows(a.tg, tile, g, row0, nrows, local0);
    const int sub = RP::sub();
    TileRegs<D> xr;
    WRegs<D, D> wr;
    if (xin != nullptr) xr = *xin; else load_tile<D>(xr, a.x, row0, nrows, D);
    load_w<D, D>(wr, a.w_in[g] + (long long)1 * D * D, D);           // slab order k, v, q
    const float4 lw = ld4(a.ln_w[g] + 4 * sub), lb = ld4(a.ln_b[g] + 4 * sub);
    tile_to_lds<D, BF>(As, xr);
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    f32x4 acc[WaveMap<D>::ACC];
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        const int which = (s == 0) ? 1 : (s == 1) ? 2 : 0;
        const int next = (s == 0) ? 2 : 0;
        if (s < 2) load_w<D, D>(wr, a.w_in[g] + (long long)next * D * D, D);   // next slab's weights fly under the MFMAs
        zero_acc<D>(acc);
        if (s == 2) {                                     // the q slab: Qn sits in the A image, its global copy leaves under these MFMAs
            const ImageRowsPending<D> pq{As, a.qn + row0 * D, D, nrows};
            mma_tile<D, D, BF>(As, Ws, acc, pq);
        } else if (s == 0 && xin != nullptr && !BF) {     // handed-over x: its global copy leaves under the k slab
            const ImageRowsPending<D> px{As, const_cast<float*>(a.x) + row0 * D, D, nrows};
            mma_tile<D, D, BF>(As, Ws, acc, px);
        } else {
            mma_tile<D, D, BF>(As, Ws, acc);
        }
        float* out = (which == 0) ? a.q : (which == 1) ? a.k : a.v;
        acc_to_global<D>(out, row0, nrows, D, a.b_in[g] + which * D, acc);
        if (s == 2) break;
        __syncthreads();                                  // all waves are done with As / Ws of this slab
        if (s == 1) {                                     // normalise the tile in place before the q slab; Qn also goes to global
#pragma unroll
            for (int i = 0; i < RP::NR; ++i) {
                const int r = RP::first_row() + i * RP::RPP;
                if (r < TileCfg<D>::ROWS) {                       // uniform over the QPR lanes of a row
                    float mean, rstd;
                    row_stats<RP::QPR>(xr.v[i], D, a.ln_eps, mean, rstd);
                    const float4 y = ln_apply(xr.v[i], mean, rstd, lw, lb);
                    store_a4<D, BF>(As, r, sub, y);
                    if constexpr (BF) { if (r < nrows) st4(a.qn + (row0 + r) * D + 4 * sub, y); }     // fp32: from the A image, below
                }
            }
        }
        w_to_lds<D, D, BF>(Ws, wr);
        __syncthreads();
    }
}

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_qkv_fwd_kernel(const QkvFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    qkv_fwd_body<D, BF>(a, smem, blockIdx.x);
}


struct OprojFwdArgs {
    const float* o;                  // [2M, D] attention output (heads merged)
    const float* w_o[2]; const float* b_o[2];
    const float* qn;                 // residual source
    const float* ln_w[2]; const float* ln_b[2];
    float* r; float* y;              // pre-LN2 sum, LN2 output
    float ln_eps;
    TileGeom tg;
};

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_oproj_fwd_kernel(const OprojFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<D>;
    constexpr int LDC = D + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<D>::A_FLOATS;
    float* Cs = Ws;
    int g, nrows, local0; long long row0;
    tile_rows(a.tg, blockIdx.x, g, row0, nrows, local0);
    const int sub = RP::sub();
    TileRegs<D> orr, res;
    WRegs<D, D> wr;
    load_tile<D>(orr, a.o, row0, nrows, D);
    load_w<D, D>(wr, a.w_o[g], D);
    load_tile<D>(res, a.qn, row0, nrows, D);              // residual rows: in flight during the MFMAs
    const float4 bias = ld4(a.b_o[g] + 4 * sub), w = ld4(a.ln_w[g] + 4 * sub), b = ld4(a.ln_b[g] + 4 * sub);
    tile_to_lds<D, BF>(As, orr);
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    f32x4 acc[WaveMap<D>::ACC];
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < nrows) {
            const long long off = (row0 + r) * D + 4 * sub;
            const float4 x = f4add(res.v[i], f4add(ld4(Cs + r * LDC + 4 * sub), bias));
            st4(a.r + off, x);
            float mean, rstd;
            row_stats<RP::QPR>(x, D, a.ln_eps, mean, rstd);
            st4(a.y + off, ln_apply(x, mean, rstd, w, b));
        }
    }
}

struct FfnFwdArgs {
    const float* y;                  // [2M, D] LN2 output
    const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];   // conv{1,2}.weight[:, :, 0], bias
    const unsigned char* tmq;        // [2M, D/4] feature-level "== 0" bits (may be null: no mask)
    float* h; float* xo;             // relu output, layer output
    const StepState* st; int train; unsigned thr16; float scale; int layer;
    TileGeom tg;
};

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_ffn_fwd_kernel(const FfnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<D>;
    constexpr int LDC = D + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<D>::A_FLOATS;
    float* Cs = Ws;
    int g, nrows, local0; long long row0;
    tile_rows(a.tg, blockIdx.x, g, row0, nrows, local0);
    const int sub = RP::sub();
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    TileRegs<D> yr;                                       // kept: the residual of the second epilogue
    WRegs<D, D> wr;
    load_tile<D>(yr, a.y, row0, nrows, D);
    load_w<D, D>(wr, a.w1[g], D);
    unsigned tm[RP::NR];
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        tm[i] = (a.tmq && r < nrows) ? a.tmq[(row0 + r) * (D / 4) + sub] : 0u;
    }
    const float4 bias1 = ld4(a.b1[g] + 4 * sub), bias2 = ld4(a.b2[g] + 4 * sub);
    tile_to_lds<D, BF>(As, yr);
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_w<D, D>(wr, a.w2[g], D);                         // second weight matrix flies under the first GEMM
    f32x4 acc[WaveMap<D>::ACC];
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
    // h = relu(drop1(C + c1)) -> global and the A image of the second GEMM
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < TileCfg<D>::ROWS) {
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrows) {
                hv = f4add(ld4(Cs + r * LDC + 4 * sub), bias1);
                if (a.train) hv = f4mul(hv, dropout_mult4(seed, site_id(g, a.layer, SITE_FFN1), step,
                                                          (unsigned long long)(local0 + r) * D + 4 * sub, a.thr16, a.scale));
                hv.x = fmaxf(hv.x, 0.f); hv.y = fmaxf(hv.y, 0.f); hv.z = fmaxf(hv.z, 0.f); hv.w = fmaxf(hv.w, 0.f);
                st4(a.h + (row0 + r) * D + 4 * sub, hv);
            }
            store_a4<D, BF>(As, r, sub, hv);
        }
    }
    __syncthreads();                                      // Cs (= Ws) fully read, As rewritten
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < nrows) {
            float4 z = f4add(ld4(Cs + r * LDC + 4 * sub), bias2);
            if (a.train) z = f4mul(z, dropout_mult4(seed, site_id(g, a.layer, SITE_FFN2), step,
                                                    (unsigned long long)(local0 + r) * D + 4 * sub, a.thr16, a.scale));
            z = f4add(z, yr.v[i]);
            const unsigned bits = tm[i];
            if (bits) {
                if (bits & 1u) z.x = 0.f;
                if (bits & 2u) z.y = 0.f;
                if (bits & 4u) z.z = 0.f;
                if (bits & 8u) z.w = 0.f;
            }
            st4(a.xo + (row0 + r) * D + 4 * sub, z);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// out-projection + residual + LayerNorm2 + point-wise feed-forward in ONE kernel (sas_oproj_fwd_kernel followed by
// sas_ffn_fwd_kernel on the same row tile): the LN2 output y never leaves the chip between the two, one prologue and one
// launch instead of two.  Three chained GEMMs on one A image: o Wo^T -> (r, y) ; y C1^T -> h ; h C2^T -> xo.
// ---------------------------------------------------------------------------------------------------------------------
struct OprojFfnFwdArgs {
    const float* o; const float* qn;
    const float* w_o[2]; const float* b_o[2]; const float* ln_w[2]; const float* ln_b[2];
    const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
    const unsigned char* tmq;
    float* r; float* y; float* h; float* xo;
    float ln_eps;
    const StepState* st; int train; unsigned thr16; float scale; int layer;
    TileGeom tg;
};

// keep != nullptr: the layer output rows are ALSO left in registers for a fused successor (and in fp32 not stored here: the
// successor stores them from its A image)
template <int D, bool BF>
__device__ __forceinline__ void oproj_ffn_fwd_body(const OprojFfnFwdArgs& a, float* __restrict__ smem, int tile, TileRegs<D>* keep = nullptr) {
    using RP = RowPass<D>;
    constexpr int LDC = D + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<D>::A_FLOATS;
    float* Cs = Ws;
    int g, nrows, local0; long long row0;
    tile_rows(a.tg, tile, g, row0, nrows, local0);
    const int sub = RP::sub();
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    TileRegs<D> tr;                                       // o rows, then the residual (qn), then y (the FFN residual)
    WRegs<D, D> wr;
    load_tile<D>(tr, a.o, row0, nrows, D);
    load_w<D, D>(wr, a.w_o[g], D);
    tile_to_lds<D, BF>(As, tr);
    w_to_lds<D, D, BF>(Ws, wr);
    load_tile<D>(tr, a.qn, row0, nrows, D);               // residual rows and the next weights fly under the first GEMM
    load_w<D, D>(wr, a.w1[g], D);
    unsigned tm[RP::NR];
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        tm[i] = (a.tmq && r < nrows) ? a.tmq[(row0 + r) * (D / 4) + sub] : 0u;
    }
    __syncthreads();
    f32x4 acc[WaveMap<D>::ACC];
    zero_acc<D>(acc);
    mma_tile<D, D, BF>(As, Ws, acc);
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
    {   // r = qn + (o Wo^T + bo) ; y = LN2(r) -> global, registers (FFN residual) and the A image of the second GEMM
        const float4 bias = ld4(a.b_o[g] + 4 * sub), w = ld4(a.ln_w[g] + 4 * sub), b = ld4(a.ln_b[g] + 4 * sub);
#pragma unroll
        for (int i = 0; i < RP::NR; ++i) {
            const int r = RP::first_row() + i * RP::RPP;
            if (r < TileCfg<D>::ROWS) {
                float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrows) {
                    const long long off = (row0 + r) * D + 4 * sub;
                    const float4 x = f4add(tr.v[i], f4add(ld4(Cs + r * LDC + 4 * sub), bias));
                    st4(a.r + off, x);
                    float mean, rstd;
                    row_stats<RP::QPR>(x, D, a.ln_eps, mean, rstd);
                    yv = ln_apply(x, mean, rstd, w, b);
                    if constexpr (BF) st4(a.y + off, yv);            // fp32: stored from the A image inside the next MFMA loop
                }
                tr.v[i] = yv;
                store_a4<D, BF>(As, r, sub, yv);
            }
        }
    }
    __syncthreads();                                      // Cs (= Ws) fully read, As rewritten
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    load_w<D, D>(wr, a.w2[g], D);
    const float4 bias1 = ld4(a.b1[g] + 4 * sub), bias2 = ld4(a.b2[g] + 4 * sub);
    zero_acc<D>(acc);
    {
        const ImageRowsPending<D> py{As, a.y + row0 * D, D, nrows};
        mma_tile<D, D, BF>(As, Ws, acc, py);
    }
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {                    // h = relu(drop1(C + c1)) -> global and the A image of the third GEMM
        const int r = RP::first_row() + i * RP::RPP;
        if (r < TileCfg<D>::ROWS) {
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrows) {
                hv = f4add(ld4(Cs + r * LDC + 4 * sub), bias1);
                if (a.train) hv = f4mul(hv, dropout_mult4(seed, site_id(g, a.layer, SITE_FFN1), step,
                                                          (unsigned long long)(local0 + r) * D + 4 * sub, a.thr16, a.scale));
                hv.x = fmaxf(hv.x, 0.f); hv.y = fmaxf(hv.y, 0.f); hv.z = fmaxf(hv.z, 0.f); hv.w = fmaxf(hv.w, 0.f);
                if constexpr (BF) st4(a.h + (row0 + r) * D + 4 * sub, hv);
            }
            store_a4<D, BF>(As, r, sub, hv);
        }
    }
    __syncthreads();
    w_to_lds<D, D, BF>(Ws, wr);
    __syncthreads();
    zero_acc<D>(acc);
    {
        const ImageRowsPending<D> ph{As, a.h + row0 * D, D, nrows};
        mma_tile<D, D, BF>(As, Ws, acc, ph);
    }
    __syncthreads();
    acc_to_lds<D>(Cs, LDC, acc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows) {
            z = f4add(ld4(Cs + r * LDC + 4 * sub), bias2);
            if (a.train) z = f4mul(z, dropout_mult4(seed, site_id(g, a.layer, SITE_FFN2), step,
                                                    (unsigned long long)(local0 + r) * D + 4 * sub, a.thr16, a.scale));
            z = f4add(z, tr.v[i]);
            const unsigned bits = tm[i];
            if (bits) {
                if (bits & 1u) z.x = 0.f;
                if (bits & 2u) z.y = 0.f;
                if (bits & 4u) z.z = 0.f;
                if (bits & 8u) z.w = 0.f;
            }
            if (keep == nullptr || BF) st4(a.xo + (row0 + r) * D + 4 * sub, z);
        }
        if (keep != nullptr) keep->v[i] = z;
    }
}

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_oproj_ffn_fwd_kernel(const OprojFfnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    oproj_ffn_fwd_body<D, BF>(a, smem, blockIdx.x);
}

// layer l's out-projection + feed-forward followed by layer l + 1's LayerNorm + q / k / v projections on the same row tile, one
// launch: the tile's rows of x[l + 1] stay in registers between the two bodies (their global copy, the saved layer input, is
// written from the A image under the k slab's MFMAs), and one launch + prologue of the forward pass disappears
struct OprojFfnQkvArgs { OprojFfnFwdArgs of; QkvFwdArgs qkv; TileGeom tg; };

template <int D, bool BF>
__global__ __launch_bounds__(GEMM_THREADS) void sas_oproj_ffn_qkv_fwd_kernel(const OprojFfnQkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    TileRegs<D> xnext;          // layer l + 1's input rows go from the feed-forward epilogue to the q / k / v body in registers
    oproj_ffn_fwd_body<D, BF>(a.of, smem, blockIdx.x, &xnext);
    __syncthreads();            // the C image (= W region) of the last epilogue is read; the second body restages both LDS regions
    qkv_fwd_body<D, BF>(a.qkv, smem, blockIdx.x, &xnext);
}

}  // namespace amid

using namespace amid;

template <int D> static constexpr size_t fused_lds_bytes() { return (size_t)(TileCfg<D>::A_FLOATS + TileCfg<D>::W_FLOATS) * sizeof(float); }

static int make_geom(int M, int rows_per_tile, TileGeom* tg) {
    if (M <= 0 || rows_per_tile <= 0 || rows_per_tile > TILE_ROWS) return AMID_ERR_ARG;
    tg->M = M;
    tg->rows_per_tile = rows_per_tile;
    tg->tiles_per_group = (M + rows_per_tile - 1) / rows_per_tile;
    return AMID_OK;
}

#if AMID_TILE_RT == 7
// Rows per tile for a launch over 2*M rows: spread the rows evenly over a whole number of rounds of the
// 256 CUs (one 512-thread workgroup per CU), at most TILE_ROWS rows per tile.
extern "C" int amid_rows_per_tile(int M) {
    const long long total = 2LL * M;
    for (int rounds = 1;; ++rounds) {
        const long long tiles = 256LL * rounds;
        long long rpt = (total + tiles - 1) / tiles;
        if (rpt <= TILE_ROWS) {
            if (rpt < 16) rpt = 16;
            return (int)rpt;
        }
    }
}

#endif

#define AMID_LAUNCH_FUSED(KERNEL, ARGS, DVAL, BFVAL)                                                                        \
    do {                                                                                                                   \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL<DVAL, BFVAL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                               (int)fused_lds_bytes<DVAL>());                                              \
            if (e != hipSuccess) return (int)e;                                                                            \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        KERNEL<DVAL, BFVAL><<<2 * ARGS.tg.tiles_per_group, GEMM_THREADS, fused_lds_bytes<DVAL>(), (hipStream_t)stream>>>(ARGS); \
    } while (0)

extern "C" int AMID_ENTRY(amid_sas_qkv_fwd_f32)(const float* x, const float* const* ln_w, const float* const* ln_b, const float* const* w_in,
                                    const float* const* b_in, float ln_eps, int M, int D, int rows_per_tile, float* qn, float* q, float* k,
                                    float* v, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(x && ln_w && ln_b && w_in && b_in && qn && q && k && v);
    QkvFwdArgs a;
    a.x = x; a.qn = qn; a.q = q; a.k = k; a.v = v; a.ln_eps = ln_eps;
    for (int g = 0; g < 2; ++g) { a.ln_w[g] = ln_w[g]; a.ln_b[g] = ln_b[g]; a.w_in[g] = w_in[g]; a.b_in[g] = b_in[g]; }
    if (int e = make_geom(M, rows_per_tile, &a.tg)) return e;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED(sas_qkv_fwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED(sas_qkv_fwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED(sas_qkv_fwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_oproj_fwd_f32)(const float* o, const float* const* w_o, const float* const* b_o, const float* qn,
                                      const float* const* ln_w, const float* const* ln_b, float ln_eps, int M, int D, int rows_per_tile,
                                      float* r, float* y, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(o && w_o && b_o && qn && ln_w && ln_b && r && y);
    OprojFwdArgs a;
    a.o = o; a.qn = qn; a.r = r; a.y = y; a.ln_eps = ln_eps;
    for (int g = 0; g < 2; ++g) { a.w_o[g] = w_o[g]; a.b_o[g] = b_o[g]; a.ln_w[g] = ln_w[g]; a.ln_b[g] = ln_b[g]; }
    if (int e = make_geom(M, rows_per_tile, &a.tg)) return e;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED(sas_oproj_fwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED(sas_oproj_fwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED(sas_oproj_fwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_ffn_fwd_f32)(const float* y, const float* const* w1, const float* const* b1, const float* const* w2,
                                    const float* const* b2, const unsigned char* tmq, int M, int D, int rows_per_tile, int layer,
                                    const void* step_state, int train, float p_drop, float* h, float* xo, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(y && w1 && b1 && w2 && b2 && h && xo && (!train || step_state));
    FfnFwdArgs a;
    a.y = y; a.tmq = tmq; a.h = h; a.xo = xo; a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.thr16 = keep_thr16(p_drop);
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) { a.w1[g] = w1[g]; a.b1[g] = b1[g]; a.w2[g] = w2[g]; a.b2[g] = b2[g]; }
    if (int e = make_geom(M, rows_per_tile, &a.tg)) return e;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED(sas_ffn_fwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED(sas_ffn_fwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED(sas_ffn_fwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_oproj_ffn_fwd_f32)(const float* o, const float* qn, const float* const* w_o, const float* const* b_o,
                                          const float* const* ln_w, const float* const* ln_b, const float* const* w1, const float* const* b1,
                                          const float* const* w2, const float* const* b2, const unsigned char* tmq, float ln_eps, int M, int D,
                                          int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* r, float* y,
                                          float* h, float* xo, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(o && qn && w_o && b_o && ln_w && ln_b && w1 && b1 && w2 && b2 && r && y && h && xo && (!train || step_state));
    OprojFfnFwdArgs a;
    a.o = o; a.qn = qn; a.tmq = tmq; a.r = r; a.y = y; a.h = h; a.xo = xo; a.ln_eps = ln_eps;
    a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.thr16 = keep_thr16(p_drop);
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) {
        a.w_o[g] = w_o[g]; a.b_o[g] = b_o[g]; a.ln_w[g] = ln_w[g]; a.ln_b[g] = ln_b[g];
        a.w1[g] = w1[g]; a.b1[g] = b1[g]; a.w2[g] = w2[g]; a.b2[g] = b2[g];
    }
    if (int e = make_geom(M, rows_per_tile, &a.tg)) return e;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED(sas_oproj_ffn_fwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED(sas_oproj_ffn_fwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED(sas_oproj_ffn_fwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_sas_oproj_ffn_qkv_fwd_f32)(const float* o, const float* qn, const float* const* w_o, const float* const* b_o,
                                              const float* const* ln_w, const float* const* ln_b, const float* const* w1, const float* const* b1,
                                              const float* const* w2, const float* const* b2, const unsigned char* tmq, float ln_eps, int M, int D,
                                              int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* r, float* y,
                                              float* h, float* xo, const float* const* nln_w, const float* const* nln_b,
                                              const float* const* nw_in, const float* const* nb_in, float* nqn, float* nq, float* nk, float* nv,
                                              int mma_bf16, void* stream) {
    AMID_CHECK_ARG(o && qn && w_o && b_o && ln_w && ln_b && w1 && b1 && w2 && b2 && r && y && h && xo && (!train || step_state));
    AMID_CHECK_ARG(nln_w && nln_b && nw_in && nb_in && nqn && nq && nk && nv);
    OprojFfnQkvArgs a;
    a.of.o = o; a.of.qn = qn; a.of.tmq = tmq; a.of.r = r; a.of.y = y; a.of.h = h; a.of.xo = xo; a.of.ln_eps = ln_eps;
    a.of.st = (const StepState*)step_state; a.of.layer = layer;
    a.of.train = (train && p_drop > 0.f) ? 1 : 0;
    a.of.thr16 = keep_thr16(p_drop);
    a.of.scale = a.of.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) {
        a.of.w_o[g] = w_o[g]; a.of.b_o[g] = b_o[g]; a.of.ln_w[g] = ln_w[g]; a.of.ln_b[g] = ln_b[g];
        a.of.w1[g] = w1[g]; a.of.b1[g] = b1[g]; a.of.w2[g] = w2[g]; a.of.b2[g] = b2[g];
        a.qkv.ln_w[g] = nln_w[g]; a.qkv.ln_b[g] = nln_b[g]; a.qkv.w_in[g] = nw_in[g]; a.qkv.b_in[g] = nb_in[g];
    }
    a.qkv.x = xo; a.qkv.qn = nqn; a.qkv.q = nq; a.qkv.k = nk; a.qkv.v = nv; a.qkv.ln_eps = ln_eps;
    if (int e = make_geom(M, rows_per_tile, &a.of.tg)) return e;
    a.qkv.tg = a.of.tg; a.tg = a.of.tg;
    if (D == 128 && mma_bf16) AMID_LAUNCH_FUSED(sas_oproj_ffn_qkv_fwd_kernel, a, 128, true);
    else if (D == 128) AMID_LAUNCH_FUSED(sas_oproj_ffn_qkv_fwd_kernel, a, 128, false);
    else if (D == 64 && !mma_bf16) AMID_LAUNCH_FUSED(sas_oproj_ffn_qkv_fwd_kernel, a, 64, false);
    else return AMID_ERR_UNSUPPORTED;
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
