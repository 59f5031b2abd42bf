// BERT4Rec encoder block (reference: TransformerBlock.forward model_seq.py:242-245 with SublayerConnection :140-142,
// the hand-written LayerNorm :124-127 -- UNBIASED std, eps added to the std --, MultiHeadedAttention :183-196,
// PositionwiseFeedForward :216-217 with the tanh GELU :204; hidden size 128 / 4 heads / FFN 512 / dropout 0.1 are hard-coded
// by the reference, :264-267), forward and backward, on the same fp32 MFMA row-tile machinery as the SASRec layer.
//
//   qkv_fwd   : y = LNb_in(x) ; q, k, v = y W{0,1,2}^T + b                          (all three from the normed y)
//   [attention: attention.hip general kernels -- bidirectional, key mask from seq_d2 > 0, scores / sqrt(d_k)]
//   oproj_fwd : x1 = x + drop_sub_in(o Wo^T + bo)
//   ffn1_fwd  : y2 = LNb_out(x1) ; pre = y2 W1^T + b1 (128 -> 512) ; h = drop_ffn(gelu(pre))
//   ffn2_fwd  : x2 = drop_block(x1 + drop_sub_out(h W2^T + b2))                      (512 -> 128, K in four chunks)
// Backward mirrors it with transposed weights; weight gradients reuse the split-partials scheme (bert_wgrad).
#include "common.h"
#include "rng.h"
#include "tile_gemm.h"
#include "wgrad_split.h"
#include "bert_math.h"

namespace amid {

constexpr int BD = 128;      // hidden
constexpr int BF = 512;      // feed-forward

struct BGeom { int M; int rows_per_tile; int tiles_per_group; };
__device__ __forceinline__ void btile(const BGeom& tg, int tile, int& g, long long& row0, int& nrows, int& local0) {
    g = tile / tg.tiles_per_group;
    const int tl = tile - g * tg.tiles_per_group;
    local0 = tl * tg.rows_per_tile;
    nrows = min(tg.rows_per_tile, tg.M - local0);
    row0 = (long long)g * tg.M + local0;
}

// the backward kernels' view of a tile.  (A variant that gathered tiles of LIVE sequences only -- half the tiles -- was built and measured:
// no gain, each launch is one round of workgroups either way and lasts as long as one tile's chain of weight slabs; removed.)
struct BTile {
    int g, nrows, slot, local0;
    long long base;                    // g * M
    __device__ __forceinline__ int lrow(int r) const { return local0 + r; }
    __device__ __forceinline__ long long grow(int r) const { return base + lrow(r); }
};
__device__ __forceinline__ BTile btile_bwd(const BGeom& tg, int tile) {
    BTile t;
    long long row0;
    btile(tg, tile, t.g, row0, t.nrows, t.local0);
    t.base = (long long)t.g * tg.M;
    t.slot = tile;
    return t;
}

struct DropCfg { const StepState* st; int train; unsigned spec; float scale; int layer; };
__device__ __forceinline__ float4 bdrop(const DropCfg& d, unsigned long long seed, unsigned step, int g, int kind, unsigned long long e0, float4 v) {
    return d.train ? f4mul(v, dropout_mult4(seed, site_id(g, d.layer, kind), step, e0, d.spec, d.scale)) : v;
}

// reference LayerNorm: a * (x - mean) / (std_unbiased + eps) + b, one row over 32 lanes (float4 each)
__device__ __forceinline__ float4 lnb_fwd(float4 x, float4 a, float4 b, float& mean, float& r /* 1/(std+eps) */) {
    mean = group_sum<32>(f4hsum(x)) * (1.0f / BD);
    const float4 xc = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
    const float sd = sqrtf(group_sum<32>(f4hsum(f4mul(xc, xc))) * (1.0f / (BD - 1)));
    r = 1.0f / (sd + BERT_EPS);
    return make_float4(a.x * xc.x * r + b.x, a.y * xc.y * r + b.y, a.z * xc.z * r + b.z, a.w * xc.w * r + b.w);
}
// backward: dx = r (g - mean(g)) - t r^2 xc / (std (D-1)), g = a dy, t = sum(g xc); accumulates d a (dy xc r) and d b (dy)
__device__ __forceinline__ float4 lnb_bwd(float4 dy, float4 x, float4 a, float4& da, float4& db) {
    const float mean = group_sum<32>(f4hsum(x)) * (1.0f / BD);
    const float4 xc = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
    const float sd = sqrtf(group_sum<32>(f4hsum(f4mul(xc, xc))) * (1.0f / (BD - 1)));
    const float r = 1.0f / (sd + BERT_EPS);
    const float4 gg = f4mul(a, dy);
    const float gm = group_sum<32>(f4hsum(gg)) * (1.0f / BD);
    const float t = group_sum<32>(f4hsum(f4mul(gg, xc)));
    const float c = (sd > 0.f) ? t * r * r / (sd * (BD - 1)) : 0.f;
    da = f4add(da, f4scale(f4mul(dy, xc), r));
    db = f4add(db, dy);
    return make_float4(r * (gg.x - gm) - c * xc.x, r * (gg.y - gm) - c * xc.y, r * (gg.z - gm) - c * xc.z, r * (gg.w - gm) - c * xc.w);
}

// visit the accumulator tiles of a wave: f(row r in tile, first column n, float4 value)
template <class F>
__device__ __forceinline__ void acc_visit(const f32x4 (&acc)[WaveMap<BD>::ACC], int nrows, F f) {
    using WM = WaveMap<BD>;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int m = lane & 15, n = ct * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < WM::ACC; ++t) {
        const int r = (rg + t * WM::WR) * 16 + m;
        if (r < nrows) f(r, n, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
    }
}

template <int D>
__device__ __forceinline__ void ln_part_out(float* __restrict__ scratch, float4 dgam, float4 dbet, float* __restrict__ part) {
    using RP = RowPass<D>;
    const int sub = RP::sub(), slot = RP::first_row();
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

// ------------------------------------------------------------------------------------------------ forward
struct BQkvArgs {
    const float* x; const float* la[2]; const float* lb[2];
    const float* w[3][2]; const float* b[3][2];            // linear_layers.{0,1,2}.{weight,bias} per domain
    float* y; float* out[3];                               // LNb(x) ; q, k, v
    BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_qkv_fwd_kernel(const BQkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<BD>;
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    int g, nrows, local0; long long row0;
    btile(a.tg, blockIdx.x, g, row0, nrows, local0);
    const int sub = RP::sub();
    TileRegs<BD> xr;
    WRegs<BD, BD> wr;
    load_tile<BD>(xr, a.x, row0, nrows, BD);
    load_w<BD, BD>(wr, a.w[0][g], BD);
    const float4 la = ld4(a.la[g] + 4 * sub), lb = ld4(a.lb[g] + 4 * sub);
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        float mean, rr;
        const float4 y = (r < nrows) ? lnb_fwd(xr.v[i], la, lb, mean, rr) : make_float4(0.f, 0.f, 0.f, 0.f);
        xr.v[i] = y;
        if (r < nrows) st4(a.y + (row0 + r) * BD + 4 * sub, y);
    }
    tile_to_lds<BD>(As, xr);
    w_to_lds<BD, BD>(Ws, wr);
    __syncthreads();
    f32x4 acc[WaveMap<BD>::ACC];
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        if (s < 2) load_w<BD, BD>(wr, a.w[s + 1][g], BD);
        zero_acc<BD>(acc);
        mma_tile<BD, BD>(As, Ws, acc);
        acc_to_global<BD>(a.out[s], row0, nrows, BD, a.b[s][g], acc);
        if (s == 2) break;
        __syncthreads();
        w_to_lds<BD, BD>(Ws, wr);
        __syncthreads();
    }
}

struct BOprojArgs {
    const float* o; const float* x; const float* w[2]; const float* b[2];
    float* x1; DropCfg dc; BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_oproj_fwd_kernel(const BOprojArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    int g, nrows, local0; long long row0;
    btile(a.tg, blockIdx.x, g, row0, nrows, local0);
    unsigned long long seed = 0; unsigned step = 0;
    if (a.dc.train) { seed = a.dc.st->seed; step = (unsigned)a.dc.st->step; }
    TileRegs<BD> orr;
    WRegs<BD, BD> wr;
    load_tile<BD>(orr, a.o, row0, nrows, BD);
    load_w<BD, BD>(wr, a.w[g], BD);
    tile_to_lds<BD>(As, orr);
    w_to_lds<BD, BD>(Ws, wr);
    __syncthreads();
    f32x4 acc[WaveMap<BD>::ACC];
    zero_acc<BD>(acc);
    mma_tile<BD, BD>(As, Ws, acc);
    const float* bias = a.b[g];
    acc_visit(acc, nrows, [&](int r, int n, float4 c) {
        const long long off = (row0 + r) * BD + n;
        float4 t = f4add(c, ld4(bias + n));
        t = bdrop(a.dc, seed, step, g, SITE_SUB_IN, (unsigned long long)(local0 + r) * BD + n, t);
        st4(a.x1 + off, f4add(ld4(a.x + off), t));
    });
}

struct BFfn1Args {
    const float* x1; const float* la[2]; const float* lb[2];
    const float* w1[2]; const float* b1[2];                // [512,128], [512]
    float* y2; float* pre; float* h;                       // [2M,128], [2M,512], [2M,512]
    DropCfg dc; BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_ffn1_fwd_kernel(const BFfn1Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<BD>;
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    int g, nrows, local0; long long row0;
    btile(a.tg, blockIdx.x, g, row0, nrows, local0);
    const int sub = RP::sub();
    unsigned long long seed = 0; unsigned step = 0;
    if (a.dc.train) { seed = a.dc.st->seed; step = (unsigned)a.dc.st->step; }
    TileRegs<BD> xr;
    WRegs<BD, BD> wr;
    load_tile<BD>(xr, a.x1, row0, nrows, BD);
    load_w<BD, BD>(wr, a.w1[g], BD);
    const float4 la = ld4(a.la[g] + 4 * sub), lb = ld4(a.lb[g] + 4 * sub);
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        float mean, rr;
        const float4 y = (r < nrows) ? lnb_fwd(xr.v[i], la, lb, mean, rr) : make_float4(0.f, 0.f, 0.f, 0.f);
        xr.v[i] = y;
        if (r < nrows) st4(a.y2 + (row0 + r) * BD + 4 * sub, y);
    }
    tile_to_lds<BD>(As, xr);
    w_to_lds<BD, BD>(Ws, wr);
    __syncthreads();
    f32x4 acc[WaveMap<BD>::ACC];
#pragma unroll 1
    for (int s = 0; s < BF / BD; ++s) {
        if (s + 1 < BF / BD) load_w<BD, BD>(wr, a.w1[g] + (long long)(s + 1) * BD * BD, BD);
        zero_acc<BD>(acc);
        mma_tile<BD, BD>(As, Ws, acc);
        const float* bias = a.b1[g] + s * BD;
        acc_visit(acc, nrows, [&](int r, int n, float4 c) {
            const long long off = (row0 + r) * BF + s * BD + n;
            const float4 p = f4add(c, ld4(bias + n));
            st4(a.pre + off, p);
            float4 hv = make_float4(gelu_f(p.x), gelu_f(p.y), gelu_f(p.z), gelu_f(p.w));
            hv = bdrop(a.dc, seed, step, g, SITE_FFN1, (unsigned long long)(local0 + r) * BF + s * BD + n, hv);
            st4(a.h + off, hv);
        });
        if (s + 1 == BF / BD) break;
        __syncthreads();
        w_to_lds<BD, BD>(Ws, wr);
        __syncthreads();
    }
}

struct BFfn2Args {
    const float* h; const float* x1; const float* w2[2]; const float* b2[2];   // [128,512], [128]
    float* x2; DropCfg dc; BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_ffn2_fwd_kernel(const BFfn2Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    int g, nrows, local0; long long row0;
    btile(a.tg, blockIdx.x, g, row0, nrows, local0);
    unsigned long long seed = 0; unsigned step = 0;
    if (a.dc.train) { seed = a.dc.st->seed; step = (unsigned)a.dc.st->step; }
    TileRegs<BD> ar;
    WRegs<BD, BD> wr;
    load_tile<BD>(ar, a.h, row0, nrows, BF);
    load_w<BD, BD>(wr, a.w2[g], BF);
    f32x4 acc[WaveMap<BD>::ACC];
    zero_acc<BD>(acc);
#pragma unroll 1
    for (int kc = 0; kc < BF / BD; ++kc) {
        if (kc) __syncthreads();                           // previous chunk's images fully consumed
        tile_to_lds<BD>(As, ar);
        w_to_lds<BD, BD>(Ws, wr);
        __syncthreads();
        if (kc + 1 < BF / BD) {
            load_tile<BD>(ar, a.h + (kc + 1) * BD, row0, nrows, BF);
            load_w<BD, BD>(wr, a.w2[g] + (kc + 1) * BD, BF);
        }
        mma_tile<BD, BD>(As, Ws, acc);
    }
    const float* bias = a.b2[g];
    acc_visit(acc, nrows, [&](int r, int n, float4 c) {
        const long long off = (row0 + r) * BD + n;
        const unsigned long long e0 = (unsigned long long)(local0 + r) * BD + n;
        float4 z = f4add(c, ld4(bias + n));
        z = bdrop(a.dc, seed, step, g, SITE_SUB_OUT, e0, z);
        z = f4add(z, ld4(a.x1 + off));
        z = bdrop(a.dc, seed, step, g, SITE_BLOCK, e0, z);
        st4(a.x2 + off, z);
    });
}

// ------------------------------------------------------------------------------------------------ backward
struct BFfn2BwdArgs {
    const float* dx2; const float* pre; const float* w2T[2];   // w2T [512,128]
    float* dz; float* dpre;                                     // [2M,128], [2M,512]
    DropCfg dc; BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_ffn2_bwd_kernel(const BFfn2BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<BD>;
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    const BTile tr = btile_bwd(a.tg, blockIdx.x);
    const int g = tr.g, nrows = tr.nrows;
    auto rowf = [&](int r) { return tr.grow(r); };
    const int sub = RP::sub();
    unsigned long long seed = 0; unsigned step = 0;
    if (a.dc.train) { seed = a.dc.st->seed; step = (unsigned)a.dc.st->step; }
    TileRegs<BD> dr;
    WRegs<BD, BD> wr;
    load_tile_rows<BD>(dr, a.dx2, rowf, nrows, BD);
    load_w<BD, BD>(wr, a.w2T[g], BD);
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < nrows) {
            const unsigned long long e0 = (unsigned long long)tr.lrow(r) * BD + 4 * sub;
            float4 v = bdrop(a.dc, seed, step, g, SITE_BLOCK, e0, dr.v[i]);
            v = bdrop(a.dc, seed, step, g, SITE_SUB_OUT, e0, v);
            dr.v[i] = v;
            st4(a.dz + tr.grow(r) * BD + 4 * sub, v);
        }
    }
    tile_to_lds<BD>(As, dr);
    w_to_lds<BD, BD>(Ws, wr);
    __syncthreads();
    f32x4 acc[WaveMap<BD>::ACC];
#pragma unroll 1
    for (int s = 0; s < BF / BD; ++s) {
        if (s + 1 < BF / BD) load_w<BD, BD>(wr, a.w2T[g] + (long long)(s + 1) * BD * BD, BD);
        zero_acc<BD>(acc);
        mma_tile<BD, BD>(As, Ws, acc);
        acc_visit(acc, nrows, [&](int r, int n, float4 c) {
            const long long off = tr.grow(r) * BF + s * BD + n;
            const float4 p = ld4(a.pre + off);
            float4 d = bdrop(a.dc, seed, step, g, SITE_FFN1, (unsigned long long)tr.lrow(r) * BF + s * BD + n, c);
            d = make_float4(d.x * gelu_df(p.x), d.y * gelu_df(p.y), d.z * gelu_df(p.z), d.w * gelu_df(p.w));
            st4(a.dpre + off, d);
        });
        if (s + 1 == BF / BD) break;
        __syncthreads();
        w_to_lds<BD, BD>(Ws, wr);
        __syncthreads();
    }
}

struct BFfn1BwdArgs {
    const float* dpre; const float* dx2; const float* x1; const float* la[2];   // LNb_out gamma
    const float* w1T[2]; const float* woT[2];              // w1T [128,512] ; woT [128,128]
    float* dx1; float* dt; float* d_o; float* ln_part;     // [2M,128] x3 ; [tiles][2][128]
    DropCfg dc; BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_ffn1_bwd_kernel(const BFfn1BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<BD>;
    constexpr int LDK = TileCfg<BD>::LDK, LDC = BD + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    float* Cs = Ws;
    const BTile tr = btile_bwd(a.tg, blockIdx.x);
    const int g = tr.g, nrows = tr.nrows;
    auto rowf = [&](int r) { return tr.grow(r); };
    const int sub = RP::sub();
    unsigned long long seed = 0; unsigned step = 0;
    if (a.dc.train) { seed = a.dc.st->seed; step = (unsigned)a.dc.st->step; }
    TileRegs<BD> ar, xr, dxr;
    WRegs<BD, BD> wr;
    load_tile_rows<BD>(ar, a.dpre, rowf, nrows, BF);
    load_w<BD, BD>(wr, a.w1T[g], BF);
    f32x4 acc[WaveMap<BD>::ACC];
    zero_acc<BD>(acc);
#pragma unroll 1
    for (int kc = 0; kc < BF / BD; ++kc) {
        if (kc) __syncthreads();
        tile_to_lds<BD>(As, ar);
        w_to_lds<BD, BD>(Ws, wr);
        __syncthreads();
        if (kc + 1 < BF / BD) {
            load_tile_rows<BD>(ar, a.dpre + (kc + 1) * BD, rowf, nrows, BF);
            load_w<BD, BD>(wr, a.w1T[g] + (kc + 1) * BD, BF);
        } else {
            load_w<BD, BD>(wr, a.woT[g], BD);              // next weights fly under the last chunk
        }
        mma_tile<BD, BD>(As, Ws, acc);
    }
    load_tile_rows<BD>(xr, a.x1, rowf, nrows, BD);              // epilogue inputs (kept out of the MFMA loop: register budget)
    load_tile_rows<BD>(dxr, a.dx2, rowf, nrows, BD);
    __syncthreads();
    acc_to_lds<BD>(Cs, LDC, acc);
    __syncthreads();
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    const float4 gam = ld4(a.la[g] + 4 * sub);
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows) {
            const unsigned long long e0 = (unsigned long long)tr.lrow(r) * BD + 4 * sub;
            const float4 dy2 = ld4(Cs + r * LDC + 4 * sub);
            const float4 dxb = bdrop(a.dc, seed, step, g, SITE_BLOCK, e0, dxr.v[i]);      // residual path of x1 -> x2
            const float4 dx1 = f4add(lnb_bwd(dy2, xr.v[i], gam, dgam, dbet), dxb);
            st4(a.dx1 + tr.grow(r) * BD + 4 * sub, dx1);
            t = bdrop(a.dc, seed, step, g, SITE_SUB_IN, e0, dx1);
            st4(a.dt + tr.grow(r) * BD + 4 * sub, t);
        }
        st4(As + r * LDK + 4 * sub, t);
    }
    __syncthreads();
    w_to_lds<BD, BD>(Ws, wr);
    __syncthreads();
    zero_acc<BD>(acc);
    mma_tile<BD, BD>(As, Ws, acc);
    acc_to_global_rows<BD>(a.d_o, rowf, nrows, BD, nullptr, acc);
    __syncthreads();
    ln_part_out<BD>(As, dgam, dbet, a.ln_part + (long long)tr.slot * 2 * BD);
}

struct BQkvBwdArgs {
    const float* dq; const float* dk; const float* dv; const float* dx1; const float* x; const float* la[2];
    const float* wT[3][2];
    float* dx; float* ln_part; BGeom tg;
};
__global__ __launch_bounds__(GEMM_THREADS) void bert_qkv_bwd_kernel(const BQkvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using RP = RowPass<BD>;
    constexpr int LDC = BD + 4;
    float* As = smem;
    float* Ws = smem + TileCfg<BD>::A_FLOATS;
    float* Cs = Ws;
    const BTile tr = btile_bwd(a.tg, blockIdx.x);
    const int g = tr.g, nrows = tr.nrows;
    auto rowf = [&](int r) { return tr.grow(r); };
    const int sub = RP::sub();
    const float* src[3] = {a.dq, a.dk, a.dv};
    TileRegs<BD> ar, xr, rr;
    WRegs<BD, BD> wr;
    load_tile_rows<BD>(ar, src[0], rowf, nrows, BD);
    load_w<BD, BD>(wr, a.wT[0][g], BD);
    f32x4 acc[WaveMap<BD>::ACC];
    zero_acc<BD>(acc);
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        if (s) __syncthreads();
        tile_to_lds<BD>(As, ar);
        w_to_lds<BD, BD>(Ws, wr);
        __syncthreads();
        if (s < 2) {
            load_tile_rows<BD>(ar, src[s + 1], rowf, nrows, BD);
            load_w<BD, BD>(wr, a.wT[s + 1][g], BD);
        }
        mma_tile<BD, BD>(As, Ws, acc);
    }
    load_tile_rows<BD>(xr, a.x, rowf, nrows, BD);
    load_tile_rows<BD>(rr, a.dx1, rowf, nrows, BD);
    __syncthreads();
    acc_to_lds<BD>(Cs, LDC, acc);
    __syncthreads();
    float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = dgam;
    const float4 gam = ld4(a.la[g] + 4 * sub);
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = RP::first_row() + i * RP::RPP;
        if (r < nrows) {
            const float4 dy = ld4(Cs + r * LDC + 4 * sub);
            st4(a.dx + tr.grow(r) * BD + 4 * sub, f4add(lnb_bwd(dy, xr.v[i], gam, dgam, dbet), rr.v[i]));
        }
    }
    __syncthreads();
    ln_part_out<BD>(As, dgam, dbet, a.ln_part + (long long)tr.slot * 2 * BD);
}

// ------------------------------------------------------------------------------------------------ weight gradients
// generic 128 x 128 output tiles: dW_tile[n][k] = sum_m dY[m][yoff + n] X[m][xoff + k], db_tile[n] = sum_m dY[m][yoff + n]
constexpr int BW_MAX = 12;
struct BWgradArgs {
    const float* dy[BW_MAX]; const float* x[BW_MAX];
    int ldy[BW_MAX], ldx[BW_MAX];                           // row strides (column offsets are folded into the pointers)
    int ldw[BW_MAX], wgrp[BW_MAX], wcol[BW_MAX];            // output placement: tiles of one wide matrix share a group (see the C entry)
    float* w_part;                                          // [2][n_ent][splits][128*128]
    float* b_part;                                          // [2][n_ent][splits][128]
    int n_ent, M, splits, rows_per_split;
    const long long* row_domain; int B, T;                  // optional hint, as WgradArgs::row_domain (sasrec_bwd.hip): walk the live sequences only
};
constexpr int BWG_ROWS = 64;
[[maybe_unused]] constexpr int BWG_LIVE_MAX = 1024;
__global__ __launch_bounds__(GEMM_THREADS) void bert_wgrad_kernel(const BWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = BD, LD = D + 16;
    float* Ys = smem;
    float* Xs = smem + BWG_ROWS * LD;
    const int split = blockIdx.x, e = blockIdx.y, g = blockIdx.z;
    const float* __restrict__ dy = a.dy[e];
    const float* __restrict__ xin = a.x[e];
    const int ldy = a.ldy[e], ldx = a.ldx[e];
    int local_beg = split * a.rows_per_split, local_end = min(a.M, local_beg + a.rows_per_split);
    const int w = wave_id(), lane = lane_id();
    const int nt = w, i = lane & 15, gq = lane >> 4;
    int* live = reinterpret_cast<int*>(smem + 2 * BWG_ROWS * LD);        // [B] (hint only)
    const bool hint = a.row_domain != nullptr;
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
    constexpr int QPR = D / 4, RPP = GEMM_THREADS / QPR, NRW = BWG_ROWS / RPP;
    const int sub = threadIdx.x % QPR, rl = threadIdx.x / QPR;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 py[NRW], px[NRW];
    auto fetch = [&](int c0) {
        const int nr = min(BWG_ROWS, local_end - c0);
        const long long grow = (long long)g * a.M + c0;
#pragma unroll
        for (int j = 0; j < NRW; ++j) {
            const int r = rl + j * RPP;
            const bool ok = r < nr;
            long long row = grow + r;
            if (hint && ok) {
                const int v = c0 + r, sq = v / a.T;
                row = (long long)g * a.M + (long long)live[sq - sq0] * a.T + (v - sq * a.T);
            }
            py[j] = ok ? ld4(dy + row * ldy + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
            px[j] = ok ? ld4(xin + row * ldx + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (local_beg < local_end) fetch(local_beg);
    for (int c0 = local_beg; c0 < local_end; c0 += BWG_ROWS) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NRW; ++j) {
            const int r = rl + j * RPP;
            bsum = f4add(bsum, py[j]);
            st4(Ys + r * LD + 4 * sub, py[j]);
            st4(Xs + r * LD + 4 * sub, px[j]);
        }
        __syncthreads();
        if (c0 + BWG_ROWS < local_end) fetch(c0 + BWG_ROWS);
        // branch-free, operands of step ms + 1 read under the MFMAs of step ms (same loop as sas_wgrad_kernel)
        const float* yp = Ys + gq * LD + nt * 16 + i;
        const float* xp = Xs + gq * LD + i;
        float a_cur = yp[0], x_cur[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) x_cur[t] = xp[t * 16];
#pragma unroll
        for (int ms = 0; ms < BWG_ROWS / 4; ++ms) {
            float a_nxt = a_cur, x_nxt[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) x_nxt[t] = x_cur[t];
            if (ms + 1 < BWG_ROWS / 4) {
                a_nxt = yp[(ms + 1) * 4 * LD];
#pragma unroll
                for (int t = 0; t < 8; ++t) x_nxt[t] = xp[(ms + 1) * 4 * LD + t * 16];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = mfma16(a_cur, x_cur[t], acc[t]);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            a_cur = a_nxt;
#pragma unroll
            for (int t = 0; t < 8; ++t) x_cur[t] = x_nxt[t];
        }
    }
    const int ldw = a.ldw[e];
    float* wp = a.w_part + ((long long)g * a.n_ent + a.wgrp[e]) * a.splits * D * D + (long long)split * ldw * D + a.wcol[e];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[(long long)(nt * 16 + gq * 4 + r) * ldw + t * 16 + i] = acc[t][r];
    __syncthreads();
    st4(Ys + rl * D + 4 * sub, bsum);
    __syncthreads();
    float* bp = a.b_part + (((long long)g * a.n_ent + e) * a.splits + split) * D;
    for (int k = threadIdx.x; k < D; k += GEMM_THREADS) {
        float s = 0.f;
#pragma unroll 4
        for (int j = 0; j < RPP; ++j) s += Ys[j * D + k];
        bp[k] = s;
    }
}

#if AMID_TILE_RT == 7
// the same tiles on the bf16 matrix cores at fp32 accuracy (csrc/wgrad_split.h): mode 2 / 3 of amid_bert_wgrad_mode_f32
template <int NTERM, bool HINT>
__global__ __launch_bounds__(GEMM_THREADS, 4) void bert_wgrad_split_kernel(const BWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int D = BD;
    const int split = blockIdx.x, e = blockIdx.y, g = blockIdx.z;
    const WgsRows rw{a.M, a.splits, a.rows_per_split, a.row_domain, a.B, a.T};
    f32x4 acc[8];
    wgrad_split_tile<NTERM, HINT>(smem, a.dy[e], a.ldy[e], a.x[e], a.ldx[e], g, split, rw, acc,
                                  a.b_part + (((long long)g * a.n_ent + e) * a.splits + split) * D);
    const int w = wave_id(), lane = lane_id(), i = lane & 15, gq = lane >> 4;
    const int ldw = a.ldw[e];
    float* wp = a.w_part + ((long long)g * a.n_ent + a.wgrp[e]) * a.splits * D * D + (long long)split * ldw * D + a.wcol[e];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[(long long)(w * 16 + gq * 4 + r) * ldw + t * 16 + i] = acc[t][r];
}
#endif

// out[c][r] = in[r][c] for rectangular matrices (rows, cols multiples of 32)
struct BTransArgs { const float* src[32]; float* dst[32]; int rows[32], cols[32]; int n; };
__global__ __launch_bounds__(256) void transpose_rect_kernel(const BTransArgs a) {
    __shared__ float tile[32][33];
    const int m = blockIdx.z;
    const int R = a.rows[m], C = a.cols[m];
    const int tilesx = C / 32, tiles = tilesx * (R / 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int tt = blockIdx.x; tt < tiles; tt += gridDim.x) {
        const int bx = (tt % tilesx) * 32, by = (tt / tilesx) * 32;
        __syncthreads();
        for (int r = ty; r < 32; r += 8) tile[r][tx] = a.src[m][(long long)(by + r) * C + bx + tx];
        __syncthreads();
        for (int r = ty; r < 32; r += 8) a.dst[m][(long long)(bx + r) * R + by + tx] = tile[tx][r];
    }
}

}  // namespace amid

using namespace amid;

static constexpr size_t bert_lds() { return (size_t)(TileCfg<BD>::A_FLOATS + TileCfg<BD>::W_FLOATS) * sizeof(float); }
static int bgeom(int M, int rpt, BGeom* tg) {
    if (M <= 0 || rpt <= 0 || rpt > TILE_ROWS) return AMID_ERR_ARG;
    tg->M = M; tg->rows_per_tile = rpt; tg->tiles_per_group = (M + rpt - 1) / rpt;
    return AMID_OK;
}
static DropCfg bdropcfg(const void* st, int train, float p, int layer) {
    DropCfg d;
    d.st = (const StepState*)st;
    d.train = (train && p > 0.f) ? 1 : 0;
    d.spec = drop_spec(p);
    d.scale = d.train ? 1.0f / (1.0f - p) : 1.0f;
    d.layer = layer;
    return d;
}
#define BERT_LAUNCH(KERNEL, ARGS)                                                                                   \
    do {                                                                                                            \
        static bool attr_set = false;                                                                               \
        if (!attr_set) {                                                                                            \
            hipError_t e = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bert_lds()); \
            if (e != hipSuccess) return (int)e;                                                                     \
            attr_set = true;                                                                                        \
        }                                                                                                           \
        KERNEL<<<2 * ARGS.tg.tiles_per_group, GEMM_THREADS, bert_lds(), (hipStream_t)stream>>>(ARGS);               \
        AMID_LAUNCH_CHECK();                                                                                        \
    } while (0)

// Pointer-array parameters are HOST arrays of device pointers indexed [which * 2 + domain] or [domain].
extern "C" int AMID_ENTRY(amid_bert_qkv_fwd_f32)(const float* x, const float* const* ln_a, const float* const* ln_b, const float* const* w3x2,
                                     const float* const* b3x2, int M, int rows_per_tile, float* y, float* q, float* k, float* v, void* stream) {
    AMID_CHECK_ARG(x && ln_a && ln_b && w3x2 && b3x2 && y && q && k && v);
    BQkvArgs a;
    a.x = x; a.y = y; a.out[0] = q; a.out[1] = k; a.out[2] = v;
    for (int g = 0; g < 2; ++g) { a.la[g] = ln_a[g]; a.lb[g] = ln_b[g]; for (int j = 0; j < 3; ++j) { a.w[j][g] = w3x2[j * 2 + g]; a.b[j][g] = b3x2[j * 2 + g]; } }
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_qkv_fwd_kernel, a);
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_bert_oproj_fwd_f32)(const float* o, const float* x, const float* const* w, const float* const* b, int M, int rows_per_tile,
                                       int layer, const void* step_state, int train, float p_drop, float* x1, void* stream) {
    AMID_CHECK_ARG(o && x && w && b && x1 && (!train || step_state));
    BOprojArgs a;
    a.o = o; a.x = x; a.x1 = x1; a.dc = bdropcfg(step_state, train, p_drop, layer);
    for (int g = 0; g < 2; ++g) { a.w[g] = w[g]; a.b[g] = b[g]; }
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_oproj_fwd_kernel, a);
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_bert_ffn1_fwd_f32)(const float* x1, const float* const* ln_a, const float* const* ln_b, const float* const* w1,
                                      const float* const* b1, int M, int rows_per_tile, int layer, const void* step_state, int train,
                                      float p_drop, float* y2, float* pre, float* h, void* stream) {
    AMID_CHECK_ARG(x1 && ln_a && ln_b && w1 && b1 && y2 && pre && h && (!train || step_state));
    BFfn1Args a;
    a.x1 = x1; a.y2 = y2; a.pre = pre; a.h = h; a.dc = bdropcfg(step_state, train, p_drop, layer);
    for (int g = 0; g < 2; ++g) { a.la[g] = ln_a[g]; a.lb[g] = ln_b[g]; a.w1[g] = w1[g]; a.b1[g] = b1[g]; }
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_ffn1_fwd_kernel, a);
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_bert_ffn2_fwd_f32)(const float* h, const float* x1, const float* const* w2, const float* const* b2, int M, int rows_per_tile,
                                      int layer, const void* step_state, int train, float p_drop, float* x2, void* stream) {
    AMID_CHECK_ARG(h && x1 && w2 && b2 && x2 && (!train || step_state));
    BFfn2Args a;
    a.h = h; a.x1 = x1; a.x2 = x2; a.dc = bdropcfg(step_state, train, p_drop, layer);
    for (int g = 0; g < 2; ++g) { a.w2[g] = w2[g]; a.b2[g] = b2[g]; }
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_ffn2_fwd_kernel, a);
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_bert_ffn2_bwd_f32)(const float* dx2, const float* pre, const float* const* w2T, int M, int rows_per_tile, int layer,
                                      const void* step_state, int train, float p_drop, float* dz, float* dpre, void* stream) {
    AMID_CHECK_ARG(dx2 && pre && w2T && dz && dpre && (!train || step_state));
    BFfn2BwdArgs a;
    a.dx2 = dx2; a.pre = pre; a.dz = dz; a.dpre = dpre; a.dc = bdropcfg(step_state, train, p_drop, layer);
    for (int g = 0; g < 2; ++g) a.w2T[g] = w2T[g];
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_ffn2_bwd_kernel, a);
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_bert_ffn1_bwd_f32)(const float* dpre, const float* dx2, const float* x1, const float* const* ln_a, const float* const* w1T,
                                      const float* const* woT, int M, int rows_per_tile, int layer, const void* step_state, int train,
                                      float p_drop, float* dx1, float* dt, float* d_o, float* ln_part, void* stream) {
    AMID_CHECK_ARG(dpre && dx2 && x1 && ln_a && w1T && woT && dx1 && dt && d_o && ln_part && (!train || step_state));
    BFfn1BwdArgs a;
    a.dpre = dpre; a.dx2 = dx2; a.x1 = x1; a.dx1 = dx1; a.dt = dt; a.d_o = d_o; a.ln_part = ln_part;
    a.dc = bdropcfg(step_state, train, p_drop, layer);
    for (int g = 0; g < 2; ++g) { a.la[g] = ln_a[g]; a.w1T[g] = w1T[g]; a.woT[g] = woT[g]; }
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_ffn1_bwd_kernel, a);
    return AMID_OK;
}

extern "C" int AMID_ENTRY(amid_bert_qkv_bwd_f32)(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x, const float* const* ln_a,
                                     const float* const* wT3x2, int M, int rows_per_tile, float* dx, float* ln_part, void* stream) {
    AMID_CHECK_ARG(dq && dk && dv && dx1 && x && ln_a && wT3x2 && dx && ln_part);
    BQkvBwdArgs a;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dx1 = dx1; a.x = x; a.dx = dx; a.ln_part = ln_part;
    for (int g = 0; g < 2; ++g) { a.la[g] = ln_a[g]; for (int j = 0; j < 3; ++j) a.wT[j][g] = wT3x2[j * 2 + g]; }
    if (int e = bgeom(M, rows_per_tile, &a.tg)) return e;
    BERT_LAUNCH(bert_qkv_bwd_kernel, a);
    return AMID_OK;
}

// n_ent (<= 12) output tiles of 128 x 128; dy / x: host arrays of n_ent device pointers (column offsets folded in), ld*: row strides.
// Output placement: tile e of domain g, split s goes to w_part[g][out_group[e]][s] at column out_col[e] with row stride out_ld[e]; a
// standalone tile has (out_ld, out_group, out_col) = (128, e, 0); the out_ld/128 tiles forming one [128, out_ld] matrix (w_2) share
// out_group = their first entry, so that matrix's partials are contiguous: [splits][128 * out_ld] starting at entry out_group.
#if AMID_TILE_RT == 7      // independent of the row-tile height: one copy only
static int bert_wgrad(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                      const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part, float* b_part,
                      const long long* row_domain, int B, int T, void* stream, int mode = 0) {
    AMID_CHECK_ARG(dy && x && ldy && ldx && out_ld && out_group && out_col && w_part && b_part && n_ent > 0 && n_ent <= BW_MAX && M > 0 &&
                   splits > 0);
    BWgradArgs a;
    for (int i = 0; i < n_ent; ++i) {
        AMID_CHECK_ARG(dy[i] && x[i] && out_ld[i] >= BD && out_ld[i] % BD == 0 && out_group[i] >= 0 &&
                       out_group[i] + out_ld[i] / BD <= n_ent && out_col[i] >= 0 && out_col[i] + BD <= out_ld[i]);
        a.dy[i] = dy[i]; a.x[i] = x[i]; a.ldy[i] = ldy[i]; a.ldx[i] = ldx[i];
        a.ldw[i] = out_ld[i]; a.wgrp[i] = out_group[i]; a.wcol[i] = out_col[i];
    }
    a.w_part = w_part; a.b_part = b_part; a.n_ent = n_ent; a.M = M; a.splits = splits; a.rows_per_split = (M + splits - 1) / splits;
    AMID_CHECK_ARG(!row_domain || (B > 0 && T > 0 && (long long)B * T == M));
    a.row_domain = (row_domain && a.rows_per_split / T + 2 <= BWG_LIVE_MAX) ? row_domain : nullptr; a.B = B; a.T = T;
    if (mode == 2 || mode == 3) {          // fp32 operands as three bf16 pieces each, nine / six piece pairs
        static_assert(BWG_LIVE_MAX == WGS_LIVE_MAX, "one window size");
        const size_t lds = WGS_LDS_FIXED + WGS_LIVE_MAX * sizeof(int);
        static unsigned long long done[4] = {0, 0, 0, 0};
        const dim3 grid(splits, n_ent, 2);
#define AMID_BWGS_LAUNCH(NT, H, SLOT)                                                                                              \
        do {                                                                                                                   \
            if (int e = lds_attr_once((const void*)bert_wgrad_split_kernel<NT, H>, lds, done[SLOT])) return e;                    \
            bert_wgrad_split_kernel<NT, H><<<grid, GEMM_THREADS, lds, (hipStream_t)stream>>>(a);                                   \
        } while (0)
        const bool h = a.row_domain != nullptr;
        if (mode == 2) { if (h) AMID_BWGS_LAUNCH(9, true, 0); else AMID_BWGS_LAUNCH(9, false, 1); }
        else { if (h) AMID_BWGS_LAUNCH(6, true, 2); else AMID_BWGS_LAUNCH(6, false, 3); }
#undef AMID_BWGS_LAUNCH
        AMID_LAUNCH_CHECK();
        return AMID_OK;
    }
    if (mode != 0) return AMID_ERR_ARG;
    const size_t lds = (size_t)2 * BWG_ROWS * (BD + 16) * sizeof(float) + BWG_LIVE_MAX * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)bert_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    bert_wgrad_kernel<<<dim3(splits, n_ent, 2), GEMM_THREADS, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_bert_wgrad_f32(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                                   const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part, float* b_part,
                                   void* stream) {
    return bert_wgrad(dy, x, ldy, ldx, out_ld, out_group, out_col, n_ent, M, splits, w_part, b_part, nullptr, 0, 0, stream);
}

// the live sequences only, as amid_sas_wgrad_rows_f32 (M = B * T, row_domain [B] = the batch's domain ids)
extern "C" int amid_bert_wgrad_rows_f32(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                                        const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part,
                                        float* b_part, const long long* row_domain, int B, int T, void* stream) {
    return bert_wgrad(dy, x, ldy, ldx, out_ld, out_group, out_col, n_ent, M, splits, w_part, b_part, row_domain, B, T, stream);
}

// either of the two with the products' mode: 0 = fp32 matrix instructions (as above), 2 / 3 = every fp32 operand as three bf16 pieces,
// nine / six piece pairs on v_mfma_f32_16x16x32_bf16 (csrc/wgrad_split.h); row_domain may be NULL (every row walked)
extern "C" int amid_bert_wgrad_mode_f32(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                                        const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part,
                                        float* b_part, const long long* row_domain, int B, int T, int mode, void* stream) {
    return bert_wgrad(dy, x, ldy, ldx, out_ld, out_group, out_col, n_ent, M, splits, w_part, b_part, row_domain, B, T, stream, mode);
}

extern "C" int amid_transpose_rect_f32(const float* const* src, float* const* dst, const int* rows, const int* cols, int n, void* stream) {
    AMID_CHECK_ARG(src && dst && rows && cols && n > 0 && n <= 32);
    BTransArgs a;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        AMID_CHECK_ARG(src[i] && dst[i] && rows[i] > 0 && cols[i] > 0 && rows[i] % 32 == 0 && cols[i] % 32 == 0);
        a.src[i] = src[i]; a.dst[i] = dst[i]; a.rows[i] = rows[i]; a.cols[i] = cols[i];
    }
    transpose_rect_kernel<<<dim3(16, 1, n), 256, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
#endif
