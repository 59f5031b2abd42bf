// BERT4Rec encoder block as register-resident strip chains (strip_gemm.h / strip_chain.h), forward and backward.
// Reference arithmetic: TransformerBlock.forward model_seq.py:242-245 with SublayerConnection :140-142, the hand-written LayerNorm
// :124-127 (UNBIASED std, eps added to the std), MultiHeadedAttention's projections :183-196, PositionwiseFeedForward :216-217 with the
// tanh GELU :204 -- hidden 128 / 4 heads / feed-forward 512 / dropout 0.1 hard-coded by the reference (:264-267) -- and their autograd
// (loss.backward(), train_sr.py:214).  Same operations, operands, dropout counters and saved tensors as the row-tile kernels of bert.hip
// (which stay for activations beyond 2 GiB); the attention core between the projections runs in its own launch
// (attention_mfma_bert.hip).
//
//   bert_strip_qkv_fwd        y = LNb_in(x) ; q, k, v = y W{0,1,2}^T + b                                     (all three from the normed y)
//   bert_strip_oproj_ffn_fwd  x1 = x + drop_in(o Wo^T + bo) ; y2 = LNb_out(x1) ; per 128-column chunk c of the 512 hidden units:
//                             pre_c = y2 W1_c^T + b1_c, h_c = drop_ffn(gelu(pre_c)), z += h_c W2_c^T ;
//                             x2 = drop_block(x1 + drop_out(z + b2))        [+ the next block's bert_strip_qkv_fwd on x2 in registers]
//   bert_strip_ffn_bwd        dr = dx2 * drop_block ; dz = dr * drop_out ; per chunk: dpre_c = (dz W2_c) * drop_ffn * gelu'(pre_c),
//                             dy2 += dpre_c W1_c ; dx1 = LNb_out'(dy2 ; x1) + dr ; dt = dx1 * drop_in ; d_o = dt Wo
//   bert_strip_qkv_bwd        dx = LNb_in'(dq Wq + dk Wk + dv Wv ; x) + dx1      [+ the block below's bert_strip_ffn_bwd on dx in registers]
// A wave owns 16 rows and all 128 columns; a workgroup is 4 waves = 64 rows of one domain; in a train step only the LIVE sequences
// are walked (StripGeom::live).  The 128 x 128 weight tiles stream through the two-slab LDS ring by LDS-DMA; the tiles of w_2
// [128, 512] and of the transposed w_1 [128, 512] are column blocks of a 512-float row (WDmaLd).  Backward weights arrive TRANSPOSED
// (amid_transpose_rect_f32) so that a data gradient is again C[rows, N] = A[rows, K] W'[N, K]^T.  2 D D FLOP per row and tile.
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "strip_chain.h"
#include "bert_math.h"
#include "weights_image.h"

namespace amid {

constexpr int BSD = 128;            // hidden
constexpr int BSF = 512;            // feed-forward
constexpr int BSC = BSF / BSD;      // 128-column chunks of the feed-forward
constexpr int BNT = BSD / 16;

// strip_gemm.h's WDma for a tile whose rows lie LD floats apart in global memory (a column block of a wider matrix)
template <int D, int LD> struct WDmaLd {
    static constexpr int CPR = D / 4;
    static constexpr int PER_WAVE = D * CPR / 64 / STRIP_WAVES;
    static constexpr unsigned STRIDE2 = 2u * (256 / CPR) * LD * 4;
    unsigned off[2];
    int w;
    __device__ __forceinline__ WDmaLd() {
        const int lane = lane_id();
        w = wave_id();
#pragma unroll
        for (int k0 = 0; k0 < 2; ++k0) {
            const int p = (k0 * STRIP_WAVES + w) * 64 + lane;
            const int n = p / CPR, pos = p % CPR;
            off[k0] = (unsigned)((n * LD + ((pos ^ (n & 15)) * 4)) * 4);
        }
    }
    __device__ __forceinline__ void piece(float* __restrict__ buf, const float* __restrict__ W, int k0) const {
        const unsigned voff = off[k0 & 1] + (unsigned)(k0 >> 1) * STRIDE2;
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(buf + (k0 * STRIP_WAVES + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
};

// the two-slab ring of strip_chain.h with a second source stride: fetch() takes a [128][128] tile with contiguous rows, fetch_ld() a
// column block of a [128][512] matrix
struct BRing {
    static constexpr bool BF16 = false;
    static constexpr int SLAB = BSD * BSD;
    float* buf; int s; WDma<BSD> dma; WDmaLd<BSD, BSF> dml;
    __device__ __forceinline__ explicit BRing(float* lds) : buf(lds), s(0) {}
    __device__ __forceinline__ void first(const float* __restrict__ W0) { dma.all(buf, W0); }
    __device__ __forceinline__ const float* next() {
        w_ring_wait();
        __syncthreads();
        const float* cur = buf + (s & 1) * SLAB;
        ++s;
        return cur;
    }
    static constexpr int SLOTS = 8 * BNT, EVERY = (SLOTS / 2) / WDma<BSD>::PER_WAVE;
    __device__ __forceinline__ void fetch(const float* __restrict__ W, int ct, int j) const {
        const int slot = ct * 8 + j;
        if (slot % EVERY == 0 && slot / EVERY < WDma<BSD>::PER_WAVE) dma.piece(buf + (s & 1) * SLAB, W, slot / EVERY);
    }
    __device__ __forceinline__ void fetch_ld(const float* __restrict__ W, int ct, int j) const {
        const int slot = ct * 8 + j;
        if (slot % EVERY == 0 && slot / EVERY < WDma<BSD>::PER_WAVE) dml.piece(buf + (s & 1) * SLAB, W, slot / EVERY);
    }
};

// MODE 3 (round 5): the products on bf16 pieces -- fp32 operands as hi + mid + lo, six piece pairs of v_mfma_f32_16x16x32_bf16, fp32
// accuracy (strip_gemm.h strip_mma16x6, strip_chain.h RingP3; what SASRec's strips run on since round 4).  Every 128 x 128 weight TILE
// the chains multiply with is then a three-plane fragment image (amid_bert_weight_images_f32): the weight arguments of the kernels point
// at images, a feed-forward weight at its four tiles' images one behind the other.
#ifndef AMID_BS_SPREAD
#define AMID_BS_SPREAD 1
#endif
constexpr bool BSPREAD = AMID_BS_SPREAD != 0;      // the hooks' slots between the piece products' matrix instructions (strip_gemm.h strip_mma16x6 SPREAD)
constexpr int BIMG = 3 * (BSD * BSD / 2);            // floats per tile image (three 32 KB planes)
template <int MODE> struct BRingSel { using type = BRing; };
template <> struct BRingSel<3> { using type = RingP3<BSD>; };
// tile c of a feed-forward weight whose tiles are ROW blocks of the fp32 matrix (w_1 [512][128], w_2^T [512][128]) ...
template <class R> __device__ __forceinline__ const float* btile_rows(const float* base, int c) {
    return base + (long long)c * (ring_is_p3<R>::value ? BIMG : BSD * BSD);
}
// ... and whose tiles are COLUMN blocks (w_2 [128][512], w_1^T [128][512]): fetched with the wide stride, or as the c-th image
template <class R> __device__ __forceinline__ void bfetch_cols(R& ring, const float* base, int c, int ct, int j) {
    if constexpr (ring_is_p3<R>::value) ring.fetch(base + (long long)c * BIMG, ct, j);
    else ring.fetch_ld(base + c * BSD, ct, j);
}

// ---- the reference LayerNorm on a strip: a (x - mean) / (std_unbiased + eps) + b -----------------------------------------------------
__device__ __forceinline__ void lnb_stats(const StripRegs<BSD>& x, float& mean, float& sd, float& r) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct) s += (x.v[ct][0] + x.v[ct][1]) + (x.v[ct][2] + x.v[ct][3]);
    mean = row_sum4(s) * (1.0f / BSD);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = x.v[ct][e] - mean; q = fmaf(d, d, q); }
    sd = sqrtf(row_sum4(q) * (1.0f / (BSD - 1)));
    r = 1.0f / (sd + BERT_EPS);
}
__device__ __forceinline__ void strip_lnb(StripRegs<BSD>& y, const StripRegs<BSD>& x, const ColVec<BSD>& a, const ColVec<BSD>& b) {
    float mean, sd, r;
    lnb_stats(x, mean, sd, r);
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) y.v[ct][e] = a.v[ct][e] * (x.v[ct][e] - mean) * r + b.v[ct][e];
}
// backward: dx = r (g - mean(g)) - t r^2 xc / (std (D - 1)), g = a dy, t = sum(g xc); this lane's row adds dy xc r / dy to the column partials
__device__ __forceinline__ void strip_lnb_bwd(StripRegs<BSD>& dx, const StripRegs<BSD>& dy, const StripRegs<BSD>& x, const ColVec<BSD>& a,
                                              StripRegs<BSD>& dgam, StripRegs<BSD>& dbet) {
    float mean, sd, r;
    lnb_stats(x, mean, sd, r);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xc = x.v[ct][e] - mean;
            const float gg = a.v[ct][e] * dy.v[ct][e];
            s1 += gg;
            s2 = fmaf(gg, xc, s2);
            dgam.v[ct][e] = dy.v[ct][e] * xc * r;
            dbet.v[ct][e] = dy.v[ct][e];
            dx.v[ct][e] = gg;
        }
    const float gm = row_sum4(s1) * (1.0f / BSD);
    const float t = row_sum4(s2);
    const float c = (sd > 0.f) ? t * r * r / (sd * (BSD - 1)) : 0.f;
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) dx.v[ct][e] = r * (dx.v[ct][e] - gm) - c * (x.v[ct][e] - mean);
}

// ---- dropout on a strip: the keep bits of this lane's 32 elements (bit 4 ct + e <-> column 16 ct + 4 g + e) -------------------------------
// p = 0.1 takes 16-bit decisions (rng.h): ONE Philox call decides 8 consecutive elements = the column quads of lanes (m, 2h) and
// (m, 2h + 1) of one column tile.  Lane (m, g) draws the calls of the column tiles ct with (ct & 1) == (g & 1), keeps its own quad's
// four decisions and hands the partner (lane ^ 16) the other four: four calls per lane and site instead of eight.  And the calls are
// drawn INSIDE the matrix loop in front of the epilogue that applies them, a round per MFMA group (slot()): the counters do not depend
// on data, a round is two quarter-rate multiplies + five plain instructions, a group's four matrix instructions cover them.
// (measured, cfg 2: with every lane drawing its eight calls ahead of the loop the counters cost 87 us of a 0.725 ms step)
struct BDrop { int train; unsigned spec; float scale; unsigned long long seed; unsigned step; int layer; };
struct KeepGen {
    uint4 c; unsigned k0, k1, own, give;
    unsigned long long call0; unsigned site, step, key0, key1, thr; int godd, train;
    // e_row: the row's first element (a multiple of 128: call-aligned)
    __device__ __forceinline__ void begin(const BDrop& d, int g, int kind, unsigned long long e_row) {
        const int gq = lane_id() >> 4;
        godd = gq & 1;
        call0 = (e_row >> 3) + (unsigned)(gq >> 1);            // call of column tile ct: call0 + 2 ct
        site = site_id(g, d.layer, kind); step = d.step; key0 = (unsigned)d.seed; key1 = (unsigned)(d.seed >> 32);
        thr = spec_thr(d.spec); train = d.train;
        own = 0u; give = 0u;
    }
    // slot s = 0 .. 63 of an MFMA loop (8 ct + j): call i = s / 16 in phases s % 16 = 0 (counter), 1 .. 10 (rounds), 11 (decisions)
    __device__ __forceinline__ void slot(int s) {
        const int i = s >> 4, ph = s & 15;
        const int ct = 2 * i + godd;
        if (ph == 0) {
            const unsigned long long call = call0 + 2u * (unsigned)ct;
            c = make_uint4((unsigned)call, (unsigned)(call >> 32), site, step);
            k0 = key0; k1 = key1;
        } else if (ph <= 10) {
#ifdef AMID_BS_ABLATE_PHILOX          // timing-only diagnostic build (profiles/tools/probe/bert_ablate.sh): what the rounds cost
            return;
#endif
            const unsigned long long p0 = mul_wide(0xD2511F53u, c.x), p1 = mul_wide(0xCD9E8D57u, c.z);
            c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        } else if (ph == 11) {
            const unsigned lo4 = ((c.x & 0xFFFFu) >= thr ? 1u : 0u) | ((c.x >> 16) >= thr ? 2u : 0u) | ((c.y & 0xFFFFu) >= thr ? 4u : 0u) | ((c.y >> 16) >= thr ? 8u : 0u);
            const unsigned hi4 = ((c.z & 0xFFFFu) >= thr ? 1u : 0u) | ((c.w >> 16) >= thr ? 8u : 0u) | ((c.z >> 16) >= thr ? 2u : 0u) | ((c.w & 0xFFFFu) >= thr ? 4u : 0u);
            own |= (godd ? hi4 : lo4) << (4 * ct);
            give |= (godd ? lo4 : hi4) << (4 * ct);
        }
    }
    __device__ __forceinline__ void hook(int ct, int j) { slot(ct * 8 + j); }
    // all four calls at once (no loop to hide them in)
    __device__ __forceinline__ void all() {
#pragma unroll
        for (int s = 0; s < 64; ++s) slot(s);
    }
    __device__ __forceinline__ unsigned finish() const {
        float a = __builtin_bit_cast(float, give), b = a;
        swap16(a, b);                                           // a: (r0, r0, r2, r2), b: (r1, r1, r3, r3) of `give` by lane row
        const unsigned got = __builtin_bit_cast(unsigned, godd ? a : b);
        return train ? (own | got) : ~0u;
    }
};
__device__ __forceinline__ unsigned keep_bits(const BDrop& d, int g, int kind, unsigned long long e_row) {
    KeepGen kg;
    kg.begin(d, g, kind, e_row);
    kg.all();
    return kg.finish();
}
__device__ __forceinline__ void apply_keep(StripRegs<BSD>& x, unsigned bits, float scale) {
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e) x.v[ct][e] = ((bits >> (4 * ct + e)) & 1u) ? x.v[ct][e] * scale : 0.f;
}

// a [2M, 512] tensor (pre, h, dpre): this lane's 16 bytes of column tile 0 of chunk 0; + 512 c + 64 ct for the others
__device__ __forceinline__ unsigned wide_off(const StripRow& row) {
    return row.ok ? row.off * 4u - 48u * (unsigned)(lane_id() >> 4) : STRIP_OOB;
}
__device__ __forceinline__ void wide_load(StripRegs<BSD>& x, const GBuf& g, unsigned offw, int c) {
#pragma unroll
    for (int ct = 0; ct < BNT; ++ct) x.v[ct] = g.load4(offw + c * (BSD * 4) + ct * 64);
}
// one column tile of chunk c per call, from inside an MFMA loop (first half of the loop, groups j == phase mod 4: as store_spread)
__device__ __forceinline__ void wide_spread(const GBuf& g, unsigned offw, int c, const StripRegs<BSD>& x, int ct, int j, int phase) {
    if (ct < BNT / 2 && (j & 3) == phase) { const int t = 2 * ct + (j >> 2); g.store4(offw + c * (BSD * 4) + t * 64, x.v[t]); }
}
__device__ __forceinline__ void spread_at(const GBuf& g, const StripRow& row, const StripRegs<BSD>& x, int ct, int j, int phase) {
    if (ct < BNT / 2 && (j & 3) == phase) strip_store_ct<BSD>(g, row, x, 2 * ct + (j >> 2));
}

// ================================================================================================================ forward
struct BStripQkvArgs {
    const float* x; const float* la[2]; const float* lb[2];
    const float* w[3][2]; const float* b[3][2];
    float* y; float* out[3];
};
struct BStripOffArgs {
    const float* o; const float* x;
    const float* wo[2]; const float* bo[2]; const float* la[2]; const float* lb[2];
    const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
    float* x1; float* y2; float* pre; float* h; float* x2;
    const StepState* st; int train; unsigned spec; float scale; int layer;
};

// q / k / v of one block on the strip X (in registers).  The ring's current fetch must be Wq of this block (started by the caller).
// XSTORE: X is also written to a.x (a fused predecessor produced it: the saved block input).
template <bool XSTORE, class R>
__device__ __forceinline__ void bqkv_fwd_chain(const BStripQkvArgs& a, const StripGeom& sg, R& ring, const StripRow& row, int g,
                                               const StripRegs<BSD>& X, const ColVec<BSD>& la, const ColVec<BSD>& lb) {
    const GBuf gx(a.x, sg.act_bytes), gy(a.y, sg.act_bytes);
    StripRegs<BSD> Y, P0, P1;
    ColVec<BSD> bias;
    strip_lnb(Y, X, la, lb);
    f32x4 acc[BNT];
    {   // q = y Wq^T + bq ; x's and y's global copies leave under these MFMAs
        const float* buf = ring.next();
        bias.load(a.b[0][g]);
        strip_zero<BSD>(acc);
        strip_product<BSD, BSPREAD>(acc, Y, buf, ring, [&](int ct, int j) {
            ring.fetch(a.w[1][g], ct, j);
            if constexpr (XSTORE) spread_at(gx, row, X, ct, j, 3);
            spread_at(gy, row, Y, ct, j, 1);
        });
        add_bias<BSD>(acc, bias);
        to_regs<BSD>(P0, acc);
    }
    {   // k
        const float* buf = ring.next();
        bias.load(a.b[1][g]);
        strip_zero<BSD>(acc);
        const GBuf gq(a.out[0], sg.act_bytes);
        strip_product<BSD, BSPREAD>(acc, Y, buf, ring, [&](int ct, int j) { ring.fetch(a.w[2][g], ct, j); spread_at(gq, row, P0, ct, j, 1); });
        add_bias<BSD>(acc, bias);
        to_regs<BSD>(P1, acc);
    }
    {   // v
        const float* buf = ring.next();
        bias.load(a.b[2][g]);
        strip_zero<BSD>(acc);
        const GBuf gk(a.out[1], sg.act_bytes);
        strip_product<BSD, BSPREAD>(acc, Y, buf, ring, [&](int ct, int j) { spread_at(gk, row, P1, ct, j, 1); });
        add_bias<BSD>(acc, bias);
        to_regs<BSD>(P0, acc);
        strip_store<BSD>(GBuf(a.out[2], sg.act_bytes), row, P0);
    }
}

// Riders of the step's FIRST strip launch (workgroups behind the tiles'; the live tiles of a train step leave a fifth of the CUs free):
// the key mask of both encoders (model_seq.py:288: seq_d2 > 0) for the attention launch that follows, and the transposed weights the
// backward's data gradients multiply with -- two launches of their own until now (amid_key_keep_u8, amid_transpose_rect_f32).
constexpr int BPRO_MAX = 24;
struct BPrologue {
    const long long* seq; unsigned char* keep; int n_keys;       // keep[i] = seq[i] > 0; seq == nullptr: none
    const float* src[BPRO_MAX]; float* dst[BPRO_MAX]; int rows[BPRO_MAX], cols[BPRO_MAX]; int n;      // dst[c][r] = src[r][c]; rows, cols multiples of 64
    int blocks;
};
__device__ __forceinline__ void bert_prologue_block(const BPrologue& p, int blk, float* __restrict__ lds) {
    if (p.seq != nullptr)
        for (int i = blk * STRIP_THREADS + threadIdx.x; i < p.n_keys; i += p.blocks * STRIP_THREADS) p.keep[i] = p.seq[i] > 0 ? 1 : 0;
    float (*tile)[65] = (float (*)[65])lds;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    int unit0 = 0;                                               // 64 x 64 units of all matrices, dealt round-robin over the rider blocks
    for (int m = 0; m < p.n; ++m) {
        const int R = p.rows[m], C = p.cols[m];
        const int tilesx = C / 64, tiles = tilesx * (R / 64);
        const int first = (blk - unit0 % p.blocks + p.blocks) % p.blocks;
        for (int tt = first; tt < tiles; tt += p.blocks) {
            const int bx = (tt % tilesx) * 64, by = (tt / tilesx) * 64;
            __syncthreads();
            float v[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) {                          // sixteen loads in flight per thread
                const float* sp = p.src[m] + (long long)(by + ty + 8 * k) * C + bx + tx;
                v[2 * k] = sp[0]; v[2 * k + 1] = sp[32];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { tile[ty + 8 * k][tx] = v[2 * k]; tile[ty + 8 * k][tx + 32] = v[2 * k + 1]; }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float* dp = p.dst[m] + (long long)(bx + ty + 8 * k) * R + by + tx;
                dp[0] = tile[tx][ty + 8 * k]; dp[32] = tile[tx + 32][ty + 8 * k];
            }
        }
        unit0 += tiles;
    }
}

template <int MODE>
__global__ __launch_bounds__(STRIP_THREADS) void bert_strip_qkv_fwd_kernel(const BStripQkvArgs a, const StripGeom sg, const BPrologue pro) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= 2 * sg.tpg) { bert_prologue_block(pro, blockIdx.x - 2 * sg.tpg, smem); return; }
    typename BRingSel<MODE>::type ring(smem);
    ring.first(a.w[0][strip_domain(blockIdx.x)]);
    const StripTile t = strip_tile(sg, blockIdx.x);
    if (!t.live) { w_ring_wait(); return; }
    const StripRow row = strip_row<BSD>(sg, t);
    StripRegs<BSD> X;
    ColVec<BSD> la, lb;
    strip_load<BSD>(X, GBuf(a.x, sg.act_bytes), row);
    la.load(a.la[t.g]); lb.load(a.lb[t.g]);
    bqkv_fwd_chain<false>(a, sg, ring, row, t.g, X, la, lb);
}

template <bool NEXT, int MODE>
__global__ __launch_bounds__(STRIP_THREADS) void bert_strip_oproj_ffn_fwd_kernel(const BStripOffArgs a, const BStripQkvArgs nx, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using R = typename BRingSel<MODE>::type;
    R ring(smem);
    ring.first(a.wo[strip_domain(blockIdx.x)]);
    const StripTile t = strip_tile(sg, blockIdx.x);
    if (!t.live) { w_ring_wait(); return; }
    const StripRow row = strip_row<BSD>(sg, t);
    const int g = t.g;
    BDrop dc = {a.train, a.spec, a.scale, 0ull, 0u, a.layer};
    if (a.train) { dc.seed = a.st->seed; dc.step = (unsigned)a.st->step; }
    const unsigned long long e128 = (unsigned long long)row.local * BSD, e512 = (unsigned long long)row.local * BSF;
    const unsigned wide_bytes = sg.act_bytes * 4u;
    const GBuf gx1(a.x1, sg.act_bytes), gy2(a.y2, sg.act_bytes), gpre(a.pre, wide_bytes), gh(a.h, wide_bytes);
    const unsigned offw = wide_off(row);
    StripRegs<BSD> A, X1, Y2, P, Hc;
    ColVec<BSD> bias, la, lb;
    strip_load<BSD>(A, GBuf(a.o, sg.act_bytes), row);
    strip_load<BSD>(X1, GBuf(a.x, sg.act_bytes), row);
    bias.load(a.bo[g]); la.load(a.la[g]); lb.load(a.lb[g]);
    f32x4 acc[BNT], acc2[BNT];
    KeepGen kg;
    {   // x1 = x + drop_in(o Wo^T + bo) ; y2 = LNb_out(x1)
        kg.begin(dc, g, SITE_SUB_IN, e128);
        const float* buf = ring.next();
        strip_zero<BSD>(acc);
        strip_product<BSD, BSPREAD>(acc, A, buf, ring, [&](int ct, int j) { ring.fetch(a.w1[g], ct, j); kg.hook(ct, j); });
        add_bias<BSD>(acc, bias);
        to_regs<BSD>(A, acc);
        apply_keep(A, kg.finish(), dc.scale);
#pragma unroll
        for (int ct = 0; ct < BNT; ++ct) X1.v[ct] += A.v[ct];
        strip_lnb(Y2, X1, la, lb);
    }
    strip_zero<BSD>(acc2);
    unsigned kb1 = ~0u, kb2 = ~0u;
    // (Tried, round 4: the eight products software-pipelined -- W1_0, W1_1, W2_0, W1_2, ... -- with chunk c's GELU + dropout inside the matrix
    // loop of the next product that does not need them, a column tile or an element per slot, forward and backward: the loops already carry
    // the dropout counters' rounds, the DMA pieces and the deferred stores, and the extra vector work stretches them by more than it saves
    // between them -- 0.6548 -> 0.6782 / 0.6657 ms per step; left as the plain chain.)
#pragma unroll
    for (int c = 0; c < BSC; ++c) {
        {   // pre_c = y2 W1_c^T + b1_c ; h_c = drop_ffn(gelu(pre_c))
            kg.begin(dc, g, SITE_FFN1, e512 + c * BSD);
            const float* buf = ring.next();
            bias.load(a.b1[g] + c * BSD);
            strip_zero<BSD>(acc);
            strip_product<BSD, BSPREAD>(acc, Y2, buf, ring, [&](int ct, int j) {
                bfetch_cols(ring, a.w2[g], c, ct, j);
                if (c == 0) { spread_at(gx1, row, X1, ct, j, 1); spread_at(gy2, row, Y2, ct, j, 3); }
                kg.hook(ct, j);
            });
            const unsigned kb = kg.finish();
            add_bias<BSD>(acc, bias);
            to_regs<BSD>(P, acc);
#pragma unroll
            for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
#ifdef AMID_BS_ABLATE_GELU
                for (int e = 0; e < 4; ++e) Hc.v[ct][e] = P.v[ct][e] * 0.5f;
#else
                for (int e = 0; e < 4; ++e) Hc.v[ct][e] = gelu_f(P.v[ct][e]);
#endif
            apply_keep(Hc, kb, dc.scale);
        }
        {   // z += h_c W2_c^T ; pre_c's and h_c's global copies leave under these MFMAs (chunks 0 / 1: the decisions of the block's last two sites)
            if (c == 0) kg.begin(dc, g, SITE_SUB_OUT, e128);
            if (c == 1) kg.begin(dc, g, SITE_BLOCK, e128);
            const float* buf = ring.next();
            const float* nxt = c + 1 < BSC ? btile_rows<R>(a.w1[g], c + 1) : nx.w[0][g];
            strip_product<BSD, BSPREAD>(acc2, Hc, buf, ring, [&](int ct, int j) {
                if (NEXT || c + 1 < BSC) ring.fetch(nxt, ct, j);
                wide_spread(gpre, offw, c, P, ct, j, 1);
                wide_spread(gh, offw, c, Hc, ct, j, 3);
                if (c < 2) kg.hook(ct, j);
            });
            if (c == 0) kb1 = kg.finish();
            if (c == 1) kb2 = kg.finish();
        }
    }
    {   // x2 = drop_block(x1 + drop_out(z + b2))
        bias.load(a.b2[g]);
        if constexpr (NEXT) { la.load(nx.la[g]); lb.load(nx.lb[g]); }
        add_bias<BSD>(acc2, bias);
        to_regs<BSD>(A, acc2);
        apply_keep(A, kb1, dc.scale);
#pragma unroll
        for (int ct = 0; ct < BNT; ++ct) A.v[ct] += X1.v[ct];
        apply_keep(A, kb2, dc.scale);
    }
    if constexpr (NEXT) {
        bqkv_fwd_chain<true>(nx, sg, ring, row, g, A, la, lb);
    } else {
        strip_store<BSD>(GBuf(a.x2, sg.act_bytes), row, A);
    }
}

// ================================================================================================================ backward
struct BStripFfnBwdArgs {
    const float* dx2; const float* pre; const float* x1; const float* la[2];           // LNb_out gamma
    const float* w2T[2]; const float* w1T[2]; const float* woT[2];                   // [512,128] ; [128,512] ; [128,128]
    float* dz; float* dpre; float* dx1; float* dt; float* d_o; float* ln_part;      // [2M,128], [2M,512], [2M,128] x 3, [2 tpg][2][128]
    const StepState* st; int train; unsigned spec; float scale; int layer;
};
struct BStripQkvBwdArgs {
    const float* dq; const float* dk; const float* dv; const float* dx1; const float* x; const float* la[2];     // LNb_in gamma
    const float* wT[3][2];
    float* dx; float* ln_part;
    int zero_blocks;        // > 0 (live list given, dx written): that many extra workgroups behind the tiles' zero the rows of dx the live walk
                            // never writes -- the DEAD sequences' (their gradient is exactly zero; the segment reduce reads every row)
};

// dead sequence j of the live list: the other domain's sequence of the sample live[j]
__device__ __forceinline__ void zero_dead_rows(float* __restrict__ dx, const StripGeom& sg, int blk, int nblk) {
    const int n0 = sg.live[sg.B];
    const int q = sg.T * (BSD / 4);
    for (int j = blk; j < sg.B; j += nblk) {
        const int g_dead = j < n0 ? 1 : 0;
        float* base = dx + ((long long)g_dead * sg.M + (long long)sg.live[j] * sg.T) * BSD;
        for (int i = threadIdx.x; i < q; i += STRIP_THREADS) st4_global(base + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

// d x2 (DX2, in registers) -> dz, dpre, dx1, dt, d_o of this block; the ring's current fetch must be chunk 0 of w2T.
// TAIL: a slab (`tail`) is fetched under the last MFMA loop (a fused successor's first weight).
// kb_block / kb_out: the keep bits of the block's last two dropout sites (bffn_drop(a) + keep_bits, or drawn by a fused predecessor
// under its own matrix loops)
__device__ __forceinline__ BDrop bffn_drop(const BStripFfnBwdArgs& a) {
    BDrop dc = {a.train, a.spec, a.scale, 0ull, 0u, a.layer};
    if (a.train) { dc.seed = a.st->seed; dc.step = (unsigned)a.st->step; }
    return dc;
}
template <class R>
__device__ __forceinline__ void bffn_bwd_chain(const BStripFfnBwdArgs& a, const StripGeom& sg, R& ring, const StripRow& row, int g,
                                               StripRegs<BSD>& DX2, float* __restrict__ scratch, unsigned kb_block, unsigned kb_out) {
    const BDrop dc = bffn_drop(a);
    const unsigned long long e128 = (unsigned long long)row.local * BSD, e512 = (unsigned long long)row.local * BSF;
    const unsigned wide_bytes = sg.act_bytes * 4u;
    const GBuf gpre(a.pre, wide_bytes), gdpre(a.dpre, wide_bytes), gdz(a.dz, sg.act_bytes), gdx1(a.dx1, sg.act_bytes), gdt(a.dt, sg.act_bytes);
    const unsigned offw = wide_off(row);
    StripRegs<BSD> DZ, DP, PRE, X1;
    ColVec<BSD> gam;
    // dr = dx2 * drop_block (kept in DX2: the residual path into x1) ; dz = dr * drop_out
    apply_keep(DX2, kb_block, dc.scale);
    DZ = DX2;
    apply_keep(DZ, kb_out, dc.scale);
    KeepGen kg;
    unsigned kb_in = ~0u;
    f32x4 acc[BNT], accy[BNT];
    strip_zero<BSD>(accy);
#pragma unroll
    for (int c = 0; c < BSC; ++c) {
        {   // dh_c = dz W2_c ; dpre_c = dh_c * drop_ffn * gelu'(pre_c)
            kg.begin(dc, g, SITE_FFN1, e512 + c * BSD);
            const float* buf = ring.next();
            wide_load(PRE, gpre, offw, c);
            strip_zero<BSD>(acc);
            strip_product<BSD, BSPREAD>(acc, DZ, buf, ring, [&](int ct, int j) {
                bfetch_cols(ring, a.w1T[g], c, ct, j);
                if (c == 0) spread_at(gdz, row, DZ, ct, j, 1);
                kg.hook(ct, j);
            });
            const unsigned kb = kg.finish();
            to_regs<BSD>(DP, acc);
            apply_keep(DP, kb, dc.scale);
#pragma unroll
            for (int ct = 0; ct < BNT; ++ct)
#pragma unroll
#ifdef AMID_BS_ABLATE_GELU
                for (int e = 0; e < 4; ++e) DP.v[ct][e] *= PRE.v[ct][e];
#else
                for (int e = 0; e < 4; ++e) DP.v[ct][e] *= gelu_df(PRE.v[ct][e]);
#endif
        }
        {   // dy2 += dpre_c W1_c
            const float* buf = ring.next();
            const float* nxt = c + 1 < BSC ? btile_rows<R>(a.w2T[g], c + 1) : a.woT[g];
            if (c + 1 == BSC) { strip_load<BSD>(X1, GBuf(a.x1, sg.act_bytes), row); gam.load(a.la[g]); }
            if (c == 0) kg.begin(dc, g, SITE_SUB_IN, e128);
            strip_product<BSD, BSPREAD>(accy, DP, buf, ring, [&](int ct, int j) {
                ring.fetch(nxt, ct, j);
                wide_spread(gdpre, offw, c, DP, ct, j, 1);
                if (c == 0) kg.hook(ct, j);
            });
            if (c == 0) kb_in = kg.finish();
        }
    }
    StripRegs<BSD> dgam, dbet;
    {   // dx1 = LNb_out'(dy2 ; x1) + dr ; dt = dx1 * drop_in ; d_o = dt Wo
        const unsigned kb = kb_in;
        to_regs<BSD>(DZ, accy);
        strip_lnb_bwd(DP, DZ, X1, gam, dgam, dbet);
#pragma unroll
        for (int ct = 0; ct < BNT; ++ct) DP.v[ct] += DX2.v[ct];
        DZ = DP;
        apply_keep(DZ, kb, dc.scale);
        const float* buf = ring.next();
        strip_zero<BSD>(acc);
        strip_product<BSD, BSPREAD>(acc, DZ, buf, ring, [&](int ct, int j) { spread_at(gdx1, row, DP, ct, j, 1); spread_at(gdt, row, DZ, ct, j, 3); });
        to_regs<BSD>(PRE, acc);
        strip_store<BSD>(GBuf(a.d_o, sg.act_bytes), row, PRE);
    }
    ln_partials_wave<BSD>(scratch, dgam, dbet);
}

template <int MODE>
__global__ __launch_bounds__(STRIP_THREADS) void bert_strip_ffn_bwd_kernel(const BStripFfnBwdArgs a, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    typename BRingSel<MODE>::type ring(smem);
    ring.first(a.w2T[strip_domain(blockIdx.x)]);
    const StripTile t = strip_tile(sg, blockIdx.x);
    if (!t.live) { zero_slot<BSD>(a.ln_part, t.slot); w_ring_wait(); return; }
    const StripRow row = strip_row<BSD>(sg, t);
    StripRegs<BSD> DX2;
    strip_load<BSD>(DX2, GBuf(a.dx2, sg.act_bytes), row);
    const BDrop dc = bffn_drop(a);          // (the first slab is still landing: these two sites' counters cost no matrix time)
    const unsigned long long e128 = (unsigned long long)row.local * BSD;
    const unsigned kb_block = keep_bits(dc, t.g, SITE_BLOCK, e128), kb_out = keep_bits(dc, t.g, SITE_SUB_OUT, e128);
    bffn_bwd_chain(a, sg, ring, row, t.g, DX2, ln_scratch<BSD>(smem, 0), kb_block, kb_out);
    __syncthreads();
    ln_partials_out<BSD>(ln_scratch<BSD>(smem, 0), a.ln_part + (long long)t.slot * 2 * BSD);
}

// FFN = true: the block below's feed-forward / out-projection backward continues on d x in registers (d x is then never stored)
template <bool FFN, int MODE>
__global__ __launch_bounds__(STRIP_THREADS) void bert_strip_qkv_bwd_kernel(const BStripQkvBwdArgs a, const BStripFfnBwdArgs f, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (!FFN) {
        if ((int)blockIdx.x >= 2 * sg.tpg) { zero_dead_rows(a.dx, sg, blockIdx.x - 2 * sg.tpg, a.zero_blocks); return; }
    }
    typename BRingSel<MODE>::type ring(smem);
    ring.first(a.wT[0][strip_domain(blockIdx.x)]);
    const StripTile t = strip_tile(sg, blockIdx.x);
    if (!t.live) {
        zero_slot<BSD>(a.ln_part, t.slot);
        if constexpr (FFN) zero_slot<BSD>(f.ln_part, t.slot);
        w_ring_wait();
        return;
    }
    const StripRow row = strip_row<BSD>(sg, t);
    const int g = t.g;
    StripRegs<BSD> D0, D1, Xs, DX;
    ColVec<BSD> gam;
    strip_load<BSD>(D0, GBuf(a.dq, sg.act_bytes), row);
    f32x4 acc[BNT];
    strip_zero<BSD>(acc);
    // FFN: the keep bits of the block below's last two dropout sites are drawn under this chain's first two products
    KeepGen kg;
    unsigned kb_block = ~0u, kb_out = ~0u;
    BDrop fdc = {0, 0u, 1.f, 0ull, 0u, 0};
    if constexpr (FFN) fdc = bffn_drop(f);
    const unsigned long long e128 = (unsigned long long)row.local * BSD;
    {   // dq Wq          (every operand is requested one slab ahead of its use)
        if constexpr (FFN) kg.begin(fdc, g, SITE_BLOCK, e128);
        const float* buf = ring.next();
        strip_load<BSD>(D1, GBuf(a.dk, sg.act_bytes), row);
        strip_product<BSD, BSPREAD>(acc, D0, buf, ring, [&](int ct, int j) { ring.fetch(a.wT[1][g], ct, j); if constexpr (FFN) kg.hook(ct, j); });
        if constexpr (FFN) kb_block = kg.finish();
    }
    {   // + dk Wk
        if constexpr (FFN) kg.begin(fdc, g, SITE_SUB_OUT, e128);
        const float* buf = ring.next();
        strip_load<BSD>(D0, GBuf(a.dv, sg.act_bytes), row);
        strip_product<BSD, BSPREAD>(acc, D1, buf, ring, [&](int ct, int j) { ring.fetch(a.wT[2][g], ct, j); if constexpr (FFN) kg.hook(ct, j); });
        if constexpr (FFN) kb_out = kg.finish();
    }
    StripRegs<BSD> dgam, dbet;
    {   // + dv Wv ; dx = LNb_in'(. ; x) + dx1
        const float* buf = ring.next();
        strip_load<BSD>(Xs, GBuf(a.x, sg.act_bytes), row);
        strip_load<BSD>(D1, GBuf(a.dx1, sg.act_bytes), row);
        gam.load(a.la[g]);
        strip_product<BSD, BSPREAD>(acc, D0, buf, ring, [&](int ct, int j) { if constexpr (FFN) ring.fetch(f.w2T[g], ct, j); });
        to_regs<BSD>(D0, acc);
        strip_lnb_bwd(DX, D0, Xs, gam, dgam, dbet);
#pragma unroll
        for (int ct = 0; ct < BNT; ++ct) DX.v[ct] += D1.v[ct];
    }
    ln_partials_wave<BSD>(ln_scratch<BSD>(smem, 0), dgam, dbet);
    if constexpr (FFN) {
        bffn_bwd_chain(f, sg, ring, row, g, DX, ln_scratch<BSD>(smem, 1), kb_block, kb_out);
    } else {
        strip_store<BSD>(GBuf(a.dx, sg.act_bytes), row, DX);
    }
    __syncthreads();
    ln_partials_out<BSD>(ln_scratch<BSD>(smem, 0), a.ln_part + (long long)t.slot * 2 * BSD);
    if constexpr (FFN) ln_partials_out<BSD>(ln_scratch<BSD>(smem, 1), f.ln_part + (long long)t.slot * 2 * BSD);
}

}  // namespace amid

using namespace amid;
using namespace amid_strip_host;

// the 512-wide tensors go through buffer descriptors too: 4 x the bytes of a [2M, 128] tensor must stay below 2 GiB
static int bert_strip_geom(int B, int T, const int* live, StripGeom* sg) {
    if (int e = make_strip_geom(B, T, BSD, live, sg)) return e;
    if (4LL * sg->act_bytes > 0x7FFFFFF0LL) return AMID_ERR_UNSUPPORTED;
    return AMID_OK;
}

// 1 when the strip kernels cover the shape (hidden 128; activations [2 B T, 512] fp32 within 2 GiB); the caller falls back to the
// row-tile kernels of bert.hip otherwise
extern "C" int amid_bert_strip_supported(int B, int T, int D) {
    return (D == BSD && B > 0 && T > 0 && 2LL * B * T * BSF * 4 <= 0x7FFFFFF0LL) ? 1 : 0;
}

static void fill_bqkv(BStripQkvArgs& a, const float* x, const float* const* la, const float* const* lb, const float* const* w3,
                      const float* const* b3, float* y, float* q, float* k, float* v) {
    a.x = x; a.y = y; a.out[0] = q; a.out[1] = k; a.out[2] = v;
    for (int g = 0; g < 2; ++g) {
        a.la[g] = la[g]; a.lb[g] = lb[g];
        for (int j = 0; j < 3; ++j) { a.w[j][g] = w3[j * 2 + g]; a.b[j][g] = b3[j * 2 + g]; }
    }
}

static int bqkv_fwd(const float* x, const float* const* la, const float* const* lb, const float* const* w3, const float* const* b3, int B,
                    int T, const int* live, float* y, float* q, float* k, float* v, const BPrologue& pro, void* stream, int mode = 0) {
    AMID_CHECK_ARG(x && la && lb && w3 && b3 && y && q && k && v);
    BStripQkvArgs a;
    fill_bqkv(a, x, la, lb, w3, b3, y, q, k, v);
    StripGeom sg;
    if (int e = bert_strip_geom(B, T, live, &sg)) return e;
    static unsigned long long attr_done[2] = {0, 0};
    if (mode == 3) {
        if (int rc = lds_attr_once((const void*)bert_strip_qkv_fwd_kernel<3>, strip_lds_bytes<BSD>(), attr_done[1])) return rc;
        bert_strip_qkv_fwd_kernel<3><<<2 * sg.tpg + pro.blocks, STRIP_THREADS, strip_lds_bytes<BSD>(), (hipStream_t)stream>>>(a, sg, pro);
    } else {
        if (int rc = lds_attr_once((const void*)bert_strip_qkv_fwd_kernel<0>, strip_lds_bytes<BSD>(), attr_done[0])) return rc;
        bert_strip_qkv_fwd_kernel<0><<<2 * sg.tpg + pro.blocks, STRIP_THREADS, strip_lds_bytes<BSD>(), (hipStream_t)stream>>>(a, sg, pro);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}

// w3 / b3: host arrays of six device pointers ordered [q, k, v][domain] (as amid_bert_qkv_fwd_f32)
extern "C" int amid_bert_strip_qkv_fwd_f32(const float* x, const float* const* la, const float* const* lb, const float* const* w3,
                                           const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k, float* v,
                                           void* stream) {
    BPrologue pro = {};
    return bqkv_fwd(x, la, lb, w3, b3, B, T, live, y, q, k, v, pro, stream);
}

// ... with the step's prologue riding as extra workgroups: key_keep[i] = seq_d2[i] > 0 for i < n_keys (seq_d2 == NULL: none), and
// tr_dst[m][c][r] = tr_src[m][r][c] for n_tr <= 24 matrices of tr_rows[m] x tr_cols[m] floats (multiples of 64) -- what
// amid_key_keep_u8 and amid_transpose_rect_f32 do in launches of their own
static int bqkv_fwd_pro(const float* x, const float* const* la, const float* const* lb, const float* const* w3,
                        const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k,
                        float* v, const long long* seq_d2, int n_keys, unsigned char* key_keep,
                        const float* const* tr_src, float* const* tr_dst, const int* tr_rows, const int* tr_cols,
                        int n_tr, void* stream, int mode) {
    AMID_CHECK_ARG(n_tr >= 0 && n_tr <= BPRO_MAX && (n_tr == 0 || (tr_src && tr_dst && tr_rows && tr_cols)));
    AMID_CHECK_ARG(seq_d2 == nullptr || (key_keep != nullptr && n_keys > 0));
    BPrologue pro = {};
    pro.seq = seq_d2; pro.keep = key_keep; pro.n_keys = n_keys; pro.n = n_tr;
    for (int i = 0; i < n_tr; ++i) {
        AMID_CHECK_ARG(tr_src[i] && tr_dst[i] && tr_rows[i] > 0 && tr_cols[i] > 0 && tr_rows[i] % 64 == 0 && tr_cols[i] % 64 == 0);
        pro.src[i] = tr_src[i]; pro.dst[i] = tr_dst[i]; pro.rows[i] = tr_rows[i]; pro.cols[i] = tr_cols[i];
    }
    pro.blocks = (seq_d2 != nullptr || n_tr > 0) ? 96 : 0;
    return bqkv_fwd(x, la, lb, w3, b3, B, T, live, y, q, k, v, pro, stream, mode);
}
extern "C" int amid_bert_strip_qkv_fwd_pro_f32(const float* x, const float* const* la, const float* const* lb, const float* const* w3,
                                               const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k,
                                               float* v, const long long* seq_d2, int n_keys, unsigned char* key_keep,
                                               const float* const* tr_src, float* const* tr_dst, const int* tr_rows, const int* tr_cols,
                                               int n_tr, void* stream) {
    return bqkv_fwd_pro(x, la, lb, w3, b3, B, T, live, y, q, k, v, seq_d2, n_keys, key_keep, tr_src, tr_dst, tr_rows, tr_cols, n_tr, stream, 0);
}
// ... on bf16 pieces: w3 = the tiles' three-plane images (amid_bert_weight_images_f32), everything else as above
extern "C" int amid_bert_strip_qkv_fwd_pro_p3_f32(const float* x, const float* const* la, const float* const* lb, const float* const* w3_img,
                                                  const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k,
                                                  float* v, const long long* seq_d2, int n_keys, unsigned char* key_keep,
                                                  const float* const* tr_src, float* const* tr_dst, const int* tr_rows, const int* tr_cols,
                                                  int n_tr, void* stream) {
    return bqkv_fwd_pro(x, la, lb, w3_img, b3, B, T, live, y, q, k, v, seq_d2, n_keys, key_keep, tr_src, tr_dst, tr_rows, tr_cols, n_tr, stream, 3);
}

// out-projection + feed-forward of a block; nla != NULL: the next block's LayerNorm + q / k / v on x2 in the same launch (x2 is
// then also the next block's saved input)
static int boproj_ffn_fwd(const float* o, const float* x, const float* const* wo, const float* const* bo,
                          const float* const* la, const float* const* lb, const float* const* w1,
                          const float* const* b1, const float* const* w2, const float* const* b2, int B, int T,
                          const int* live, int layer, const void* step_state, int train, float p_drop, float* x1,
                          float* y2, float* pre, float* h, float* x2, const float* const* nla,
                          const float* const* nlb, const float* const* nw3, const float* const* nb3, float* ny,
                          float* nq, float* nk, float* nv, void* stream, int mode) {
    AMID_CHECK_ARG(o && x && wo && bo && la && lb && w1 && b1 && w2 && b2 && x1 && y2 && pre && h && x2 && (!train || step_state));
    const bool next = nla != nullptr;
    AMID_CHECK_ARG(!next || (nlb && nw3 && nb3 && ny && nq && nk && nv));
    BStripOffArgs a;
    a.o = o; a.x = x; a.x1 = x1; a.y2 = y2; a.pre = pre; a.h = h; a.x2 = x2;
    a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.spec = drop_spec(p_drop);
    if (a.train && spec_bits(a.spec) != 16) return AMID_ERR_UNSUPPORTED;      // KeepGen: the 16-bit decisions of p = 0.1 (the reference's rate, model_seq.py:267)
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) {
        a.wo[g] = wo[g]; a.bo[g] = bo[g]; a.la[g] = la[g]; a.lb[g] = lb[g];
        a.w1[g] = w1[g]; a.b1[g] = b1[g]; a.w2[g] = w2[g]; a.b2[g] = b2[g];
    }
    BStripQkvArgs nx = {};
    if (next) fill_bqkv(nx, x2, nla, nlb, nw3, nb3, ny, nq, nk, nv);
    StripGeom sg;
    if (int e = bert_strip_geom(B, T, live, &sg)) return e;
    if (mode == 3)
        return next ? launch_strip<bert_strip_oproj_ffn_fwd_kernel<true, 3>, BSD>(sg, stream, a, nx)
                    : launch_strip<bert_strip_oproj_ffn_fwd_kernel<false, 3>, BSD>(sg, stream, a, nx);
    return next ? launch_strip<bert_strip_oproj_ffn_fwd_kernel<true, 0>, BSD>(sg, stream, a, nx)
                : launch_strip<bert_strip_oproj_ffn_fwd_kernel<false, 0>, BSD>(sg, stream, a, nx);
}
extern "C" int amid_bert_strip_oproj_ffn_fwd_f32(const float* o, const float* x, const float* const* wo, const float* const* bo,
                                                 const float* const* la, const float* const* lb, const float* const* w1,
                                                 const float* const* b1, const float* const* w2, const float* const* b2, int B, int T,
                                                 const int* live, int layer, const void* step_state, int train, float p_drop, float* x1,
                                                 float* y2, float* pre, float* h, float* x2, const float* const* nla,
                                                 const float* const* nlb, const float* const* nw3, const float* const* nb3, float* ny,
                                                 float* nq, float* nk, float* nv, void* stream) {
    return boproj_ffn_fwd(o, x, wo, bo, la, lb, w1, b1, w2, b2, B, T, live, layer, step_state, train, p_drop, x1, y2, pre, h, x2, nla, nlb, nw3, nb3,
                          ny, nq, nk, nv, stream, 0);
}
// ... on bf16 pieces: wo / nw3 = tile images, w1 / w2 = the first of their four tiles' images (one behind the other)
extern "C" int amid_bert_strip_oproj_ffn_fwd_p3_f32(const float* o, const float* x, const float* const* wo_img, const float* const* bo,
                                                    const float* const* la, const float* const* lb, const float* const* w1_img,
                                                    const float* const* b1, const float* const* w2_img, const float* const* b2, int B, int T,
                                                    const int* live, int layer, const void* step_state, int train, float p_drop, float* x1,
                                                    float* y2, float* pre, float* h, float* x2, const float* const* nla,
                                                    const float* const* nlb, const float* const* nw3_img, const float* const* nb3, float* ny,
                                                    float* nq, float* nk, float* nv, void* stream) {
    return boproj_ffn_fwd(o, x, wo_img, bo, la, lb, w1_img, b1, w2_img, b2, B, T, live, layer, step_state, train, p_drop, x1, y2, pre, h, x2, nla, nlb,
                          nw3_img, nb3, ny, nq, nk, nv, stream, 3);
}

static int fill_bffn_bwd(BStripFfnBwdArgs& a, const float* dx2, const float* pre, const float* x1, const float* const* la,
                         const float* const* w2T, const float* const* w1T, const float* const* woT, int layer, const void* step_state,
                         int train, float p_drop, float* dz, float* dpre, float* dx1, float* dt, float* d_o, float* ln_part) {
    AMID_CHECK_ARG(pre && x1 && la && w2T && w1T && woT && dz && dpre && dx1 && dt && d_o && ln_part && (!train || step_state));
    a.dx2 = dx2; a.pre = pre; a.x1 = x1; a.dz = dz; a.dpre = dpre; a.dx1 = dx1; a.dt = dt; a.d_o = d_o; a.ln_part = ln_part;
    a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.spec = drop_spec(p_drop);
    if (a.train && spec_bits(a.spec) != 16) return AMID_ERR_UNSUPPORTED;
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) { a.la[g] = la[g]; a.w2T[g] = w2T[g]; a.w1T[g] = w1T[g]; a.woT[g] = woT[g]; }
    return AMID_OK;
}

// ln_part: [2 * ceil(B T / amid_sas_strip_tile_rows())][2][128]; domain g's partial sums are slots [g * tpg, (g + 1) * tpg)
static int bffn_bwd(const float* dx2, const float* pre, const float* x1, const float* const* la,
                    const float* const* w2T, const float* const* w1T, const float* const* woT, int B, int T,
                    const int* live, int layer, const void* step_state, int train, float p_drop, float* dz,
                    float* dpre, float* dx1, float* dt, float* d_o, float* ln_part, void* stream, int mode) {
    AMID_CHECK_ARG(dx2);
    BStripFfnBwdArgs a;
    if (int e = fill_bffn_bwd(a, dx2, pre, x1, la, w2T, w1T, woT, layer, step_state, train, p_drop, dz, dpre, dx1, dt, d_o, ln_part)) return e;
    StripGeom sg;
    if (int e = bert_strip_geom(B, T, live, &sg)) return e;
    return mode == 3 ? launch_strip<bert_strip_ffn_bwd_kernel<3>, BSD>(sg, stream, a) : launch_strip<bert_strip_ffn_bwd_kernel<0>, BSD>(sg, stream, a);
}
extern "C" int amid_bert_strip_ffn_bwd_f32(const float* dx2, const float* pre, const float* x1, const float* const* la,
                                           const float* const* w2T, const float* const* w1T, const float* const* woT, int B, int T,
                                           const int* live, int layer, const void* step_state, int train, float p_drop, float* dz,
                                           float* dpre, float* dx1, float* dt, float* d_o, float* ln_part, void* stream) {
    return bffn_bwd(dx2, pre, x1, la, w2T, w1T, woT, B, T, live, layer, step_state, train, p_drop, dz, dpre, dx1, dt, d_o, ln_part, stream, 0);
}
// ... on bf16 pieces: w2T / w1T = the first of the four TRANSPOSED tiles' images, woT = the transposed tile's image
extern "C" int amid_bert_strip_ffn_bwd_p3_f32(const float* dx2, const float* pre, const float* x1, const float* const* la,
                                              const float* const* w2T_img, const float* const* w1T_img, const float* const* woT_img, int B, int T,
                                              const int* live, int layer, const void* step_state, int train, float p_drop, float* dz,
                                              float* dpre, float* dx1, float* dt, float* d_o, float* ln_part, void* stream) {
    return bffn_bwd(dx2, pre, x1, la, w2T_img, w1T_img, woT_img, B, T, live, layer, step_state, train, p_drop, dz, dpre, dx1, dt, d_o, ln_part, stream, 3);
}

// wT3: six device pointers ordered [q, k, v][domain] (transposed weights).  fpre != NULL: the block below's feed-forward /
// out-projection backward (f* arguments, as amid_bert_strip_ffn_bwd_f32 without dx2) runs on d x in the same launch; dx is then not written.
// zero_dead (with a live list and dx): the rows of dx that belong to the sequences NOT on the list are zero-filled by extra workgroups
static int bqkv_bwd(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x,
                    const float* const* la, const float* const* wT3, int B, int T, const int* live, float* dx,
                    int zero_dead, float* ln_part, const float* fpre, const float* fx1, const float* const* fla,
                    const float* const* fw2T, const float* const* fw1T, const float* const* fwoT, int flayer,
                    const void* step_state, int train, float p_drop, float* fdz, float* fdpre, float* fdx1,
                    float* fdt, float* fd_o, float* fln_part, void* stream, int mode) {
    AMID_CHECK_ARG(dq && dk && dv && dx1 && x && la && wT3 && ln_part);
    const bool ffn = fpre != nullptr;
    AMID_CHECK_ARG(ffn || dx);
    BStripQkvBwdArgs a;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dx1 = dx1; a.x = x; a.dx = dx; a.ln_part = ln_part;
    for (int g = 0; g < 2; ++g) {
        a.la[g] = la[g];
        for (int j = 0; j < 3; ++j) a.wT[j][g] = wT3[j * 2 + g];
    }
    BStripFfnBwdArgs f = {};
    if (ffn) if (int e = fill_bffn_bwd(f, nullptr, fpre, fx1, fla, fw2T, fw1T, fwoT, flayer, step_state, train, p_drop, fdz, fdpre, fdx1, fdt, fd_o, fln_part)) return e;
    StripGeom sg;
    if (int e = bert_strip_geom(B, T, live, &sg)) return e;
    a.zero_blocks = (zero_dead && live != nullptr && !ffn) ? (B < 256 ? B : 256) : 0;
    if (ffn) return mode == 3 ? launch_strip<bert_strip_qkv_bwd_kernel<true, 3>, BSD>(sg, stream, a, f)
                              : launch_strip<bert_strip_qkv_bwd_kernel<true, 0>, BSD>(sg, stream, a, f);
    static unsigned long long attr_done[2] = {0, 0};
    if (mode == 3) {
        if (int rc = lds_attr_once((const void*)bert_strip_qkv_bwd_kernel<false, 3>, strip_lds_bytes<BSD>(), attr_done[1])) return rc;
        bert_strip_qkv_bwd_kernel<false, 3><<<2 * sg.tpg + a.zero_blocks, STRIP_THREADS, strip_lds_bytes<BSD>(), (hipStream_t)stream>>>(a, f, sg);
    } else {
        if (int rc = lds_attr_once((const void*)bert_strip_qkv_bwd_kernel<false, 0>, strip_lds_bytes<BSD>(), attr_done[0])) return rc;
        bert_strip_qkv_bwd_kernel<false, 0><<<2 * sg.tpg + a.zero_blocks, STRIP_THREADS, strip_lds_bytes<BSD>(), (hipStream_t)stream>>>(a, f, sg);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}
extern "C" int amid_bert_strip_qkv_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x,
                                           const float* const* la, const float* const* wT3, int B, int T, const int* live, float* dx,
                                           int zero_dead, float* ln_part, const float* fpre, const float* fx1, const float* const* fla,
                                           const float* const* fw2T, const float* const* fw1T, const float* const* fwoT, int flayer,
                                           const void* step_state, int train, float p_drop, float* fdz, float* fdpre, float* fdx1,
                                           float* fdt, float* fd_o, float* fln_part, void* stream) {
    return bqkv_bwd(dq, dk, dv, dx1, x, la, wT3, B, T, live, dx, zero_dead, ln_part, fpre, fx1, fla, fw2T, fw1T, fwoT, flayer, step_state, train, p_drop,
                    fdz, fdpre, fdx1, fdt, fd_o, fln_part, stream, 0);
}
// ... on bf16 pieces: every weight argument = transposed tile images (as amid_bert_strip_ffn_bwd_p3_f32)
extern "C" int amid_bert_strip_qkv_bwd_p3_f32(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x,
                                              const float* const* la, const float* const* wT3_img, int B, int T, const int* live, float* dx,
                                              int zero_dead, float* ln_part, const float* fpre, const float* fx1, const float* const* fla,
                                              const float* const* fw2T_img, const float* const* fw1T_img, const float* const* fwoT_img, int flayer,
                                              const void* step_state, int train, float p_drop, float* fdz, float* fdpre, float* fdx1,
                                              float* fdt, float* fd_o, float* fln_part, void* stream) {
    return bqkv_bwd(dq, dk, dv, dx1, x, la, wT3_img, B, T, live, dx, zero_dead, ln_part, fpre, fx1, fla, fw2T_img, fw1T_img, fwoT_img, flayer, step_state,
                    train, p_drop, fdz, fdpre, fdx1, fdt, fd_o, fln_part, stream, 3);
}

// Three-plane bf16 fragment images (weights_image.h: hi + mid + lo = the fp32 element exactly) of n <= 96 weight TILES of 128 x 128: tile i
// is src[i][r * ld[i] + c] (tr[i] = 0) or its transpose src[i][c * ld[i] + r] (tr[i] != 0), r, c < 128; dst16: [n][3][128][128] bf16.
// One launch per step in front of the first strip launch (what SASRec's gather carries as riders).
constexpr int BIMG_MAX = 96;
struct BImgArgs { const float* src[BIMG_MAX]; unsigned short ld[BIMG_MAX]; unsigned char tr[BIMG_MAX]; int n, per; unsigned short* dst; };
__global__ __launch_bounds__(256) void bert_weight_images_kernel(const BImgArgs a) {
    const int wi = blockIdx.x / a.per, b = blockIdx.x - wi * a.per;
    weights_image_block(a.src[wi], a.dst + (size_t)wi * 3 * BSD * BSD, BSD, a.tr[wi], 3, b, a.per, a.ld[wi]);
}
extern "C" int amid_bert_weight_images_f32(const float* const* src, const int* ld, const int* tr, int n, void* dst16, void* stream) {
    AMID_CHECK_ARG(src && ld && tr && dst16 && n > 0 && n <= BIMG_MAX);
    BImgArgs a;
    for (int i = 0; i < n; ++i) {
        AMID_CHECK_ARG(src[i] && ld[i] >= BSD && ld[i] < 65536);
        a.src[i] = src[i]; a.ld[i] = (unsigned short)ld[i]; a.tr[i] = tr[i] ? 1 : 0;
    }
    a.n = n; a.per = 4; a.dst = (unsigned short*)dst16;
    bert_weight_images_kernel<<<n * a.per, 256, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
