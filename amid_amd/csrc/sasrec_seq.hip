// The whole SASRec encoder forward of a sequence in ONE launch: for every layer LayerNorm1 + q / k / v projections, the causal
// multi-head attention core, out-projection + residual + LayerNorm2 + point-wise feed-forward -- the register-resident strip chains
// of sasrec_strip.hip with the attention core of attention_mfma.hip between them, nothing but the saved activations leaving the chip.
// Reference: Log2feats.forward model_seq.py:371-383 with nn.MultiheadAttention as called at :374 (softmax((q sqrt(1/hd)) k^T +
// causal(-inf)) -> dropout(p) -> . v) and PointWiseFeedForward :322-326.
//
// Why it fits: in the strip layout a wave's accumulators are the next product's operand, and that holds for the attention core as well.
//   * the tile is sequence-aligned: a workgroup (4 waves) owns 4 / WPS sequences, WPS = 1, 2 or 4 waves (16-row strips) per sequence
//     (T <= 16 / 32 / 64); wave (sq, si) holds query rows 16 si .. 16 si + 15 of its sequence.
//   * S = Qs K^T for the wave's 16 queries and key tile kt <= si: second operand = the q accumulators of head h as they stand
//     (C layout: lane (m, g) register r = q[m][16 h + 4 g + r]), first operand = 16 K rows read from an LDS image with one
//     ds_read_b128 per 4 MFMAs -> lane (m, g) holds S[m][16 kt + 4 g + r]: the softmax is in-lane + two shuffles.
//   * O = P~ V: the P~ registers are the second operand (lane group g supplies key 4 g + r), V^T comes from the LDS image as dwords;
//     the result lands as lane (m, g) register r = O[m][16 h + 4 g + r] -- column tile h of the C layout, i.e. the out-projection's
//     operand.  No transposes, no staging of Q or O.
//   * only K and V cross waves.  LDS holds the two-slab weight ring (128 KB at D = 128), so the images carry TWO heads at a time
//     (2 x 64 rows x 32 columns x 4 B = 16 KB): four rounds per layer, each: barrier, every wave writes its rows of the two heads'
//     K / V columns (from its accumulators), barrier, attention of the two heads.  Image rows are 128 B; the 16-byte chunk c of row R
//     sits at chunk position c ^ ((R >> 1) & 7): conflict-free for the ds_read_b128 K fragments and the ds_read_b32 V^T fragments.
// Saved for backward exactly what the separate kernels save: x, qn, q (unscaled), k, v, o, stats (row max, 1 / row sum), r, y, h.
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "attention_mfma.h"
#include "seq_fwd.h"
#include "weights_image.h"
#include "head_parts.h"

namespace amid {

#ifdef AMID_STRIP_STAMPS
static __device__ unsigned long long amid_seq_sched_buf[1024 * 4];     // per workgroup: start, end (100 MHz real-time counter), HW_ID
#define SEQ_SCHED(slot) do { if (threadIdx.x == 0 && blockIdx.x < 1024) amid_seq_sched_buf[blockIdx.x * 4 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SEQ_SCHED_ID() do { if (threadIdx.x == 0 && blockIdx.x < 1024) amid_seq_sched_buf[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } while (0)
static __device__ unsigned long long amid_seq_fine_buf[8 * 64];        // workgroup 0, last layer: per slab after the ring wait / the MFMA loop / the epilogue
#define SEQ_FINE(i) do { if (blockIdx.x == 0 && lane_id() == 0 && l == 1) amid_seq_fine_buf[wave_id() * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEQ_FINE(i) do { } while (0)
#define SEQ_SCHED(slot) do { } while (0)
#define SEQ_SCHED_ID() do { } while (0)
#endif

template <int D>
__device__ __forceinline__ void add_bias_s(f32x4 (&acc)[D / 16], const ColVec<D>& b) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) acc[ct] += b.v[ct];
}
template <int D>
__device__ __forceinline__ void to_regs_s(StripRegs<D>& dst, const f32x4 (&acc)[D / 16]) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) dst.v[ct] = acc[ct];
}

template <int D> struct SeqRing {       // the two-slab weight ring of sasrec_strip.hip
    float* buf; int s; WDma<D> dma;
    __device__ __forceinline__ explicit SeqRing(float* lds) : buf(lds), s(0) {}
    __device__ __forceinline__ void first(const float* __restrict__ W0) { dma.all(buf, W0); }
    __device__ __forceinline__ const float* next() {
        w_ring_wait();
        __syncthreads();
        const float* cur = buf + (s & 1) * D * D;
        ++s;
        return cur;
    }
    __device__ __forceinline__ void fetch(const float* __restrict__ W, int ct, int j) const {
        constexpr int SLOTS = 8 * (D / 16), EVERY = (SLOTS / 2) / WDma<D>::PER_WAVE;
        const int slot = ct * 8 + j;
        if (slot % EVERY == 0 && slot / EVERY < WDma<D>::PER_WAVE) dma.piece(buf + (s & 1) * D * D, W, slot / EVERY);
    }
};

template <int D>
__device__ __forceinline__ void spread(const GBuf& g, const StripRow& row, const StripRegs<D>& x, int ct, int j, int phase) {
    constexpr int NT = D / 16;
    if (ct < NT / 2 && (j & 3) == phase) strip_store_ct<D>(g, row, x, 2 * ct + (j >> 2));
}

constexpr int IMG_COLS = 32;            // two heads of 16
constexpr int IMG_ROWS = 64;
// K image [64 key rows][32]: 128-byte rows, chunk c of row R at chunk position c ^ ((R >> 1) & 7).
// V image TRANSPOSED [32 = 2 heads x 16 dims][64 keys]: 256-byte rows, the chunk of keys 4 c .. 4 c + 3 of row d at position
// c ^ (d & 15): the V^T fragment of lane (d, g) -- keys 16 kt + 4 g .. + 3 -- is ONE ds_read_b128 (a row-major image costs four
// ds_read_b32 per fragment: 32 reads per round instead of 8), conflict-free like the weight image; the writes (4 dwords per lane and
// column tile: keys are lanes, dims are registers) are conflict-free too.

// attention of the wave's 16 query rows (strip si of its sequence) over all heads; Q, K, V in the C layout
template <int D, int WPS>
__device__ __forceinline__ void seq_attention_fwd(StripRegs<D>& O, f32x4 (&stat)[D / 32], const StripRegs<D>& Q, const StripRegs<D>& K,
                                                  const StripRegs<D>& V, float* __restrict__ kimg, float* __restrict__ vimg, int si, int t,
                                                  int T, unsigned long long rowbase_bh, float scale, int train, unsigned long long seed,
                                                  unsigned site, unsigned step, unsigned spec, float dscale) {
    constexpr int H = D / 16;
    const int lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int R = wave_id() * 16 + m;                      // this lane's row of the K image / key column of the V^T image
    const int img0 = (wave_id() - si) * 16;                // first image row of the wave's sequence
    const int qrow = min(t, T - 1);
    // causal mask as the accumulators' initial value (-inf + anything finite stays -inf): tiles above the diagonal and the diagonal
    // tile's upper triangle; the same for every head
    f32x4 minit[WPS];
#pragma unroll
    for (int kt = 0; kt < WPS; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) minit[kt][r] = (kt * 16 + 4 * gq + r > t) ? -INFINITY : 0.f;
    // dropout keep words (64 keys each) of this lane's query row: lane group g draws the words of heads 2 g and 2 g + 1 -- one
    // Philox call each instead of eight per lane -- and round rd fetches its two from group rd
    unsigned kwl[2], kwh[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        unsigned long long kw = ~0ull;
        if (train) kw = row_keep_word(seed, site, step, (rowbase_bh + 2 * gq + e) * T + qrow, T, spec);
        kwl[e] = (unsigned)kw; kwh[e] = (unsigned)(kw >> 32);
    }
    const float qscale = scale * LOG2E;                    // softmax in base 2: p = 2^(s' - max s'), s' = s log2(e)
#pragma unroll
    for (int rd = 0; rd < H / 2; ++rd) {
        if (rd == 1) STRIP_STAMP(22);
        lds_barrier();                                     // everybody is done reading the previous round's images
        if (rd == 1) STRIP_STAMP(23);
        {
            const int sw = (R >> 1) & 7;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 kk = K.v[2 * rd + hh], vv = V.v[2 * rd + hh];
                st4(kimg + R * IMG_COLS + ((hh * 4 + gq) ^ sw) * 4, make_float4(kk[0], kk[1], kk[2], kk[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = hh * 16 + 4 * gq + r;    // this register's dim: row of the transposed image; this lane's key R: column
                    vimg[d * IMG_ROWS + (((R >> 2) ^ (d & 15)) * 4) + (R & 3)] = vv[r];
                }
            }
        }
        if (rd == 1) STRIP_STAMP(24);
        lds_barrier();                                     // the two heads' K / V of every sequence are in place
        if (rd == 1) STRIP_STAMP(25);
        // Both heads of the round and ALL key tiles of the sequence in one branch-free block (tiles above the diagonal are masked
        // like the diagonal's upper triangle: with one wave per SIMD the scheduler needs every independent chain it can get, and the
        // waves below the last strip would wait for it at the next barrier anyway).
        float4 kf[2][WPS], vf[2][WPS];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int d = hh * 16 + m;                     // V^T fragment: lane (d, g) reads keys 16 kt + 4 g .. + 3 of row d
#pragma unroll
            for (int kt = 0; kt < WPS; ++kt) {
                const int Rk = img0 + kt * 16 + m;
                kf[hh][kt] = ld4(kimg + Rk * IMG_COLS + (((hh * 4 + gq) ^ ((Rk >> 1) & 7)) * 4));
                vf[hh][kt] = ld4(vimg + d * IMG_ROWS + ((((img0 >> 2) + kt * 4 + gq) ^ (d & 15)) * 4));
            }
        }
        unsigned kl[2], kh[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) { kl[e] = bcast_group(kwl[e], rd); kh[e] = bcast_group(kwh[e], rd); }
        if (rd == 1) STRIP_STAMP(26);
        f32x4 s[2][WPS];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
#pragma unroll
            for (int kt = 0; kt < WPS; ++kt) s[hh][kt] = minit[kt];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const float qs = Q.v[2 * rd + hh][r] * qscale;
#pragma unroll
                for (int kt = 0; kt < WPS; ++kt) {
                    const float4 k4 = kf[hh][kt];
                    s[hh][kt] = mfma4(r == 0 ? k4.x : r == 1 ? k4.y : r == 2 ? k4.z : k4.w, qs, s[hh][kt]);
                }
            }
        if (rd == 1) STRIP_STAMP(27);
        float mx[2], l[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float v = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < WPS; ++kt) v = fmaxf(fmaxf(v, fmaxf(s[hh][kt][0], s[hh][kt][1])), fmaxf(s[hh][kt][2], s[hh][kt][3]));
            mx[hh] = row_max4(v);
        }
        if (rd == 1) STRIP_STAMP(28);
        f32x4 oacc[2][WPS];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float lsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < WPS; ++kt) {
                oacc[hh][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const unsigned kwd = (kt < 2 ? kl[hh] : kh[hh]) >> ((kt & 1) * 16 + 4 * gq);      // this lane's keys 16 kt + 4 g + r: bits 0..3
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[hh][kt][r] - mx[hh]);
                    lsum += p;
                    s[hh][kt][r] = ((kwd >> r) & 1u) ? p : 0.f;
                }
            }
            l[hh] = lsum;
        }
        if (rd == 1) STRIP_STAMP(29);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int kt = 0; kt < WPS; ++kt) {
                    const float4 v4 = vf[hh][kt];
                    oacc[hh][kt] = mfma4(r == 0 ? v4.x : r == 1 ? v4.y : r == 2 ? v4.z : v4.w, s[hh][kt][r], oacc[hh][kt]);
                }
        if (rd == 1) STRIP_STAMP(30);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * rd + hh;
            const float rl = 1.0f / row_sum4(l[hh]);
            const float ro = rl * dscale;                  // dropout's 1 / (1 - p) rides on the normalisation
            f32x4 o = oacc[hh][0];
#pragma unroll
            for (int kt = 1; kt < WPS; ++kt) o += oacc[hh][kt];
            O.v[h] = f32x4{o[0] * ro, o[1] * ro, o[2] * ro, o[3] * ro};
            const float mxn = mx[hh] * (1.0f / LOG2E);     // saved in natural units, as the separate kernels save it
            if (hh == 0) { stat[rd][0] = mxn; stat[rd][1] = rl; } else { stat[rd][2] = mxn; stat[rd][3] = rl; }
        }
    }
}

template <int D, int WPS>
__global__ __launch_bounds__(STRIP_THREADS) void seq_fwd_kernel(const SeqFwdArgs a, const SeqGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = D / 16, H = D / 16, SPW = STRIP_WAVES / WPS;
    // Workgroup -> (domain, tile): the live tiles are the FIRST workgroups, domain 0's then domain 1's.  The hardware deals workgroups
    // round-robin over the 8 XCDs and, inside an XCD, over its 4 shader engines of 8 CUs (index mod 32 picks the engine); a workgroup
    // here needs a whole CU (147 KB of LDS, 512 registers per lane), so at 256 live sequences on 256 CUs every engine must get
    // EXACTLY 8 of them: only a contiguous range of live indices guarantees that.  (Mappings that derive the domain from the index
    // -- by parity, or in chunks of 8 -- put 9 or more on some engines whenever the batch's domain split is uneven: two rounds,
    // 176 us instead of 91.)  The price is one scalar load (n0) in front of the first weight DMA.
    STRIP_STAMP(0);
    STRIP_RSTAMP(20);
    SEQ_SCHED(0);
    const int w = wave_id(), lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int sq = w / WPS, si = w - sq * WPS;
    int n0 = sg.B;
    if (sg.live != nullptr) n0 = sg.live[sg.B];
    // one sequence per workgroup: the live list holds domain 0's batch rows, then domain 1's, and tile i of the launch is its entry i --
    // requested together with n0 instead of behind it (one scalar round trip less in front of the first loads)
    int b_direct = 0;
    if (SPW == 1 && sg.live != nullptr) b_direct = sg.live[min((int)blockIdx.x, sg.B - 1)];
    const int n1 = sg.live != nullptr ? sg.B - n0 : sg.B;
    const int t0 = (n0 + SPW - 1) / SPW, t1 = (n1 + SPW - 1) / SPW;
    if ((int)blockIdx.x >= t0 + t1) { SEQ_SCHED(1); return; }
    const int g = (int)blockIdx.x >= t0 ? 1 : 0;
    const int tl = (int)blockIdx.x - (g ? t0 : 0);
    const int n_g = g ? n1 : n0, s0 = (g && sg.live != nullptr) ? n0 : 0;
    const int sidx = tl * SPW + sq;
    SeqRing<D> ring(smem);
    ring.first(a.L[0].w_in[g] + 1LL * D * D);
    float* kimg = smem + 2 * D * D;
    float* vimg = kimg + IMG_ROWS * IMG_COLS;
    int b = sidx;
    if (sg.live != nullptr) b = SPW == 1 ? b_direct : sg.live[min(s0 + sidx, sg.B - 1)];
    if (sidx >= n_g) b = 0;
    const bool seq_ok = sidx < n_g;
    const int t = si * 16 + m;
    StripRow row;
    row.ok = seq_ok && t < sg.T;
    row.local = b * sg.T + min(t, sg.T - 1);
    const unsigned phys = (unsigned)g * (unsigned)sg.M + (unsigned)(b * sg.T + t);
    row.off = row.ok ? phys * (unsigned)(D * 4) + 16u * (unsigned)gq : STRIP_OOB;
    const unsigned stat_off = row.ok ? phys * (unsigned)(H * 8) + 16u * (unsigned)gq : STRIP_OOB;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }

    StripRegs<D> X, Qn, Kr, Vr, Qr, O, R, Y;
    StripTm<D> tm;
    ColVec<D> bias, lw, lb;
    strip_load<D>(X, GBuf(a.x0, sg.act_bytes), row);
    const bool has_tm = a.tmq != nullptr;
    if (has_tm) strip_tm_load<D>(tm, GBuf(a.tmq, sg.tm_bytes), row);
    lw.load(a.L[0].ln1_w[g]); lb.load(a.L[0].ln1_b[g]);
    f32x4 acc[NT];
    f32x4 stat[H / 2];
    STRIP_STAMP(1);
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
        const SeqLayer& P = a.L[l];
        const bool last = l + 1 == a.n_layers;
        const GBuf gx(P.x, sg.act_bytes), gqn(P.qn, sg.act_bytes), gq_(P.q, sg.act_bytes), gk(P.k, sg.act_bytes), gv(P.v, sg.act_bytes),
                   go(P.o, sg.act_bytes), gst(P.stats, sg.stats_bytes), gr(P.r, sg.act_bytes), gy(P.y, sg.act_bytes), gh(P.h, sg.act_bytes);
        strip_layernorm<D>(Qn, X, lw, lb, a.ln_eps);
        STRIP_STAMP(2 + 10 * l);
        StripRow rowx = row;                                     // layer 0's input is the caller's buffer: nothing to write back
        rowx.off = l > 0 ? row.off : STRIP_OOB;
        {   // k = x Wk^T + bk  (the layer input's and Qn's global copies leave under these MFMAs)
            SEQ_FINE(0);
            const float* buf = ring.next();
            SEQ_FINE(1);
            bias.load(P.b_in[g] + D);
            strip_zero<D>(acc);
            strip_mma<D>(acc, X, buf, [&](int ct, int j) {
                ring.fetch(P.w_in[g] + 2LL * D * D, ct, j);
                spread<D>(gx, rowx, X, ct, j, 1);
                spread<D>(gqn, row, Qn, ct, j, 3);
            });
            SEQ_FINE(2);
            add_bias_s<D>(acc, bias);
            to_regs_s<D>(Kr, acc);
            SEQ_FINE(3);
            STRIP_STAMP(3 + 10 * l);
        }
        {   // v = x Wv^T + bv
            const float* buf = ring.next();
            SEQ_FINE(4);
            bias.load(P.b_in[g] + 2 * D);
            strip_zero<D>(acc);
            strip_mma<D>(acc, X, buf, [&](int ct, int j) { ring.fetch(P.w_in[g], ct, j); spread<D>(gk, row, Kr, ct, j, 1); });
            SEQ_FINE(5);
            add_bias_s<D>(acc, bias);
            to_regs_s<D>(Vr, acc);
            SEQ_FINE(6);
            STRIP_STAMP(4 + 10 * l);
        }
        {   // q = Qn Wq^T + bq
            const float* buf = ring.next();
            SEQ_FINE(7);
            bias.load(P.b_in[g]);
            strip_zero<D>(acc);
            strip_mma<D>(acc, Qn, buf, [&](int ct, int j) { ring.fetch(P.w_o[g], ct, j); spread<D>(gv, row, Vr, ct, j, 1); });
            SEQ_FINE(8);
            add_bias_s<D>(acc, bias);
            to_regs_s<D>(Qr, acc);
            SEQ_FINE(9);
            STRIP_STAMP(5 + 10 * l);
        }
        strip_store<D>(gq_, row, Qr);
        SEQ_FINE(10);
        seq_attention_fwd<D, WPS>(O, stat, Qr, Kr, Vr, kimg, vimg, si, t, sg.T, (unsigned long long)b * H, a.att_scale, a.train, seed,
                                  site_id(g, l, SITE_ATTN), step, a.spec, a.dscale);
        STRIP_STAMP(6 + 10 * l);
        SEQ_FINE(11);
        bias.load(P.b_o[g]); lw.load(P.ln2_w[g]); lb.load(P.ln2_b[g]);      // (not across the attention rounds: 96 registers)
        {   // the statistics of heads 2 gq, 2 gq + 1 of this lane's row: 16 contiguous bytes
            f32x4 sv = stat[0];
#pragma unroll
            for (int k = 1; k < H / 2; ++k) sv = (gq == k) ? stat[k] : sv;
            gst.store4(stat_off, sv);
        }
        {   // r = Qn + (o Wo^T + bo) ; y = LN2(r)
            SEQ_FINE(12);
            const float* buf = ring.next();
            SEQ_FINE(13);
            strip_zero<D>(acc);
            strip_mma<D>(acc, O, buf, [&](int ct, int j) { ring.fetch(P.w1[g], ct, j); spread<D>(go, row, O, ct, j, 1); });
            SEQ_FINE(14);
            add_bias_s<D>(acc, bias);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) R.v[ct] = Qn.v[ct] + acc[ct];
            strip_layernorm<D>(Y, R, lw, lb, a.ln_eps);
            SEQ_FINE(15);
            STRIP_STAMP(7 + 10 * l);
        }
        {   // h = relu(drop1(y C1^T + c1))
            const float* buf = ring.next();
            SEQ_FINE(16);
            bias.load(P.b1[g]);
            strip_zero<D>(acc);
            strip_mma<D>(acc, Y, buf, [&](int ct, int j) { ring.fetch(P.w2[g], ct, j); spread<D>(gr, row, R, ct, j, 1); });
            SEQ_FINE(17);
            add_bias_s<D>(acc, bias);
            to_regs_s<D>(Kr, acc);                               // Kr: the relu output from here on
            if (a.train) strip_dropout<D>(Kr, seed, site_id(g, l, SITE_FFN1), step, row.local, a.spec, a.ffn_scale);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) Kr.v[ct][r] = fmaxf(Kr.v[ct][r], 0.f);
            SEQ_FINE(18);
            STRIP_STAMP(8 + 10 * l);
        }
        {   // x' = (drop2(h C2^T + c2) + y) * ~tm
            const float* buf = ring.next();
            SEQ_FINE(19);
            bias.load(P.b2[g]);
            const SeqLayer& Pn = a.L[last ? l : l + 1];          // (the last layer refetches its own Wk into the free buffer: harmless)
            lw.load(Pn.ln1_w[g]); lb.load(Pn.ln1_b[g]);
            strip_zero<D>(acc);
            strip_mma<D>(acc, Kr, buf, [&](int ct, int j) {
                ring.fetch(Pn.w_in[g] + 1LL * D * D, ct, j);
                spread<D>(gy, row, Y, ct, j, 1);
                spread<D>(gh, row, Kr, ct, j, 3);
            });
            SEQ_FINE(20);
            add_bias_s<D>(acc, bias);
            to_regs_s<D>(X, acc);
            if (a.train) strip_dropout<D>(X, seed, site_id(g, l, SITE_FFN2), step, row.local, a.spec, a.ffn_scale);
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) X.v[ct] += Y.v[ct];
            if (has_tm) strip_apply_tm<D>(X, tm);
            SEQ_FINE(21);
            STRIP_STAMP(9 + 10 * l);
        }
    }
    strip_store<D>(GBuf(a.xout, sg.act_bytes), row, X);
    STRIP_STAMP(21);
    STRIP_RSTAMP(31);
    w_ring_wait();
    SEQ_SCHED(1);                                               // the last (redundant) weight fetch targets this workgroup's LDS
}

}  // namespace amid

using namespace amid;

#ifdef AMID_STRIP_STAMPS
extern "C" int amid_seq_sched_read(unsigned long long* host) {          // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_seq_sched_buf), sizeof(unsigned long long) * 1024 * 4);
}
extern "C" int amid_seq_fine_read(unsigned long long* host) {         // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_seq_fine_buf), sizeof(unsigned long long) * 8 * 64);
}
extern "C" int amid_seq_stamps_read(unsigned long long* host) {       // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_strip_stamp_buf), sizeof(unsigned long long) * STRIP_STAMP_WAVES * 32);
}
#endif

template <int D> static constexpr size_t seq_lds_bytes() { return (size_t)(2 * D * D + 2 * IMG_ROWS * IMG_COLS) * sizeof(float); }

// Which build runs: 0 = auto (the measured choice per shape), 1 = sasrec_seq.hip's whole-row waves, 2 = the N-split build's default
// split, 42 / 22 / 24 / 14 / 18 = an explicit (strips per sequence, column parts) pair of sasrec_seqn.hip.  Returns the previous value.
static int g_seq_fwd_variant = 0;
extern "C" int amid_sas_seq_fwd_variant(int v) {
    const int prev = g_seq_fwd_variant;
    if (v >= 0) g_seq_fwd_variant = v;
    return prev;
}

// 1 when the fused per-sequence forward covers this shape: 8 heads with D = 128 (head dim 16) or D = 64 (head dim 8: the N-split build only,
// fp32 products, p_drop = 0.5 or eval mode), T <= 64, activations within 2 GiB
extern "C" int amid_sas_seq_supported(int B, int T, int D, int H) {
    return ((D == 128 || D == 64) && H == 8 && T > 0 && T <= 64 && B > 0 && 2LL * B * T * D * 4 <= 0x7FFFFFF0LL) ? 1 : 0;
}

// Per-layer pointer arrays: the per-domain parameter families hold 2 * n_layers entries ordered [layer][domain], the saved-tensor
// families n_layers entries; x_in[l] = layer l's input rows (x_in[0] is read, x_in[l >= 1] written), xout = the last layer's output.
// the gather K1 folded into the forward's prologue (seq_fwd.h SeqFwdArgs::g_*): table [n_rows, D], idx = the step's full index list
// [seq_d1 B T | seq_d2 B T | items B ni], pos = the two position tables, items (optional) = where the samples' ni item rows go, done
// (optional) = the step state whose step_done the launch re-joins
struct SeqGather { const float* table; const int* idx; const float* pos[2]; float* items; int ni; StepState* done; };

// set around a call by the *_p1_f32 entries: the pieces build multiplies ONE piece per operand (bf16 products on the three-plane images' hi planes)
static thread_local int tl_one_piece = 0;

static int seq_fwd_impl(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                        const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                        const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                        float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                        const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                        const void* step_state, int train, float p_drop, const void* w16, void* stream, int w16_planes = 1,
                        float* const* ln_stat = nullptr, const HeadArgs* head = nullptr, bool infer = false, const SeqGather* gat = nullptr) {
    AMID_CHECK_ARG(n_layers >= 1 && n_layers <= 2 && x_in && (xout || head) && ln1_w && ln1_b && w_in && b_in && w_o && b_o && ln2_w && ln2_b && w1 && b1 &&
                   w2 && b2 && (!train || step_state));
    AMID_CHECK_ARG(infer || ((ln_stat || (qn && y)) && q && k && v && o && stats && r && h));
    if (ln_stat != nullptr && !(w16 != nullptr && w16_planes == 3 && D == 128)) return AMID_ERR_UNSUPPORTED;      // the piece forward only
    // an inference forward saves nothing (SeqGeom::save_bytes = 0): the producer-side pieces build at D 128 only, T a multiple-of-16 strip count it tiles
    if (infer && !(w16 != nullptr && w16_planes == 3 && D == 128 && !train && head == nullptr)) return AMID_ERR_UNSUPPORTED;
    if (!amid_sas_seq_supported(B, T, D, H)) return AMID_ERR_UNSUPPORTED;
    SeqFwdArgs a = {};
    a.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        SeqLayer& P = a.L[l];
        for (int g = 0; g < 2; ++g) {
            const int i = 2 * l + g;
            AMID_CHECK_ARG(ln1_w[i] && ln1_b[i] && w_in[i] && b_in[i] && w_o[i] && b_o[i] && ln2_w[i] && ln2_b[i] && w1[i] && b1[i] && w2[i] && b2[i]);
            P.ln1_w[g] = ln1_w[i]; P.ln1_b[g] = ln1_b[i]; P.w_in[g] = w_in[i]; P.b_in[g] = b_in[i]; P.w_o[g] = w_o[i]; P.b_o[g] = b_o[i];
            P.ln2_w[g] = ln2_w[i]; P.ln2_b[g] = ln2_b[i]; P.w1[g] = w1[i]; P.b1[g] = b1[i]; P.w2[g] = w2[i]; P.b2[g] = b2[i];
        }
        if (infer) {
            AMID_CHECK_ARG(l > 0 || x_in[0]);
            P.x = l == 0 ? const_cast<float*>(x_in[0]) : nullptr;
            P.qn = P.q = P.k = P.v = P.o = P.stats = P.r = P.y = P.h = P.ln_stat = nullptr;
            continue;
        }
        AMID_CHECK_ARG(x_in[l] && q[l] && k[l] && v[l] && o[l] && stats[l] && r[l] && h[l] && (ln_stat ? ln_stat[l] != nullptr : (qn[l] && y[l])));
        P.x = const_cast<float*>(x_in[l]); P.qn = ln_stat ? nullptr : qn[l]; P.q = q[l]; P.k = k[l]; P.v = v[l]; P.o = o[l]; P.stats = stats[l];
        P.r = r[l]; P.y = ln_stat ? nullptr : y[l]; P.h = h[l]; P.ln_stat = ln_stat ? ln_stat[l] : nullptr;
    }
    a.x0 = x_in[0]; a.xout = xout; a.tmq = tmq; a.ln_eps = ln_eps;
    if (gat != nullptr) {      // the gather as the workgroups' prologue: the producer-side pieces build, a live list, p = 0.5 or eval
        if (!(w16 != nullptr && w16_planes == 3 && D == 128 && live != nullptr)) return AMID_ERR_UNSUPPORTED;
        AMID_CHECK_ARG(gat->table && gat->idx && gat->pos[0] && gat->pos[1] && tmq && gat->ni >= 0 && (gat->items == nullptr || gat->ni > 0));
        a.g_table = gat->table; a.g_idx = gat->idx; a.g_pos[0] = gat->pos[0]; a.g_pos[1] = gat->pos[1]; a.g_items = gat->items; a.g_ni = gat->ni;
        a.g_scale = (train && p_drop > 0.f) ? 1.0f / (1.0f - p_drop) : 1.0f;
        a.g_done = gat->done;
    }
    a.w16 = (const unsigned short*)w16; a.w16_planes = w16_planes;
    a.one_piece = (tl_one_piece && w16 != nullptr && w16_planes == 3) ? 1 : 0;
    a.att_scale = sqrtf(1.0f / (float)(D / H));
    a.st = (const StepState*)step_state;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.spec = drop_spec(p_drop);
    a.dscale = a.ffn_scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    SeqGeom sg;
    sg.B = B; sg.T = T; sg.M = B * T; sg.live = live;
    const long long bytes = 2LL * B * T * D * 4;
    sg.act_bytes = (unsigned)bytes; sg.tm_bytes = (unsigned)(bytes / 16); sg.stats_bytes = (unsigned)(2LL * B * T * H * 8);
    sg.save_bytes = infer ? 0u : sg.act_bytes; sg.save_stats_bytes = infer ? 0u : sg.stats_bytes;
    if (head != nullptr) return launch_seqn_fwd(a, sg, D, 0, stream, head);       // (the default N-split build or nothing)
    if (infer) return launch_seqn_fwd(a, sg, D, 0, stream, nullptr);             // (the default split of the pieces build or nothing)
    {
        int v = g_seq_fwd_variant;
        if (v == 0) v = 2;                                 // auto: the N-split build wins at every measured shape (profiles/r03_*)
        if (w16 != nullptr && v == 1) return AMID_ERR_UNSUPPORTED;      // the whole-row build has no bf16 products
        if (v != 1) {
            int rc = launch_seqn_fwd(a, sg, D, v == 2 ? 0 : v, stream);
            if (rc == AMID_ERR_UNSUPPORTED && w16 != nullptr) rc = launch_seqn_fwd(a, sg, D, 0, stream);
            if (rc != AMID_ERR_UNSUPPORTED || w16 != nullptr) return rc;
        }
        if (D != 128) return AMID_ERR_UNSUPPORTED;          // the whole-row build below: D = 128 only
    }
    const int wps = T <= 16 ? 1 : T <= 32 ? 2 : 4, spw = STRIP_WAVES / wps;
    const int tiles = (B + spw - 1) / spw;
    const int grid = live != nullptr ? tiles + 1 : 2 * tiles;      // the live tiles of both domains (one more when both are ragged) / every tile
    const size_t lds = seq_lds_bytes<128>();
    auto launch = [&](auto kern) -> int {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        kern<<<grid, STRIP_THREADS, lds, (hipStream_t)stream>>>(a, sg);
        e = hipGetLastError();
        return e == hipSuccess ? AMID_OK : (int)e;
    };
    if (wps == 1) return launch(seq_fwd_kernel<128, 1>);
    if (wps == 2) return launch(seq_fwd_kernel<128, 2>);
    return launch(seq_fwd_kernel<128, 4>);
}

extern "C" int amid_sas_seq_fwd_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                    const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                    const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                    const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                                    float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                                    const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                    const void* step_state, int train, float p_drop, void* stream) {
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, qn, q, k, v, o, stats, r, y, h, tmq,
                        ln_eps, B, T, D, H, live, step_state, train, p_drop, nullptr, stream);
}

// amid_sas_seq_fwd_split_lnstat_f32 with the train step's head (amid_head_fwd_bwd_own_vec_f32: LN_last + mean over T -- model_seq.py:385,
// :432-434 --, predictModule.forward -- :40-54 --, the masked BCE term and dLoss/dp -- train_sr.py:203-212 --, the scorer's backward and
// LN_last') on the tail of every workgroup: a live sequence IS a sample, so the head of sample b runs where its rows were just computed.
// The last layer's output is stored only when xout != NULL (only the head read it; tests ask for it).  Same arithmetic in the same order as the two launches: same bits.
// T 33 ... 64, D 128, live != NULL, head_lds_floats(D, hid) within the forward's plane slots (hid <= 32); otherwise AMID_ERR_UNSUPPORTED
// and nothing is enqueued.
extern "C" int amid_sas_seq_fwd_split_lnstat_head_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                                      const float* const* w_in, const float* const* b_in, const float* const* w_o,
                                                      const float* const* b_o, const float* const* ln2_w, const float* const* ln2_b,
                                                      const float* const* w1, const float* const* b1, const float* const* w2,
                                                      const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                                      float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h,
                                                      const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                                      const void* step_state, int train, float p_drop, const void* w16x3,
                                                      const float* const* last_ln_w, const float* const* last_ln_b, const float* items,
                                                      const float* sw1, const float* sb1, const float* sw2, const float* sb2, const float* labels,
                                                      const long long* domain_id, int NI, int hid, float* u, float* p1, float* p2, float* dp1,
                                                      float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg,
                                                      void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr && ln_stat != nullptr && live != nullptr && last_ln_w != nullptr && last_ln_b != nullptr);
    HeadArgs ha;
    if (int e = head_own_vec_args(ha, last_ln_w, last_ln_b, items, sw1, sb1, sw2, sb2, labels, domain_id, B, T, NI, D, hid, ln_eps, u, p1, p2, dp1,
                                  dp2, loss_part, dx, ditems, ln_part, hidg)) return e;
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, nullptr, q, k, v, o, stats, r, nullptr,
                        h, tmq, ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, stream, 3, ln_stat, &ha);
}

// The same forward with the twelve projections' matrix products on v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, fp32
// accumulation; LayerNorm, the attention core, residuals, dropout, everything stored stays fp32): w16 = the weights' bf16 fragment images
// written by amid_sas_weights_bf16 for this step's weights.  BASELINE.json configs[2]; arithmetic: model_seq.py:371-383.
extern "C" int amid_sas_seq_fwd_bf16w_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                          const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                          const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                          const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                                          float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                                          const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                          const void* step_state, int train, float p_drop, const void* w16, void* stream) {
    AMID_CHECK_ARG(w16 != nullptr);
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, qn, q, k, v, o, stats, r, y, h, tmq,
                        ln_eps, B, T, D, H, live, step_state, train, p_drop, w16, stream);
}

// bf16 fragment images of n square [D][D] fp32 matrices (row-major [out n][in k]; transposed != 0: of their transposes):
// dst16 [n][D][D / 8 chunks][8] bf16, chunk 4 s + g of row n = W[n][32 s + 4 g + 0..3], W[n][32 s + 16 + 4 g + 0..3] -- the eight k a lane
// group supplies in k-step s of v_mfma_f32_16x16x32_bf16 when the other operand sits in the strip kernels' C layout (csrc/sasrec_seqn.hip).
namespace amid {
struct W16Args { const float* src[48]; int n; };
// planes = 3: every element as hi + mid + lo (csrc/bf16_pieces.h), one image per piece: dst [n][3][D][D] bf16
__global__ __launch_bounds__(256) void weights_bf16_kernel(const W16Args a, unsigned short* __restrict__ dst, int D, int transposed, int planes) {
    weights_image_block(a.src[blockIdx.y], dst + (size_t)blockIdx.y * planes * D * D, D, transposed, planes, blockIdx.x, gridDim.x);
}
}  // namespace amid

static int weights_bf16(const float* const* src, int n, int D, int transposed, int planes, void* dst16, void* stream) {
    AMID_CHECK_ARG(src && dst16 && n > 0 && n <= 48 && D > 0 && (D % 32) == 0 && (planes == 1 || planes == 3));
    amid::W16Args a;
    a.n = n;
    for (int i = 0; i < n; ++i) { AMID_CHECK_ARG(src[i]); a.src[i] = src[i]; }
    const int per = (D * (D / 8) + 255) / 256;
    amid::weights_bf16_kernel<<<dim3(per, n), 256, 0, (hipStream_t)stream>>>(a, (unsigned short*)dst16, D, transposed, planes);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_sas_weights_bf16(const float* const* src, int n, int D, int transposed, void* dst16, void* stream) {
    return weights_bf16(src, n, D, transposed, 1, dst16, stream);
}

// planes = 3: three images per matrix, dst16 [n][3][D][D] bf16 -- every element as hi + mid + lo, three bf16 pieces whose sum is the
// fp32 element exactly (csrc/bf16_pieces.h): what amid_sas_seq_fwd_split_f32 consumes
extern "C" int amid_sas_weights_bf16_planes(const float* const* src, int n, int D, int transposed, int planes, void* dst16, void* stream) {
    return weights_bf16(src, n, D, transposed, planes, dst16, stream);
}

// amid_sas_seq_fwd_f32 with the twelve projections' products on the bf16 matrix cores AT FP32 ACCURACY: every operand element as three
// bf16 pieces, six piece pairs per product (csrc/seqn_parts.h SeqRing16x3 / part_mma16x6); w16x3 = amid_sas_weights_bf16_planes(...,
// planes = 3, ...) images of THIS step's weights, [layer][domain][q, k, v, o, conv1, conv2][3][D][D] bf16.  N-split builds only (D 128).
extern "C" int amid_sas_seq_fwd_split_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                          const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                          const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                          const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                                          float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                                          const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                          const void* step_state, int train, float p_drop, const void* w16x3, void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr);
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, qn, q, k, v, o, stats, r, y, h, tmq,
                        ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, stream, 3);
}

// amid_sas_seq_fwd_split_f32 as an INFERENCE forward (evaluation, train_sr.py:31-128: model(..., False) under no_grad): the same products
// in the same order -- xout comes out bit-identical -- but none of the tensors a backward would read is stored (18 stores of 6.5 MB at the
// headline shape); x0 = layer 0's input rows.  D = 128; no dropout.
extern "C" int amid_sas_seq_fwd_split_infer_f32(int n_layers, const float* x0, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                                const float* const* w_in, const float* const* b_in, const float* const* w_o,
                                                const float* const* b_o, const float* const* ln2_w, const float* const* ln2_b,
                                                const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                                                const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                                const void* w16x3, void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr && x0 != nullptr && xout != nullptr);
    const float* xin[2] = {x0, nullptr};
    return seq_fwd_impl(n_layers, xin, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, nullptr, tmq, ln_eps, B, T, D, H, live, nullptr, 0, 0.f, w16x3, stream, 3, nullptr, nullptr,
                        true);
}

// The three forwards of the folded step / the evaluation batch with the gather K1 as their workgroups' PROLOGUE (round 6): layer 0's input
// rows are built from table[idx] + pos (dropout, == 0 mask: embed.hip's arithmetic, the same bits) by the workgroup that encodes the sequence
// and stored to x_in[0] / tmq for the backward (not in the inference forward); the samples' item rows go to `items` (the head on the tail
// and the scorer sums read them there); the launch re-joins StepState::step_done.  The weight images must be current when the launch starts
// (amid_step_head_w16_f32's riders, amid_sas_weights_bf16_planes).  Replaces amid_embed_fwd_w16_f32 + the forward: one launch fewer.
extern "C" int amid_sas_seq_fwd_gather_head_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                                const float* const* w_in, const float* const* b_in, const float* const* w_o,
                                                const float* const* b_o, const float* const* ln2_w, const float* const* ln2_b,
                                                const float* const* w1, const float* const* b1, const float* const* w2,
                                                const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                                float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h,
                                                unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                                void* step_state, int train, float p_drop, const void* w16x3,
                                                const float* const* last_ln_w, const float* const* last_ln_b, float* items,
                                                const float* sw1, const float* sb1, const float* sw2, const float* sb2, const float* labels,
                                                const long long* domain_id, int NI, int hid, float* u, float* p1, float* p2, float* dp1,
                                                float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg,
                                                const float* table, const int* idx_all, const float* pos0, const float* pos1, void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr && ln_stat != nullptr && live != nullptr && last_ln_w != nullptr && last_ln_b != nullptr && step_state);
    HeadArgs ha;
    if (int e = head_own_vec_args(ha, last_ln_w, last_ln_b, items, sw1, sb1, sw2, sb2, labels, domain_id, B, T, NI, D, hid, ln_eps, u, p1, p2, dp1,
                                  dp2, loss_part, dx, ditems, ln_part, hidg)) return e;
    const SeqGather gat{table, idx_all, {pos0, pos1}, items, NI, (StepState*)step_state};
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, nullptr, q, k, v, o, stats, r, nullptr,
                        h, tmq, ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, stream, 3, ln_stat, &ha, false, &gat);
}
extern "C" int amid_sas_seq_fwd_gather_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w,
                                           const float* const* ln1_b, const float* const* w_in, const float* const* b_in,
                                           const float* const* w_o, const float* const* b_o, const float* const* ln2_w,
                                           const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                           const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q,
                                           float* const* k, float* const* v, float* const* o, float* const* stats, float* const* r,
                                           float* const* h, unsigned char* tmq, float ln_eps, int B, int T, int D, int H,
                                           const int* live, void* step_state, int train, float p_drop, const void* w16x3, float* items, int NI,
                                           const float* table, const int* idx_all, const float* pos0, const float* pos1, void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr && ln_stat != nullptr && step_state);
    const SeqGather gat{table, idx_all, {pos0, pos1}, items, NI, (StepState*)step_state};
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, nullptr, q, k, v, o, stats, r, nullptr, h,
                        tmq, ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, stream, 3, ln_stat, nullptr, false, &gat);
}
// amid_sas_seq_fwd_gather_head_f32 / amid_sas_seq_fwd_gather_f32 with the twelve projection products on ONE bf16 piece per operand (compute = "bf16"
// on the folded step, round 6): the same launches -- three-plane images, pieces in the exchange (the LayerNorm statistics still see the exact
// rows) -- reading only the hi planes and multiplying only the hi pieces: bf16 products with fp32 accumulation, a sixth of the matrix work.
extern "C" int amid_sas_seq_fwd_gather_head_p1_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                                   const float* const* w_in, const float* const* b_in, const float* const* w_o,
                                                   const float* const* b_o, const float* const* ln2_w, const float* const* ln2_b,
                                                   const float* const* w1, const float* const* b1, const float* const* w2,
                                                   const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                                   float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h,
                                                   unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                                   void* step_state, int train, float p_drop, const void* w16x3,
                                                   const float* const* last_ln_w, const float* const* last_ln_b, float* items,
                                                   const float* sw1, const float* sb1, const float* sw2, const float* sb2, const float* labels,
                                                   const long long* domain_id, int NI, int hid, float* u, float* p1, float* p2, float* dp1,
                                                   float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg,
                                                   const float* table, const int* idx_all, const float* pos0, const float* pos1, void* stream) {
    tl_one_piece = 1;
    const int rc = amid_sas_seq_fwd_gather_head_f32(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, ln_stat, q, k, v, o,
                                                    stats, r, h, tmq, ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, last_ln_w, last_ln_b, items,
                                                    sw1, sb1, sw2, sb2, labels, domain_id, NI, hid, u, p1, p2, dp1, dp2, loss_part, dx, ditems, ln_part, hidg,
                                                    table, idx_all, pos0, pos1, stream);
    tl_one_piece = 0;
    return rc;
}
extern "C" int amid_sas_seq_fwd_gather_p1_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w,
                                              const float* const* ln1_b, const float* const* w_in, const float* const* b_in,
                                              const float* const* w_o, const float* const* b_o, const float* const* ln2_w,
                                              const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                              const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q,
                                              float* const* k, float* const* v, float* const* o, float* const* stats, float* const* r,
                                              float* const* h, unsigned char* tmq, float ln_eps, int B, int T, int D, int H,
                                              const int* live, void* step_state, int train, float p_drop, const void* w16x3, float* items, int NI,
                                              const float* table, const int* idx_all, const float* pos0, const float* pos1, void* stream) {
    tl_one_piece = 1;
    const int rc = amid_sas_seq_fwd_gather_f32(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, ln_stat, q, k, v, o, stats, r,
                                               h, tmq, ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, items, NI, table, idx_all, pos0, pos1, stream);
    tl_one_piece = 0;
    return rc;
}
extern "C" int amid_sas_seq_fwd_gather_infer_f32(int n_layers, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                                 const float* const* w_in, const float* const* b_in, const float* const* w_o,
                                                 const float* const* b_o, const float* const* ln2_w, const float* const* ln2_b,
                                                 const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                                                 float ln_eps, int B, int T, int D, int H, const int* live, const void* w16x3,
                                                 const float* table, const int* idx_all, const float* pos0, const float* pos1, void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr && xout != nullptr);
    // (nothing is stored but xout: x_in[0] and the mask bytes are only names here -- the mask word stays in registers)
    const float* xin[2] = {xout, nullptr};
    const SeqGather gat{table, idx_all, {pos0, pos1}, nullptr, 0, nullptr};
    return seq_fwd_impl(n_layers, xin, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, nullptr, nullptr, nullptr, nullptr,
                        nullptr, nullptr, nullptr, nullptr, nullptr, (const unsigned char*)xout, ln_eps, B, T, D, H, live, nullptr, 0, 0.f, w16x3, stream, 3,
                        nullptr, nullptr, true, &gat);
}

// amid_sas_seq_fwd_split_f32 that saves SEVEN tensors per layer instead of nine: qn = LN1(x) and y = LN2(r) are not stored; ln_stat[l]
// [2 B T][4] receives every row's (mean, rstd) of LayerNorm 1 and of LayerNorm 2 instead (the backward strips rebuild the normalised rows
// from x / r already; amid_sas_wgrad_rows_sort_ln_f32 applies (row - mean) rstd gamma + beta while it stages the operand).  D = 128.
extern "C" int amid_sas_seq_fwd_split_lnstat_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w,
                                                 const float* const* ln1_b, const float* const* w_in, const float* const* b_in,
                                                 const float* const* w_o, const float* const* b_o, const float* const* ln2_w,
                                                 const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                                 const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q,
                                                 float* const* k, float* const* v, float* const* o, float* const* stats, float* const* r,
                                                 float* const* h, const unsigned char* tmq, float ln_eps, int B, int T, int D, int H,
                                                 const int* live, const void* step_state, int train, float p_drop, const void* w16x3,
                                                 void* stream) {
    AMID_CHECK_ARG(w16x3 != nullptr && ln_stat != nullptr);
    return seq_fwd_impl(n_layers, x_in, xout, ln1_w, ln1_b, w_in, b_in, w_o, b_o, ln2_w, ln2_b, w1, b1, w2, b2, nullptr, q, k, v, o, stats, r, nullptr, h,
                        tmq, ln_eps, B, T, D, H, live, step_state, train, p_drop, w16x3, stream, 3, ln_stat);
}
