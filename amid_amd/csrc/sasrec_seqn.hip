// The one-launch SASRec encoder forward, N-split build: NS waves share a 16-row strip of a sequence, each owning D / NS output columns
// (= 8 / NS heads) of every product.  Same operations, operands, dropout counters and saved tensors as sasrec_seq.hip (reference:
// Log2feats.forward model_seq.py:371-383, nn.MultiheadAttention as called at :374, PointWiseFeedForward :322-326).
//
// Why a second build.  In sasrec_seq.hip a wave owns a strip and ALL columns, so a launch lasts as long as one wave's serial chain:
// 12 weight slabs x 256 MFMAs + the attention core, with nothing to cover its epilogues, barriers and LDS round trips (one wave per
// SIMD), and at T <= 32 only 2 * T / 32 of the chip's SIMDs have a strip at all (batch 256, T 20: 128 workgroups on 256 CUs).  Here
//   * a workgroup is ONE sequence: WPS strips (T <= 16 * WPS) x NS column parts = WPS * NS waves.  WPS = 4, NS = 2: eight waves, two per
//     SIMD, the partner's matrix work under a wave's epilogue / softmax / waits.  WPS = 2, NS = 2 (T <= 32): four waves, one per SIMD, and
//     every sequence has a workgroup of its own -- a wave's chain is 128 MFMAs per slab instead of 256.
//   * a product needs the whole row of its operand (K = D), a wave produces D / NS columns of the result: after every product the
//     strip's waves exchange their parts through LDS ([strip][column tile][lane] float4: lane (m, g) writes and reads the SAME slot, so
//     the exchange is conflict-free and needs no layout thought).  Per-row work on whole rows (LayerNorm) is done redundantly by the NS
//     waves of the strip; element-wise work (bias, relu, dropout, masks, residuals, stores) on the own columns only.
//   * attention: a wave handles its own heads (its columns of q / k / v ARE whole heads).  The K and V^T images of ALL heads sit in the
//     weight ring's idle slab (the one that held Wq; Wo is landing in the other one, W1 is fetched into it only after the core): one
//     barrier pair per layer instead of one per two heads.  K image = a weight image [key][D] with the ring's swizzle, so S = Qs K^T
//     is one k-tile of strip_mma per head; V^T image [dim][64 keys] as in sasrec_seq.hip.
// LDS: ring 2 x 64 KB + exchange WPS x 8 KB = 160 KB at WPS = 4 (the whole CU).
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "attention_mfma.h"
#include "seq_fwd.h"
#include "seqn_parts.h"
#include "head_parts.h"
#include <type_traits>

namespace amid {


#ifdef AMID_STRIP_STAMPS
static __device__ unsigned long long amid_seqn_stamp_buf[8 * 64];
#define SEQN_STAMP(i) do { if (blockIdx.x == 0 && lane_id() == 0 && l == 1) amid_seqn_stamp_buf[wave_id() * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// (slots 62 / 63: the kernel's first / last instruction; slots 60 / 61 the same moments on the constant 100 MHz clock, and the LAST workgroup
// stamps slots 58 / 59 with that clock too: the shader clock during the kernel, and how long after workgroup 0 the last one ends)
#define SEQN_STAMP0(i) do { if (lane_id() == 0) { if (blockIdx.x == 0) { amid_seqn_stamp_buf[wave_id() * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
    amid_seqn_stamp_buf[wave_id() * 64 + (i) - 2] = __builtin_amdgcn_s_memrealtime(); } \
    else if (blockIdx.x == gridDim.x - 1) amid_seqn_stamp_buf[wave_id() * 64 + (i) - 4] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define SEQN_STAMP(i) do { } while (0)
#define SEQN_STAMP0(i) do { } while (0)
#endif


// attention of the wave's 16 query rows over its own heads h = c0 .. c0 + NCT - 1; K / V^T images of the whole sequence in LDS
constexpr int NIMG_KEYS = 64;
// PAIR (D = 64 with the reference's 8 heads: head dim 8): an own column tile is TWO heads.  Per head the K operand is zeroed in the other
// head's lane groups (the matrix instruction contracts over the lane groups: groups 0, 1 carry the first head's dims, 2, 3 the
// second's), P~ V is computed for all 16 dims of the tile and a lane keeps its own head's result (attention_mfma.h attn_fwd_head<true>).
// st_max / st_rl: per own head, NCT (x 2) of them.
template <int D, int WPS, int NCT, bool PAIR, int SI = -1>
__device__ __forceinline__ void seqn_attention(PartRegs<NCT>& O, float (&st_max)[PAIR ? 2 * NCT : NCT], float (&st_rl)[PAIR ? 2 * NCT : NCT],
                                               const PartRegs<NCT>& Q, const float* __restrict__ kimg, const float* __restrict__ vimg, int c0,
                                               int si_rt, int m, int gq, int t, int T, unsigned long long rowbase_bh, float scale, int train,
                                               unsigned long long seed, unsigned site, unsigned step, unsigned spec, float dscale) {
    // SI >= 0: the wave's strip index as a compile-time constant (the caller switches on it): every `kt <= si` below folds, the four heads
    // become ONE straight-line region and the scheduler overlaps a head's softmax with its neighbour's matrix instructions (with the
    // runtime index every tile sits behind a branch)
    const int si = SI >= 0 ? SI : si_rt;
    constexpr int NE = PAIR ? 2 : 1, NH = NE * NCT;
    static_assert(NH <= 4, "a lane group draws the keep word of one own head");
    const int qrow = min(t, T - 1);
    f32x4 mdiag;                                           // causal mask of the diagonal tile (kt = si) as the accumulators' initial value
#pragma unroll
    for (int r = 0; r < 4; ++r) mdiag[r] = (4 * gq + r > m) ? -INFINITY : 0.f;
    // dropout keep word (64 keys) of this lane's query row for own head gq (lane groups past the own heads idle); own head lh's word is
    // then fetched from group lh
    unsigned kwl = ~0u, kwh = ~0u;
    if (train && gq < NH) {
        const unsigned long long kw = row_keep_word(seed, site, step, (rowbase_bh + NE * c0 + gq) * T + qrow, T, spec);
        kwl = (unsigned)kw; kwh = (unsigned)(kw >> 32);
    }
    const float qscale = scale * LOG2E;
    // key tiles above the wave's own strip are all in the future (causal): only tiles kt <= si are computed.  (Workgroups of eight waves
    // pair strip si with strip WPS - 1 - si on a SIMD, so every SIMD carries WPS + 1 tiles per head pair instead of 2 .. 2 WPS.)
#pragma unroll
    for (int hl = 0; hl < NCT; ++hl) {
        const int h = c0 + hl;
        float4 kf[WPS], vf[WPS];
#pragma unroll
        for (int kt = 0; kt < WPS; ++kt) {
            if (kt <= si) {
                kf[kt] = lds_ld4(kimg + (kt * 16 + m) * D + 4 * ((4 * h + gq) ^ m));
                vf[kt] = lds_ld4(vimg + (h * 16 + m) * NIMG_KEYS + 4 * ((kt * 4 + gq) ^ m));
            }
        }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int lh = NE * hl + e;
        const bool mine = !PAIR || (gq >> 1) == e;         // this lane's dims belong to head e of the tile
        const unsigned kl = train ? bcast_group(kwl, lh) : ~0u, kh = train ? bcast_group(kwh, lh) : ~0u;
        f32x4 s[WPS];
#pragma unroll
        for (int kt = 0; kt < WPS; ++kt) s[kt] = (kt == si) ? mdiag : (kt < si) ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float qs = Q.v[hl][r] * qscale;
#pragma unroll
            for (int kt = 0; kt < WPS; ++kt) {
                if (kt <= si) {
                    const float4 k4 = kf[kt];
                    const float kr = r == 0 ? k4.x : r == 1 ? k4.y : r == 2 ? k4.z : k4.w;
                    s[kt] = mfma4(mine ? kr : 0.f, qs, s[kt]);
                }
            }
        }
        float v = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < WPS; ++kt)
            if (kt <= si) v = fmaxf(fmaxf(v, fmaxf(s[kt][0], s[kt][1])), fmaxf(s[kt][2], s[kt][3]));
        const float mx = row_max4(v);
        float lsum = 0.f;
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < WPS; ++kt) {
            if (kt <= si) {
                const unsigned kwd = (kt < 2 ? kl : kh) >> ((kt & 1) * 16 + 4 * gq);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                    lsum += p;
                    s[kt][r] = ((kwd >> r) & 1u) ? p : 0.f;
                }
            }
        }
        // P~ V: the key tiles in ascending order, one accumulator per tile summed at the end -- the order sasrec_seq.hip adds them in
        f32x4 oacc[WPS];
#pragma unroll
        for (int kt = 0; kt < WPS; ++kt) oacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int kt = 0; kt < WPS; ++kt) {
                if (kt <= si) {
                    const float4 v4 = vf[kt];
                    oacc[kt] = mfma4(r == 0 ? v4.x : r == 1 ? v4.y : r == 2 ? v4.z : v4.w, s[kt][r], oacc[kt]);
                }
            }
        const float rl = 1.0f / row_sum4(lsum);
        const float ro = rl * dscale;
        o = oacc[0];
#pragma unroll
        for (int kt = 1; kt < WPS; ++kt) o += oacc[kt];
        if (mine) O.v[hl] = f32x4{o[0] * ro, o[1] * ro, o[2] * ro, o[3] * ro};
        st_max[lh] = mx * (1.0f / LOG2E);
        st_rl[lh] = rl;
      }
    }
}


// P3 (with BF): the weight images are three bf16 planes per weight and the products run on six piece pairs -- fp32 accuracy
// (seqn_parts.h SeqRing16x3, part_mma16x6)
template <int D, int WPS, int NS, bool BF, bool P3 = false>
__global__ __launch_bounds__(64 * WPS * NS) void seqn_fwd_kernel(const SeqFwdArgs a, const SeqGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = D / 16, NW = WPS * NS, NCT = NT / NS;
    constexpr bool PAIR = D == 64;                         // 8 heads of 8 dims: two per column tile (seqn_attention)
    constexpr int H = PAIR ? 2 * NT : NT, NH = PAIR ? 2 * NCT : NCT;        // heads; own heads of a wave
    static_assert((D == 128 || D == 64) && NT % NS == 0 && NCT >= 1 && (!BF || D == 128) && (!P3 || BF), "");
    const int w = wave_id(), lane = lane_id(), m = lane & 15, gq = lane >> 4;
    // waves w and w + 4 share a SIMD: strip si beside strip WPS - 1 - si (the causal attention core costs si + 1 key tiles per head)
    const int part = w / WPS;
    const int si = (NW == 8 && w >= 4) ? WPS - 1 - (w % WPS) : w % WPS;
    const int c0 = part * NCT;                              // first own column tile = first own head
    // workgroup -> sequence: the live sequences are the first workgroups, domain 0's then domain 1's (sasrec_seq.hip: a contiguous
    // range keeps the deal over XCDs and shader engines even)
    int n0 = sg.B, b = 0, g = 0;
    if (sg.live != nullptr) {
        n0 = sg.live[sg.B];
        b = sg.live[min((int)blockIdx.x, sg.B - 1)];
        if ((int)blockIdx.x >= sg.B) return;
        g = (int)blockIdx.x >= n0 ? 1 : 0;
    } else {
        g = (int)blockIdx.x >= sg.B ? 1 : 0;
        b = (int)blockIdx.x - g * sg.B;
    }
    // LDS: fp32 -- [ring 2 x D D floats][exchange]; the attention images alias the ring's idle slab.  bf16 -- [ring 2 x D D / 2][images
    // 2 x 64 D floats][exchange]: a 32 KB slab cannot hold them
    using Ring = typename std::conditional<P3, SeqRing16x3<D, NW>, typename std::conditional<BF, SeqRing16<D, NW>, SeqRingN<D, NW>>::type>::type;
    Ring ring(smem);
    auto w16 = [&](int layer, int which) { return a.w16 + ((size_t)((layer * 2 + g) * 6 + which)) * (P3 ? 3 : 1) * D * D; };     // q, k, v, o, c1, c2
    if constexpr (BF) ring.first(w16(0, 1)); else ring.first(a.L[0].w_in[g] + 1LL * D * D);
    // (the images of 64 keys -- 2 x 64 D floats -- fit the idle 64 KB slab of the fp32 ring at D = 128 only: bf16 slabs and D = 64's 16 KB
    // slabs are too small, those builds keep a region of their own behind the ring)
    constexpr bool SEP_IMG = (BF && !P3) || D == 64;          // (P3: the images alias the ring's M + L plane slots)
    constexpr int RING_F = (BF && !P3) ? D * D : 2 * D * D;
    float* const img16 = smem + RING_F;
    float* xb = smem + RING_F + (SEP_IMG ? 2 * NIMG_KEYS * D : 0) + si * (NT * 64 * 4);      // this strip's exchange slots
    const int t = si * 16 + m;
    const bool row_ok = t < sg.T;
    const int local = b * sg.T + min(t, sg.T - 1);
    const unsigned phys = (unsigned)g * (unsigned)sg.M + (unsigned)(b * sg.T + t);
    const unsigned off_full = row_ok ? phys * (unsigned)(D * 4) + 16u * (unsigned)gq : STRIP_OOB;
    const unsigned off_own = row_ok ? off_full + (unsigned)c0 * 64u : STRIP_OOB;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }

    StripRegs<D> F;                                         // the whole-row operand of the next product
    PartRegs<NCT> Xo, Qno, Ko, Vo, Qo, Oo, Ro, Yo, Ho, bias;
    ColVec<D> lw, lb;
    {
        StripRow row; row.ok = row_ok; row.local = local; row.off = off_full;
        strip_load<D>(F, GBuf(a.x0, sg.act_bytes), row);
    }
    part_load<NCT>(Xo, GBuf(a.x0, sg.act_bytes), off_own);
    unsigned tmw[NCT];                                      // the "== 0" bits of the own column tiles (one byte per column quad)
    const bool has_tm = a.tmq != nullptr;
    if (has_tm) {
        const GBuf gtm(a.tmq, sg.tm_bytes);
        const unsigned tbase = row_ok ? phys * (unsigned)(D / 4) + (unsigned)c0 * 4u : STRIP_OOB;
#pragma unroll
        for (int c = 0; c < NCT; ++c) tmw[c] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(gtm.r, (int)(tbase + 4 * c), 0, 0);
    }
    lw.load(a.L[0].ln1_w[g]); lb.load(a.L[0].ln1_b[g]);
    f32x4 acc[NCT];
    SEQN_STAMP0(62);
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
        const SeqLayer& P = a.L[l];
        const bool last = l + 1 == a.n_layers;
        const GBuf gx(P.x, sg.act_bytes), gqn(P.qn, sg.act_bytes), gq_(P.q, sg.act_bytes), gk(P.k, sg.act_bytes), gv(P.v, sg.act_bytes),
                   go(P.o, sg.act_bytes), gst(P.stats, sg.stats_bytes), gr(P.r, sg.act_bytes), gy(P.y, sg.act_bytes), gh(P.h, sg.act_bytes);
        SEQN_STAMP(0);
        float* bufk = ring.next();                         // Wk has landed; the previous layer's output parts are visible
        SEQN_STAMP(1);
        if (l > 0) {
            xchg_read<D>(F, xb);
#pragma unroll
            for (int c = 0; c < NCT; ++c) {                 // the own tiles again (runtime slot address, static registers)
                const float4 t4 = lds_ld4(xb + ((c0 + c) * 64 + lane) * 4);
                Xo.v[c] = f32x4{t4.x, t4.y, t4.z, t4.w};
            }
        }
        {   // Qn = LN1(x): the own columns only -- for the residual, the saved copy and, through the exchange slots behind the k
            // product, the whole row of the q product (held in registers through the k and v products it cost 32 VGPRs)
            // (the gains were requested late in the previous product -- the prologue for layer 0)
            float mean, rstd;
            strip_stats<D>(F, a.ln_eps, mean, rstd);
            PartRegs<NCT> lwo, lbo;
            own_cols<D, NCT>(lwo, lw, P.ln1_w[g], part, c0); own_cols<D, NCT>(lbo, lb, P.ln1_b[g], part, c0);
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Qno.v[c][r] = (Xo.v[c][r] - mean) * rstd * lwo.v[c][r] + lbo.v[c][r];
        }
        SEQN_STAMP(2);
        {   // k = x Wk^T + bk
            part_cols<NCT>(bias, P.b_in[g] + D, c0);
            seqn_product<D, NCT, BF>(acc, F, bufk, ring, P.w_in[g] + 2LL * D * D, BF ? w16(l, 2) : nullptr, c0,
                                     [&](int ct, int j) { part_spread<NCT>(gqn, off_own, Qno, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Ko.v[c] = acc[c] + bias.v[c];
        }
        // every wave of the strip is past its reads of the exchange slots (the v product's ring barrier lies between this write and
        // the q product's read)
        xchg_write<NCT>(xb, c0, Qno);
        SEQN_STAMP(3);
        {   // v = x Wv^T + bv
            const float* buf = ring.next();
            SEQN_STAMP(4);
            part_cols<NCT>(bias, P.b_in[g] + 2 * D, c0);
            seqn_product<D, NCT, BF>(acc, F, buf, ring, P.w_in[g], BF ? w16(l, 0) : nullptr, c0,
                                     [&](int ct, int j) { part_spread<NCT>(gk, off_own, Ko, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Vo.v[c] = acc[c] + bias.v[c];
        }
        SEQN_STAMP(5);
        float* img;
        {   // q = Qn Wq^T + bq
            float* buf = ring.next();
            SEQN_STAMP(6);
            if constexpr (P3) { img = ring.images(); ring.hold_next = true; } else img = SEP_IMG ? img16 : buf;
            part_cols<NCT>(bias, P.b_in[g], c0);
            xchg_read<D>(F, xb);                           // the whole row of Qn (x is dead behind the v product)
            seqn_product<D, NCT, BF>(acc, F, buf, ring, P.w_o[g], BF ? w16(l, 3) : nullptr, c0,
                                     [&](int ct, int j) { part_spread<NCT>(gv, off_own, Vo, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Qo.v[c] = acc[c] + bias.v[c];
        }
        SEQN_STAMP(7);
        part_store<NCT>(gq_, off_own, Qo);
        // ---- attention core: images of all heads in the slab that held Wq
        float* kimg = img;
        float* vimg = img + NIMG_KEYS * D;
        // (lane-derived LDS addresses and masks of the core are recomputed per layer: hoisted out of the layer loop they cost ~80
        // registers that two waves per SIMD do not have)
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int m_ = lane_l & 15, gl_ = lane_l >> 4;
        lds_barrier();                                     // every wave has read its last Wq fragment
        {
            const int m = m_, gq = gl_;
            const int R = si * 16 + m;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int h = c0 + c;
                lds_st4(kimg + R * D + 4 * ((4 * h + gq) ^ m), Ko.v[c]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = h * 16 + 4 * gq + r;     // row of the transposed image (d & 15 = 4 gq + r); this lane's key R: column
                    lds_st1(vimg + d * NIMG_KEYS + (((R >> 2) ^ (d & 15)) * 4) + (R & 3), Vo.v[c][r]);
                }
            }
        }
        lds_barrier();
        SEQN_STAMP(8);
        float st_max[NH], st_rl[NH];
        seqn_attention<D, WPS, NCT, PAIR>(Oo, st_max, st_rl, Qo, kimg, vimg, c0, si, m_, gl_, si * 16 + m_, sg.T, (unsigned long long)b * H, a.att_scale, a.train, seed,
                                    site_id(g, l, SITE_ATTN), step, a.spec, a.dscale);
        SEQN_STAMP(9);
        {   // row statistics of the own heads: [2M][H][2] floats
            const unsigned so = row_ok ? phys * (unsigned)(H * 8) + (unsigned)(PAIR ? 2 * c0 : c0) * 8u : STRIP_OOB;
            if constexpr (NH >= 2) {
                f32x4 sv = f32x4{st_max[0], st_rl[0], st_max[1], st_rl[1]};
#pragma unroll
                for (int k = 1; k < NH / 2; ++k) sv = (gq == k) ? f32x4{st_max[2 * k], st_rl[2 * k], st_max[2 * k + 1], st_rl[2 * k + 1]} : sv;
                gst.store4(gq < NH / 2 ? so + 16u * (unsigned)gq : STRIP_OOB, sv);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gq == 0 ? st_max[0] : st_rl[0]), gst.r,
                                                      (int)(gq < 2 ? so + 4u * (unsigned)gq : STRIP_OOB), 0, 0);
            }
        }
        // ---- r = Qn + (o Wo^T + bo) ; y = LN2(r)
        xchg_write<NCT>(xb, c0, Oo);                       // (the previous exchange's reads lie behind several barriers)
        part_cols<NCT>(bias, P.b_o[g], c0);
        {
            const float* buf = ring.next();
            SEQN_STAMP(10);
            xchg_read<D>(F, xb);
            seqn_product<D, NCT, BF>(acc, F, buf, ring, P.w1[g], BF ? w16(l, 4) : nullptr, c0,
                                     [&](int ct, int j) { part_spread<NCT>(go, off_own, Oo, ct, j, 1); },
                                     [&]() { lw.load(P.ln2_w[g]); lb.load(P.ln2_b[g]); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Ro.v[c] = Qno.v[c] + (acc[c] + bias.v[c]);
        }
        SEQN_STAMP(11);
        lds_barrier();                                     // every wave of the strip has read the o parts
        xchg_write<NCT>(xb, c0, Ro);
        uint4 rr1 = make_uint4(0, 0, 0, 0), rr2 = rr1;
        if (a.train) {                                     // the row's dropout words of both feed-forward sites, ahead of their use
            rr1 = rng_call(seed, (unsigned long long)local * D >> 7, site_id(g, l, SITE_FFN1), step);
            rr2 = rng_call(seed, (unsigned long long)local * D >> 7, site_id(g, l, SITE_FFN2), step);
        }
        {   // h = relu(drop1(y C1^T + c1))
            const float* buf = ring.next();
            SEQN_STAMP(12);
            xchg_read<D>(F, xb);                           // the whole row of r
            {
                float mean, rstd;
                strip_stats<D>(F, a.ln_eps, mean, rstd);
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) F.v[ct][r] = (F.v[ct][r] - mean) * rstd * lw.v[ct][r] + lb.v[ct][r];
                PartRegs<NCT> lwo, lbo;
                own_cols<D, NCT>(lwo, lw, P.ln2_w[g], part, c0); own_cols<D, NCT>(lbo, lb, P.ln2_b[g], part, c0);
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Yo.v[c][r] = (Ro.v[c][r] - mean) * rstd * lwo.v[c][r] + lbo.v[c][r];
            }
            SEQN_STAMP(13);
            part_cols<NCT>(bias, P.b1[g], c0);
            seqn_product<D, NCT, BF>(acc, F, buf, ring, P.w2[g], BF ? w16(l, 5) : nullptr, c0,
                                     [&](int ct, int j) { part_spread<NCT>(gr, off_own, Ro, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Ho.v[c] = acc[c] + bias.v[c];
            if (a.train) part_dropout<NCT>(Ho, rr1, c0, a.spec, a.ffn_scale, (local * D) & 127);
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ho.v[c][r] = fmaxf(Ho.v[c][r], 0.f);
        }
        SEQN_STAMP(14);
        lds_barrier();
        xchg_write<NCT>(xb, c0, Ho);
        {   // x' = (drop2(h C2^T + c2) + y) * ~tm
            const float* buf = ring.next();
            SEQN_STAMP(15);
            xchg_read<D>(F, xb);
            part_cols<NCT>(bias, P.b2[g], c0);
            const SeqLayer& Pn = a.L[last ? l : l + 1];          // (the last layer refetches its own Wk into the free buffer: harmless)
            seqn_product<D, NCT, BF>(acc, F, buf, ring, Pn.w_in[g] + 1LL * D * D, BF ? w16(last ? l : l + 1, 1) : nullptr, c0, [&](int ct, int j) {
                part_spread<NCT>(gy, off_own, Yo, ct, j, 1);
                part_spread<NCT>(gh, off_own, Ho, ct, j, 3);
            }, [&]() { if (!last) { lw.load(Pn.ln1_w[g]); lb.load(Pn.ln1_b[g]); } });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Xo.v[c] = acc[c] + bias.v[c];
            if (a.train) part_dropout<NCT>(Xo, rr2, c0, a.spec, a.ffn_scale, (local * D) & 127);
            const int sh = 8 * gq;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                Xo.v[c] += Yo.v[c];
                if (has_tm) {
                    const unsigned bits = tmw[c] >> sh;
                    Xo.v[c][0] = (bits & 1u) ? 0.f : Xo.v[c][0];
                    Xo.v[c][1] = (bits & 2u) ? 0.f : Xo.v[c][1];
                    Xo.v[c][2] = (bits & 4u) ? 0.f : Xo.v[c][2];
                    Xo.v[c][3] = (bits & 8u) ? 0.f : Xo.v[c][3];
                }
            }
        }
        SEQN_STAMP(16);
        if (!last) {
            lds_barrier();
            xchg_write<NCT>(xb, c0, Xo);                   // read at the top of the next layer, behind its ring barrier
            part_store<NCT>(GBuf(a.L[l + 1].x, sg.act_bytes), off_own, Xo);     // the next layer's saved input
        } else {
            part_store<NCT>(GBuf(a.xout, sg.act_bytes), off_own, Xo);
        }
    }
    SEQN_STAMP0(63);
    w_ring_wait();                                          // the last (redundant) weight fetch targets this workgroup's LDS
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same forward, products on bf16 pieces (fp32 accuracy), with the pieces made by the PRODUCER of every operand: the exchange between
// the column parts of a strip carries operand fragments of pieces (seqn_parts.h SeqRing3 / xp_write / part_mma_xp).  LDS: three 32 KB
// plane slots [M][L][H] + WPS x 12 KB of exchange; the attention images take M + L between the q and out-projection products.
// LayerNorm statistics come from the row put back together (hi + mid + lo is the value exactly); only the own columns' gains are held.
// HEAD: the train step's head on the workgroup's tail (seqn_fwd_px_head_kernel below) -- the last layer's output goes to an LDS image instead
// of a.xout and head_own_rows_body (csrc/head_parts.h) runs on it.
template <int D, int WPS, int NS, bool HEAD, bool ONE = false>
__device__ __forceinline__ void seqn_fwd_px_body(const SeqFwdArgs& a, const SeqGeom& sg, float* const smem, const HeadArgs* ha) {
    constexpr int NT = D / 16, NW = WPS * NS, NCT = NT / NS;
    constexpr int H = NT, NH = NCT;
    static_assert(D == 128 && NCT >= 2 && NCT % 2 == 0, "a wave owns whole k-steps of the next product's operand");
    const int w = wave_id(), lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int part = w / WPS;
    const int si = (NW == 8 && w >= 4) ? WPS - 1 - (w % WPS) : w % WPS;
    const int c0 = part * NCT;
    int n0 = sg.B, b = 0, g = 0;
    if (sg.live != nullptr) {
        n0 = sg.live[sg.B];
        b = sg.live[min((int)blockIdx.x, sg.B - 1)];
        if ((int)blockIdx.x >= sg.B) return;
        g = (int)blockIdx.x >= n0 ? 1 : 0;
    } else {
        g = (int)blockIdx.x >= sg.B ? 1 : 0;
        b = (int)blockIdx.x - g * sg.B;
    }
    using Ring = SeqRing3<D, NW>;
    Ring ring(smem);
    ring.one = ONE;
    auto w16 = [&](int layer, int which) { return a.w16 + ((size_t)((layer * 2 + g) * 6 + which)) * 3 * D * D; };     // q, k, v, o, c1, c2
    ring.first(w16(0, 1));
    float* const xps = smem + 3 * Ring::SLAB + si * XpStrip<D>::FLOATS;         // this strip's exchange slots
    float* const stat = smem + 3 * Ring::SLAB + WPS * XpStrip<D>::FLOATS;        // [2][strip][part][16 rows] LayerNorm row sums
    const int t = si * 16 + m;
    const bool row_ok = t < sg.T;
    const int local = b * sg.T + min(t, sg.T - 1);
    const unsigned phys = (unsigned)g * (unsigned)sg.M + (unsigned)(b * sg.T + t);
    const unsigned off_full = row_ok ? phys * (unsigned)(D * 4) + 16u * (unsigned)gq : STRIP_OOB;
    const unsigned off_own = row_ok ? off_full + (unsigned)c0 * 64u : STRIP_OOB;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }

    PartRegs<NCT> Xo, Qno, Ko, Vo, Qo, Oo, Ro, Yo, Ho, bias, lwo, lbo;
    unsigned tmw[NCT];
    const bool has_tm = a.tmq != nullptr;
    if (a.g_table != nullptr) {
        // ---- the gather K1 as this workgroup's prologue (embed.hip embed_fwd_kernel's arithmetic for the rows of THIS sequence): lane (m, gq)
        // builds its 16 bytes of row t in each own column tile -- x = table[id] + pos[t]; the == 0 bits; the row's ONE Philox call (p = 0.5:
        // a bit per element) keeps or drops; masked elements zeroed -- keeps them as layer 0's input and its mask word, and (when a backward
        // follows: save_bytes != 0) stores the row and the mask bytes where K1 put them
        const int id = a.g_idx[(long long)g * sg.M + b * sg.T + min(t, sg.T - 1)];
        const float* __restrict__ trow = a.g_table + (long long)id * D;
        const float* __restrict__ prow = a.g_pos[g] + (long long)min(t, sg.T - 1) * D;
        uint4 rb = make_uint4(0u, 0u, 0u, 0u);
        if (a.train) rb = rng_call(seed, (unsigned long long)local, site_id(g, 0, SITE_EMB), step);
        const GBuf gx0(a.x0, sg.save_bytes);
        const GBuf gtmw(a.tmq, has_tm ? sg.save_bytes / 16u : 0u);
        const float sc = a.g_scale;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int col = (c0 + c) * 16 + 4 * gq;
            const float4 tv = ld4_global(trow + col), pv = ld4_global(prow + col);
            float4 x = make_float4(tv.x + pv.x, tv.y + pv.y, tv.z + pv.z, tv.w + pv.w);
            const unsigned bits = (x.x == 0.f ? 1u : 0u) | (x.y == 0.f ? 2u : 0u) | (x.z == 0.f ? 4u : 0u) | (x.w == 0.f ? 8u : 0u);
            if (a.train) {
                const unsigned wsel = rng_word(rb, (c0 + c) >> 1);
                const unsigned kw = wsel >> ((((c0 + c) & 1) * 4 + gq) * 4);
                x = make_float4((kw & 1u) ? x.x * sc : 0.f, (kw & 2u) ? x.y * sc : 0.f, (kw & 4u) ? x.z * sc : 0.f, (kw & 8u) ? x.w * sc : 0.f);
            }
            if (bits & 1u) x.x = 0.f;
            if (bits & 2u) x.y = 0.f;
            if (bits & 4u) x.z = 0.f;
            if (bits & 8u) x.w = 0.f;
            Xo.v[c] = row_ok ? f32x4{x.x, x.y, x.z, x.w} : f32x4{0.f, 0.f, 0.f, 0.f};
            tmw[c] = row_ok ? bits << (8 * gq) : 0u;
            gx0.store4(off_own + c * 64, Xo.v[c]);
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, gtmw.r, row_ok ? (int)(phys * (unsigned)(D / 4) + (unsigned)((c0 + c) * 4 + gq)) : (int)STRIP_OOB, 0, 0);
        }
        if (a.g_items != nullptr) {                         // the sample's item rows: a plain gather (two rows a wave and pass)
            const long long ibase = 2LL * sg.M + (long long)b * a.g_ni;
            for (int n0 = 2 * w; n0 < a.g_ni; n0 += 2 * NW) {
                const int n = n0 + (lane >> 5);
                if (n < a.g_ni) {
                    const long long iid = a.g_idx[ibase + n];
                    const float4 v = ld4_global(a.g_table + iid * D + 4 * (lane & 31));
                    st4_global(a.g_items + ((long long)b * a.g_ni + n) * D + 4 * (lane & 31), v);
                }
            }
        }
        if (a.g_done != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) a.g_done->step_done = a.g_done->step;
    } else {
        part_load<NCT>(Xo, GBuf(a.x0, sg.act_bytes), off_own);
        if (has_tm) {
            const GBuf gtm(a.tmq, sg.tm_bytes);
            const unsigned tbase = row_ok ? phys * (unsigned)(D / 4) + (unsigned)c0 * 4u : STRIP_OOB;
#pragma unroll
            for (int c = 0; c < NCT; ++c) tmw[c] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(gtm.r, (int)(tbase + 4 * c), 0, 0);
        }
    }
    xp_write<NCT>(xps, c0, Xo);                             // (rows past T arrive as zeros: the buffer descriptor's out-of-range reads)
    f32x4 acc[NCT];
    SEQN_STAMP0(62);
#pragma unroll 1
    for (int l = 0; l < a.n_layers; ++l) {
        const SeqLayer& P = a.L[l];
        const bool last = l + 1 == a.n_layers;
        const GBuf gqn(P.qn, sg.save_bytes), gq_(P.q, sg.save_bytes), gk(P.k, sg.save_bytes), gv(P.v, sg.save_bytes),
                   go(P.o, sg.save_bytes), gst(P.stats, sg.save_stats_bytes), gr(P.r, sg.save_bytes), gy(P.y, sg.save_bytes), gh(P.h, sg.save_bytes);
        part_cols<NCT>(lwo, P.ln1_w[g], c0); part_cols<NCT>(lbo, P.ln1_b[g], c0);
        SEQN_STAMP(0);
        ring.next();                                       // Wk has landed; the row of x is in the exchange slots
        SEQN_STAMP(1);
        // P.ln_stat: qn and y are not stored (SeqLayer::ln_stat); one lane per row (group 0 of column part 0) stores the row's statistics
        const bool lnst = P.ln_stat != nullptr;
        const GBuf gls(P.ln_stat, lnst ? sg.act_bytes / (unsigned)(D / 4) : 0u);
        const unsigned so_ln = (lnst && row_ok && gq == 0 && part == 0) ? phys * 16u : STRIP_OOB;
        {   // Qn = LN1(x) on the own columns
            StripRegs<D> F;
            xp_row<D>(F, xps);
            float mean, rstd;
            strip_stats<D>(F, a.ln_eps, mean, rstd);
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Qno.v[c][r] = (Xo.v[c][r] - mean) * rstd * lwo.v[c][r] + lbo.v[c][r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mean), gls.r, (int)so_ln, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rstd), gls.r, (int)(so_ln + (so_ln == STRIP_OOB ? 0u : 4u)), 0, 0);
        }
        SEQN_STAMP(2);
        {   // k = x Wk^T + bk
            part_cols<NCT>(bias, P.b_in[g] + D, c0);
            seqn_product_xp<D, NCT, ONE>(acc, xps, ring, w16(l, 2), c0, [&](int ct, int j) { if (!lnst) part_spread<NCT>(gqn, off_own, Qno, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Ko.v[c] = acc[c] + bias.v[c];
        }
        SEQN_STAMP(3);
        {   // v = x Wv^T + bv
            ring.next();
            SEQN_STAMP(4);
            part_cols<NCT>(bias, P.b_in[g] + 2 * D, c0);
            seqn_product_xp<D, NCT, ONE>(acc, xps, ring, w16(l, 0), c0, [&](int ct, int j) { part_spread<NCT>(gk, off_own, Ko, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Vo.v[c] = acc[c] + bias.v[c];
        }
        SEQN_STAMP(5);
        lds_barrier();                                     // every wave of the strip has read the last fragment of x
        xp_write<NCT>(xps, c0, Qno);
        {   // q = Qn Wq^T + bq
            ring.next();
            SEQN_STAMP(6);
            ring.hold_next = true;                         // M + L hold the attention images behind this product
            part_cols<NCT>(bias, P.b_in[g], c0);
            seqn_product_xp<D, NCT, ONE>(acc, xps, ring, w16(l, 3), c0, [&](int ct, int j) { part_spread<NCT>(gv, off_own, Vo, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Qo.v[c] = acc[c] + bias.v[c];
        }
        SEQN_STAMP(7);
        part_store<NCT>(gq_, off_own, Qo);
        float* kimg = ring.images();
        float* vimg = kimg + NIMG_KEYS * D;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        const int m_ = lane_l & 15, gl_ = lane_l >> 4;
        lds_barrier();                                     // (M / L were read in the q product's first pass, two barriers ago; kept for the images' sake)
        {
            const int R = si * 16 + m_;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const int h = c0 + c;
                lds_st4(kimg + R * D + 4 * ((4 * h + gl_) ^ m_), Ko.v[c]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = h * 16 + 4 * gl_ + r;
                    lds_st1(vimg + d * NIMG_KEYS + (((R >> 2) ^ (d & 15)) * 4) + (R & 3), Vo.v[c][r]);
                }
            }
        }
        lds_barrier();
        SEQN_STAMP(8);
        float st_max[NH], st_rl[NH];
        {   // the strip index as a compile-time constant of the core (seqn_attention SI): 8.9 - 16.2 k cycles -> 5.1 - 12.6 k per layer
            auto att = [&](auto SIC) {
                seqn_attention<D, WPS, NCT, false, decltype(SIC)::value>(Oo, st_max, st_rl, Qo, kimg, vimg, c0, si, m_, gl_, si * 16 + m_, sg.T, (unsigned long long)b * H,
                                                                          a.att_scale, a.train, seed, site_id(g, l, SITE_ATTN), step, a.spec, a.dscale);
            };
            if constexpr (WPS == 4) {
                if (si == 0) att(std::integral_constant<int, 0>{}); else if (si == 1) att(std::integral_constant<int, 1>{});
                else if (si == 2) att(std::integral_constant<int, 2>{}); else att(std::integral_constant<int, 3>{});
            } else if constexpr (WPS == 2) {
                if (si == 0) att(std::integral_constant<int, 0>{}); else att(std::integral_constant<int, 1>{});
            } else att(std::integral_constant<int, 0>{});
        }
        SEQN_STAMP(9);
        {
            const unsigned so = row_ok ? phys * (unsigned)(H * 8) + (unsigned)c0 * 8u : STRIP_OOB;
            f32x4 sv = f32x4{st_max[0], st_rl[0], st_max[1], st_rl[1]};
#pragma unroll
            for (int k = 1; k < NH / 2; ++k) sv = (gq == k) ? f32x4{st_max[2 * k], st_rl[2 * k], st_max[2 * k + 1], st_rl[2 * k + 1]} : sv;
            gst.store4(gq < NH / 2 ? so + 16u * (unsigned)gq : STRIP_OOB, sv);
        }
        // ---- r = Qn + (o Wo^T + bo) ; y = LN2(r)
        xp_write<NCT>(xps, c0, Oo);                        // (the q product's reads of the slots lie behind the core's barriers)
        part_cols<NCT>(bias, P.b_o[g], c0);
        {
            ring.next();
            SEQN_STAMP(10);
            seqn_product_xp<D, NCT, ONE>(acc, xps, ring, w16(l, 4), c0, [&](int ct, int j) { part_spread<NCT>(go, off_own, Oo, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Ro.v[c] = Qno.v[c] + (acc[c] + bias.v[c]);
        }
        SEQN_STAMP(11);
        part_cols<NCT>(lwo, P.ln2_w[g], c0); part_cols<NCT>(lbo, P.ln2_b[g], c0);
        uint4 rr1 = make_uint4(0, 0, 0, 0), rr2 = rr1;
        if (a.train) {
            rr1 = rng_call(seed, (unsigned long long)local * D >> 7, site_id(g, l, SITE_FFN1), step);
            rr2 = rng_call(seed, (unsigned long long)local * D >> 7, site_id(g, l, SITE_FFN2), step);
        }
        SEQN_STAMP(12);
        {   // y = LN2(r) on the own columns: r itself is nobody's operand, so only the row sums cross the column parts (two floats per row and
            // part through `stat`; the sums are taken in the order strip_stats takes them over a whole row, part by part)
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c) sm += (Ro.v[c][0] + Ro.v[c][1]) + (Ro.v[c][2] + Ro.v[c][3]);
            sm = row_sum4(sm);
            if (gq == 0) lds_st1(stat + (si * NS + part) * 16 + m, sm);
            lds_barrier();
            float tot = 0.f;
#pragma unroll
            for (int p2 = 0; p2 < NS; ++p2) tot += *(__attribute__((address_space(3))) float*)(stat + (si * NS + p2) * 16 + m);
            const float mean = tot * (1.0f / D);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = Ro.v[c][r] - mean; q = fmaf(d, d, q); }
            q = row_sum4(q);
            if (gq == 0) lds_st1(stat + (WPS * NS + si * NS + part) * 16 + m, q);
            lds_barrier();
            float qt = 0.f;
#pragma unroll
            for (int p2 = 0; p2 < NS; ++p2) qt += *(__attribute__((address_space(3))) float*)(stat + (WPS * NS + si * NS + p2) * 16 + m);
            const float rstd = 1.0f / sqrtf(qt * (1.0f / D) + a.ln_eps);
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Yo.v[c][r] = (Ro.v[c][r] - mean) * rstd * lwo.v[c][r] + lbo.v[c][r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mean), gls.r, (int)(so_ln + (so_ln == STRIP_OOB ? 0u : 8u)), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rstd), gls.r, (int)(so_ln + (so_ln == STRIP_OOB ? 0u : 12u)), 0, 0);
        }
        xp_write<NCT>(xps, c0, Yo);                         // (the o fragments were read two barriers ago)
        {   // h = relu(drop1(y C1^T + c1))
            ring.next();
            SEQN_STAMP(13);
            part_cols<NCT>(bias, P.b1[g], c0);
            seqn_product_xp<D, NCT, ONE>(acc, xps, ring, w16(l, 5), c0, [&](int ct, int j) { part_spread<NCT>(gr, off_own, Ro, ct, j, 1); });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Ho.v[c] = acc[c] + bias.v[c];
            if (a.train) part_dropout<NCT>(Ho, rr1, c0, a.spec, a.ffn_scale, (local * D) & 127);
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ho.v[c][r] = fmaxf(Ho.v[c][r], 0.f);
        }
        SEQN_STAMP(14);
        lds_barrier();
        xp_write<NCT>(xps, c0, Ho);
        {   // x' = (drop2(h C2^T + c2) + y) * ~tm
            ring.next();
            SEQN_STAMP(15);
            part_cols<NCT>(bias, P.b2[g], c0);
            seqn_product_xp<D, NCT, ONE>(acc, xps, ring, w16(last ? l : l + 1, 1), c0, [&](int ct, int j) {
                if (!lnst) part_spread<NCT>(gy, off_own, Yo, ct, j, 1);
                part_spread<NCT>(gh, off_own, Ho, ct, j, 3);
            });
#pragma unroll
            for (int c = 0; c < NCT; ++c) Xo.v[c] = acc[c] + bias.v[c];
            if (a.train) part_dropout<NCT>(Xo, rr2, c0, a.spec, a.ffn_scale, (local * D) & 127);
            const int sh = 8 * gq;
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                Xo.v[c] += Yo.v[c];
                if (has_tm) {
                    const unsigned bits = tmw[c] >> sh;
                    Xo.v[c][0] = (bits & 1u) ? 0.f : Xo.v[c][0];
                    Xo.v[c][1] = (bits & 2u) ? 0.f : Xo.v[c][1];
                    Xo.v[c][2] = (bits & 4u) ? 0.f : Xo.v[c][2];
                    Xo.v[c][3] = (bits & 8u) ? 0.f : Xo.v[c][3];
                }
            }
        }
        SEQN_STAMP(16);
        if (!last) {
            lds_barrier();
            xp_write<NCT>(xps, c0, Xo);
            part_store<NCT>(GBuf(a.L[l + 1].x, sg.save_bytes), off_own, Xo);
        } else if (!HEAD || a.xout != nullptr) {             // (with the head on the tail only its tests ask for the rows)
            part_store<NCT>(GBuf(a.xout, sg.act_bytes), off_own, Xo);
        }
    }
    SEQN_STAMP0(63);
    w_ring_wait();
    if constexpr (HEAD) {
        // The sample's head where its sequence was encoded: LN_last + mean over T, the scorer, the masked loss term, the scorer's backward
        // and LN_last' -- what amid_head_fwd_bwd_own_vec_f32's workgroup of sample b does, on the rows this workgroup still holds (no
        // xout round trip, no launch).  Every wave's weight DMA has landed (above) and behind the barrier nobody reads the plane slots or
        // the exchange slots any more: the head's LDS carve takes the former, the rows' image xl [WPS 16][D + 4] the latter.
        static_assert(NW == 8, "the head's phases are written for 512 threads");
        lds_barrier();
        float* const xl = smem + 3 * Ring::SLAB;
        constexpr int XLD = D + 4;
        static_assert(WPS * 16 * XLD <= WPS * XpStrip<D>::FLOATS, "the rows' image fits in the exchange slots");
#pragma unroll
        for (int c = 0; c < NCT; ++c) lds_st4(xl + t * XLD + (c0 + c) * 16 + 4 * gq, Xo.v[c]);
        __syncthreads();
        OwnRows R;
        own_rows_take_lds(xl, XLD, sg.T, D, R);
        head_own_rows_body<true>(*ha, smem, b, g, R);
    }
}

template <int D, int WPS, int NS, bool ONE = false>
__global__ __launch_bounds__(64 * WPS * NS) void seqn_fwd_px_kernel(const SeqFwdArgs a, const SeqGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    seqn_fwd_px_body<D, WPS, NS, false, ONE>(a, sg, smem, nullptr);
}

template <int D, int WPS, int NS, bool ONE = false>
__global__ __launch_bounds__(64 * WPS * NS) void seqn_fwd_px_head_kernel(const SeqFwdArgs a, const SeqGeom sg, const HeadArgs ha) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    seqn_fwd_px_body<D, WPS, NS, true, ONE>(a, sg, smem, &ha);
}

template <int D, int WPS, int NS>
static int seqn_launch_px(const SeqFwdArgs& a, const SeqGeom& sg, void* stream, const HeadArgs* head = nullptr) {
    constexpr size_t lds = (size_t)(3 * (D * D / 2) + WPS * XpStrip<D>::FLOATS + 2 * WPS * NS * 16) * sizeof(float);
    const int grid = sg.live != nullptr ? sg.B : 2 * sg.B;
    if (head != nullptr) {
        if constexpr ((WPS == 4 && NS == 2) || (WPS == 2 && NS == 4)) {      // (the builds launch_seqn_fwd asks the head for: T 33 ... 64; 17 ... 32, round 6)
            // (the head's carve inside the three plane slots, its rows T <= 16 HEAD_CHUNK / 2, a workgroup per LIVE sequence = per sample)
            if (sg.live == nullptr || head_lds_floats(D, head->hid) > (size_t)3 * (D * D / 2) || sg.T > 16 * (HEAD_CHUNK / 2) || head->D != D ||
                head->B != sg.B || head->T != sg.T)
                return AMID_ERR_UNSUPPORTED;
            if (a.one_piece) {
                auto kern1 = seqn_fwd_px_head_kernel<D, WPS, NS, true>;
                static unsigned long long attr_done1 = 0;
                if (int rc = lds_attr_once((const void*)kern1, lds, attr_done1)) return rc;
                kern1<<<grid, 64 * WPS * NS, lds, (hipStream_t)stream>>>(a, sg, *head);
                hipError_t e1 = hipGetLastError();
                return e1 == hipSuccess ? AMID_OK : (int)e1;
            }
            auto kern = seqn_fwd_px_head_kernel<D, WPS, NS>;
            static unsigned long long attr_done = 0;
            if (int rc = lds_attr_once((const void*)kern, lds, attr_done)) return rc;
            kern<<<grid, 64 * WPS * NS, lds, (hipStream_t)stream>>>(a, sg, *head);
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? AMID_OK : (int)e;
        } else {
            return AMID_ERR_UNSUPPORTED;
        }
    }
    if (a.one_piece) {
        if constexpr ((WPS == 4 && NS == 2) || (WPS == 2 && NS == 4)) {      // (the builds a folded step asks for)
            auto kern1 = seqn_fwd_px_kernel<D, WPS, NS, true>;
            static unsigned long long attr_done1 = 0;
            if (int rc = lds_attr_once((const void*)kern1, lds, attr_done1)) return rc;
            kern1<<<grid, 64 * WPS * NS, lds, (hipStream_t)stream>>>(a, sg);
            hipError_t e1 = hipGetLastError();
            return e1 == hipSuccess ? AMID_OK : (int)e1;
        } else {
            return AMID_ERR_UNSUPPORTED;
        }
    }
    auto kern = seqn_fwd_px_kernel<D, WPS, NS>;
    static unsigned long long attr_done = 0;
    if (int rc = lds_attr_once((const void*)kern, lds, attr_done)) return rc;
    kern<<<grid, 64 * WPS * NS, lds, (hipStream_t)stream>>>(a, sg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}

template <int D, int WPS, bool BF, bool P3 = false> static constexpr size_t seqn_lds_bytes() {
    return (size_t)(((BF && !P3) ? D * D : 2 * D * D) + (((BF && !P3) || D == 64) ? 2 * NIMG_KEYS * D : 0) + WPS * (D / 16) * 64 * 4) * sizeof(float);
}

template <int D, int WPS, int NS, bool BF, bool P3 = false>
static int seqn_launch_t(const SeqFwdArgs& a, const SeqGeom& sg, void* stream) {
    constexpr size_t lds = seqn_lds_bytes<D, WPS, BF, P3>();
    auto kern = seqn_fwd_kernel<D, WPS, NS, BF, P3>;
    static unsigned long long attr_done = 0;          // per device (common.h lds_attr_once): the same scheme as the backward's launcher
    if (int rc = lds_attr_once((const void*)kern, lds, attr_done)) return rc;
    const int grid = sg.live != nullptr ? sg.B : 2 * sg.B;
    kern<<<grid, 64 * WPS * NS, lds, (hipStream_t)stream>>>(a, sg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}

template <int WPS, int NS>
static int seqn_launch(const SeqFwdArgs& a, const SeqGeom& sg, void* stream, const HeadArgs* head = nullptr) {
    if (head != nullptr) {        // the head rides on the producer-side pieces build only
        if constexpr ((128 / 16 / NS) % 2 == 0) { if (a.w16 != nullptr && a.w16_planes == 3) return seqn_launch_px<128, WPS, NS>(a, sg, stream, head); }
        return AMID_ERR_UNSUPPORTED;
    }
    if (a.w16 != nullptr && a.w16_planes == 3) {
        // the producer-side pieces build (seqn_fwd_px_kernel) wherever a wave owns whole k-steps (an even number of column tiles);
        // SeqRing16x3's build -- every wave splits its strip's whole row itself -- for the one-tile parts of the diagnostic variant 18
        if constexpr ((128 / 16 / NS) % 2 == 0) return seqn_launch_px<128, WPS, NS>(a, sg, stream);
        else return seqn_launch_t<128, WPS, NS, true, true>(a, sg, stream);
    }
    return a.w16 != nullptr ? seqn_launch_t<128, WPS, NS, true>(a, sg, stream) : seqn_launch_t<128, WPS, NS, false>(a, sg, stream);
}

// variant: 0 = the default split for the shape; 42 / 22 / 24 / 14 / 18 = WPS, NS spelled out (diagnostics and tests).
// D = 64 (8 heads of 8 dims, two per column tile: the reference's default --emb_dim, train_sr.py:364): four column tiles, so two parts
// at four strips, four below; fp32 products only.
int launch_seqn_fwd(const SeqFwdArgs& a, const SeqGeom& sg, int D, int variant, void* stream, const HeadArgs* head) {
    const int T = sg.T;
    if (head != nullptr) {
        if (D != 128 || T <= 16 || T > 64 || (variant != 0 && variant != (T <= 32 ? 24 : 42))) return AMID_ERR_UNSUPPORTED;
        if (a.train && spec_bits(a.spec) != 1) return AMID_ERR_UNSUPPORTED;
        return T <= 32 ? seqn_launch<2, 4>(a, sg, stream, head) : seqn_launch<4, 2>(a, sg, stream, head);
    }
    if (a.train && spec_bits(a.spec) != 1) return AMID_ERR_UNSUPPORTED;      // part_dropout: the one-bit keep decisions of p = 0.5 (the reference's rate)
    const int wps = T <= 16 ? 1 : T <= 32 ? 2 : 4;
    if (D == 64) {
        if (a.w16 != nullptr || (variant != 0 && variant != 2)) return AMID_ERR_UNSUPPORTED;
        if (wps == 4) return seqn_launch_t<64, 4, 2, false>(a, sg, stream);
        if (wps == 2) return seqn_launch_t<64, 2, 4, false>(a, sg, stream);
        return seqn_launch_t<64, 1, 4, false>(a, sg, stream);
    }
    if (D != 128) return AMID_ERR_UNSUPPORTED;
    if (variant == 0) variant = wps == 4 ? 42 : wps == 2 ? 24 : 14;
    if (variant / 10 != wps) return AMID_ERR_UNSUPPORTED;
    switch (variant) {
        case 42: return seqn_launch<4, 2>(a, sg, stream);
        case 22: return seqn_launch<2, 2>(a, sg, stream);
        case 24: return seqn_launch<2, 4>(a, sg, stream);
        case 14: return seqn_launch<1, 4>(a, sg, stream);
        case 18: return seqn_launch<1, 8>(a, sg, stream);
        default: return AMID_ERR_UNSUPPORTED;
    }
}

}  // namespace amid

#ifdef AMID_STRIP_STAMPS
extern "C" int amid_seqn_stamps_read(unsigned long long* host) {       // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_seqn_stamp_buf), sizeof(unsigned long long) * 8 * 64);
}
#endif
