// The one-launch backward of the SASRec encoder, N-split build: the data gradients of a train step for the LIVE sequences, a workgroup
// per sequence as seq_bwd_kernel (sasrec_strip.hip), but with EIGHT waves -- two per 16-row strip, each owning D / 2 output columns of
// every product (seqn_parts.h, as the forward of sasrec_seqn.hip) -- and a wave per head in the attention core.  Same operations,
// operands, dropout counters, outputs and summation orders as seq_bwd_kernel: the two builds agree bit for bit
// (tests/test_gpu_timed_path.py::test_n_split_fused_backward_*).  Reference: autograd of Log2feats.forward (model_seq.py:371-383) under loss.backward(), train_sr.py:214.
//
// Why: in seq_bwd_kernel a SIMD holds ONE wave whose chain is 12 weight slabs x 256 MFMAs + two heads of the attention core per
// layer, and nothing covers its epilogues (LayerNorm backward, masks, stores), barriers and first-touch loads.  Here a SIMD holds two
// waves with 128 MFMAs per slab each -- one's epilogue under the other's matrix work -- and the attention core is one head per wave.
//   * a product needs the whole row of its operand: the operands made inside the chain (dpre2, dpre1, dy, dqn) are exchanged between
//     the strip's two waves through LDS ([strip][column tile][lane] float4); the ones that come from memory (dq, dk, dv) are loaded
//     whole by both.  Row sums of the LayerNorm backward are taken over the whole row by both waves (same order as the strip build);
//     element-wise work and stores cover the own columns.
//   Measured (profiles/tools/seqn_bwd_stamps.py, bench.py --set SEQ_BACKWARD=1 --set SEQ_BWD_VARIANT=1|2 on the diagnostic library): the products themselves run at
//   the matrix pipe's rate here (2.9 - 3.4 us per 64 x 128 x 128 slab pair against 6.5 in the strip build), but what surrounds them does not
//   shrink: the attention core's operand requests (24 per wave, 4.4 us for the eight waves of a CU), the barriers' skew, the first-use
//   round trips of dk / d_o.  In-kernel 112 - 116 us against 109; launch 124 - 128 against 124 - 126 us at cfg 2, 240.8 against 238.8 at
//   cfg 3: no gain, so "auto" stays the strip build and this one is the switchable, bit-identical alternative.
//   * LDS (fp32): ring 2 x 64 KB + exchange 32 KB = the whole CU.  While the exchange is idle (between the feed-forward chain's last
//     product and the q / k / v chain) it holds the attention core's per-wave transpose tiles and the LayerNorm partial sums of the four
//     strips: the partial sums of the q / k / v chain are carried in registers to that point of the NEXT layer (the last one has its own
//     barrier pair), so the chains pay no barrier for them.
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "attention_mfma.h"
#include "seq_fwd.h"
#include "seqn_parts.h"
#include "seq_bwd.h"
#include <type_traits>

namespace amid {

#ifdef AMID_STRIP_STAMPS
static __device__ unsigned long long amid_seqnb_stamp_buf[8 * 64];
#define SEQNB_STAMP(i) do { if (blockIdx.x == 0 && lane_id() == 0 && (i) < 64) amid_seqnb_stamp_buf[wave_id() * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SEQNB_STAMP(i) do { } while (0)
#endif

// the own column tiles of a whole row held in registers (two parts: selects on the wave-uniform part index; more parts would turn into
// scratch arrays -- those builds load their columns from memory instead, see the call sites)
template <int D, int NCT>
__device__ __forceinline__ void own_of(PartRegs<NCT>& o, const StripRegs<D>& full, int part) {
    static_assert(D / 16 == 2 * NCT, "two column parts");
#pragma unroll
    for (int c = 0; c < NCT; ++c) o.v[c] = part ? full.v[NCT + c] : full.v[c];
}

// row statistics of the LayerNorm backward over the WHOLE row, in strip_ln_bwd's order: mean / rstd of the input row, c1 = mean(gam dy),
// c2 = mean(gam dy xh)
template <int D>
__device__ __forceinline__ void ln_bwd_sums(const StripRegs<D>& dy, const StripRegs<D>& x, const ColVec<D>& gam, float eps, float& mean, float& rstd,
                                            float& c1, float& c2) {
    strip_stats<D>(x, eps, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (x.v[ct][r] - mean) * rstd;
            float gy = gam.v[ct][r] * dy.v[ct][r];
            asm volatile("" : "+v"(gy));               // the rounded product, as strip_ln_bwd
            s1 += gy;
            s2 = fmaf(gy, xh, s2);
        }
    }
    c1 = row_sum4(s1) * (1.0f / D);
    c2 = row_sum4(s2) * (1.0f / D);
}

// column sums over the strip's 16 rows of the own column tiles (4 NCT <= 16 per lane), compacted: lane (m, g), m < 4 NCT, keeps the sum of
// column (c0 + (m >> 2)) * 16 + 4 g + (m & 3) -- one register instead of up to sixteen for a tensor that waits for its LDS slot
template <int NCT>
__device__ __forceinline__ float col_sums_compact(const PartRegs<NCT>& x) {
    static_assert(NCT <= 4, "at most sixteen values over sixteen lanes");
    const int m = lane_id() & 15;
    float out = 0.f;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float sum = col_sum16(x.v[c][r]);
            out = (m == c * 4 + r) ? sum : out;
        }
    return out;
}
// -> scratch [strip][2][D] (the layout of ln_partials_wave)
template <int D, int NCT>
__device__ __forceinline__ void ln_partials_put(float* __restrict__ scratch, int si, int c0, float dgam, float dbet) {
    const int lane = lane_id(), m = lane & 15;
    if (m >= 4 * NCT) return;
    float* p = scratch + si * 2 * D + (c0 + (m >> 2)) * 16 + 4 * (lane >> 4) + (m & 3);
    lds_st1(p, dgam);
    lds_st1(p + D, dbet);
}
// the strips' partial sums of element e, in strip order
template <int D, int WPS>
__device__ __forceinline__ float ln_partials_sum(const float* __restrict__ S, int e) {
    if constexpr (WPS == 4) return (S[e] + S[2 * D + e]) + (S[4 * D + e] + S[6 * D + e]);      // (ln_partials_out's order)
    else if constexpr (WPS == 2) return S[e] + S[2 * D + e];
    else return S[e];
}

// the attention core's tile count for T rows in WPS strips, as a compile-time constant handed to f
template <int WPS, class F>
__device__ __forceinline__ void with_tiles(int T, const F& f) {
    if constexpr (WPS == 4) { if (T > 48) f(std::integral_constant<int, 4>()); else f(std::integral_constant<int, 3>()); }
    else if constexpr (WPS == 2) { if (T > 16) f(std::integral_constant<int, 2>()); else f(std::integral_constant<int, 1>()); }
    else f(std::integral_constant<int, 1>());
}

// what the feed-forward chain needs first: requested a product ahead by the caller
template <int D, int NCT> struct FfnPreN { PartRegs<NCT> Ho; unsigned tmw[NCT]; uint4 rr2; };

// floats of the exchange area: WPS strips x D / 16 tiles x 64 lanes x float4 -- or, while it is idle, two blocks of LayerNorm partial sums
// [WPS][2][D] and the eight waves' transpose tiles of the attention core
template <int D, int WPS> constexpr int seqn_bwd_exchange_floats() {
    constexpr int x = WPS * (D / 16) * 64 * 4, s = 2 * WPS * 2 * D + 8 * (ATTN_BWD_LDS_PER_WAVE / 4);
    return x > s ? x : s;
}

// WPS strips per sequence (T <= 16 WPS) x NS column parts = 8 waves: (4, 2) for 32 < T <= 64, (2, 4) for 16 < T <= 32, (1, 8) below
template <int D, int WPS, int NS, bool BF>
__global__ __launch_bounds__(512) void seqn_bwd_kernel(const SeqBwdArgs a, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = D / 16, NW = WPS * NS, NCT = NT / NS;
    static_assert((D == 128 || D == 64) && NW == 8 && NCT >= 1 && (!BF || D == 128), "eight waves: one per head in the attention core");
    constexpr bool PAIR = D == 64;                          // 8 heads of 8 dims: two per 16-column tile (attention_mfma.h), a wave per head
    const int bid = blockIdx.x;
    const int w = wave_id(), lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int part = w / WPS, si = w % WPS, c0 = part * NCT;
    const int B = sg.B, T = sg.T;
    // workgroup -> live sequence: the live list holds domain 0's batch rows, then domain 1's
    const int n0 = sg.live[B], n1 = B - n0;
    const int g = bid >= n0 ? 1 : 0;
    const int tl = bid - (g ? n0 : 0);
    const int top = a.n_layers - 1;
    using RingT = typename std::conditional<BF, SeqRing16<D, NW>, SeqRingN<D, NW>>::type;
    RingT ring(smem);
    auto W16 = [](const float* p) { return reinterpret_cast<const unsigned short*>(p); };
    if constexpr (BF) ring.first(W16(a.L[top].f.w2T[g])); else ring.first(a.L[top].f.w2T[g]);
    const int b = sg.live[bid];
    const int slot = g * B + tl;
    {   // the B slots without a live sequence hold zeros: workgroup j writes dead slot j (threads 0..255: the feed-forward LayerNorm's, 256..511: the other)
        const int dslot = bid < n1 ? n0 + bid : B + n1 + (bid - n1);
        const int e = threadIdx.x & (2 * D - 1);
        for (int l = 0; l <= top; ++l) {
            float* p = threadIdx.x < 2 * D ? a.L[l].f.ln_part : a.L[l].a.ln_part;
            p[(long long)dslot * 2 * D + e] = 0.f;
        }
    }
    float* const E = smem + (BF ? D * D : 2 * D * D);                          // the exchange: [strip][column tile][lane] float4
    float* const xb = E + si * (NT * 64 * 4);
    float* const S_f = E;                                                      // while the exchange is idle: LayerNorm partial sums [WPS][2][D] of
    float* const S_a = E + WPS * 2 * D;                                        // the feed-forward chain / of the layer above's q / k / v chain,
    float* const att_lds = E + 2 * WPS * 2 * D + w * (ATTN_BWD_LDS_PER_WAVE / 4);      // and this wave's transpose tiles of the attention core
    const int t = si * 16 + m;
    StripRow row;
    row.ok = t < T;
    row.local = b * T + min(t, T - 1);
    const unsigned phys = (unsigned)g * (unsigned)sg.M + (unsigned)(b * T + t);
    row.off = row.ok ? phys * (unsigned)(D * 4) + 16u * (unsigned)gq : STRIP_OOB;
    const unsigned off_own = row.ok ? row.off + (unsigned)c0 * 64u : STRIP_OOB;
    const unsigned tm_own = row.ok ? phys * (unsigned)(D / 4) + (unsigned)c0 * 4u : STRIP_OOB;
    const long long rowbase = (long long)g * sg.M + (long long)b * T;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.L[top].f.train) { seed = a.L[top].f.st->seed; step = (unsigned)a.L[top].f.st->step; }

    auto ffn_prefetch = [&](FfnPreN<D, NCT>& p, const StripFfnBwdArgs& f) {
        part_load<NCT>(p.Ho, GBuf(f.h, sg.act_bytes), off_own);
        if (f.tmq != nullptr) {
            const GBuf gtm(f.tmq, sg.tm_bytes);
#pragma unroll
            for (int c = 0; c < NCT; ++c) p.tmw[c] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(gtm.r, (int)(tm_own + 4 * c), 0, 0);
        }
        p.rr2 = make_uint4(0, 0, 0, 0);
        if (f.train) p.rr2 = rng_call(seed, (unsigned long long)row.local * D >> 7, site_id(g, f.layer, SITE_FFN2), step);
    };

    SEQNB_STAMP(0);
    PartRegs<NCT> DZo;                                      // d x' of the layer in hand, own columns
    float dgam_a = 0.f, dbet_a = 0.f;                       // LayerNorm-1 partial sums of the layer above (compact; carried to this layer's idle point)
    FfnPreN<D, NCT> pre;
    part_load<NCT>(DZo, GBuf(a.L[top].f.dxo, sg.act_bytes), off_own);
    ffn_prefetch(pre, a.L[top].f);
    f32x4 acc[NCT];
    StripRegs<D> F;                                         // the whole-row operand of the next product
#pragma unroll 1
    for (int l = top; l >= 0; --l) {
        const SeqBwdLayer& P = a.L[l];
        const SeqBwdLayer& Pn = a.L[l > 0 ? l - 1 : 0];
        const StripFfnBwdArgs& f = P.f;
        const StripQkvBwdArgs& q = P.a;
        [[maybe_unused]] const int sb = 1 + 8 * (top - l);
        AttnBwdOps oa;
        float dgam_f, dbet_f;
        // ---------------------------------------------------------------- feed-forward / out-projection chain: DZo -> dpre2, dpre1, dr, d_o
        {
            const GBuf gp2(f.dpre2, sg.act_bytes), gp1(f.dpre1, sg.act_bytes), gdr(f.dr, sg.act_bytes), gdo(f.d_o, sg.act_bytes);
            if (f.tmq != nullptr) {                         // dz = dx' * ~tm
                const int sh = 8 * gq;
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    const unsigned bits = pre.tmw[c] >> sh;
                    DZo.v[c][0] = (bits & 1u) ? 0.f : DZo.v[c][0];
                    DZo.v[c][1] = (bits & 2u) ? 0.f : DZo.v[c][1];
                    DZo.v[c][2] = (bits & 4u) ? 0.f : DZo.v[c][2];
                    DZo.v[c][3] = (bits & 8u) ? 0.f : DZo.v[c][3];
                }
            }
            PartRegs<NCT> Po = DZo;                         // dpre2 = dz * drop2
            if (f.train) part_dropout<NCT>(Po, pre.rr2, c0, f.spec, f.scale, (row.local * D) & 127);
            lds_barrier();                                  // the partner has read this wave's slots of the previous exchange
            xchg_write<NCT>(xb, c0, Po);
            StripRegs<D> Rs;
            ColVec<D> gam2;                                 // LayerNorm-2 gain: requested under the dy product, a product ahead of its use
            PartRegs<NCT> Ro, Go;                           // own columns of r and of the LayerNorm gain
            {   // dh = dpre2 C2 ; dpre1 = dh * relu'(h) * drop1   (h > 0 implies the unit was kept by drop1)
                const float* buf = ring.next();
                xchg_read<D>(F, xb);
                seqn_product<D, NCT, BF>(acc, F, buf, ring, f.w1T[g], W16(f.w1T[g]), c0,
                                         [&](int ct, int j) { part_spread<NCT>(gp2, off_own, Po, ct, j, 1); }, [&]() {
                    strip_load<D>(Rs, GBuf(f.r, sg.act_bytes), row);                 // LN2 input rows: needed a slab from now
                    if constexpr (NS != 2) { part_load<NCT>(Ro, GBuf(f.r, sg.act_bytes), off_own); part_cols<NCT>(Go, f.ln_w[g], c0); }
                });
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Po.v[c][r] = pre.Ho.v[c][r] > 0.f ? acc[c][r] * f.scale : 0.f;
            }
            SEQNB_STAMP(sb);
            lds_barrier();
            xchg_write<NCT>(xb, c0, Po);
            {   // dy = dpre1 C1 + dz
                const float* buf = ring.next();
                xchg_read<D>(F, xb);
                seqn_product<D, NCT, BF>(acc, F, buf, ring, f.woT[g], W16(f.woT[g]), c0,
                                         [&](int ct, int j) { part_spread<NCT>(gp1, off_own, Po, ct, j, 1); }, [&]() { gam2.load(f.ln_w[g]); });
#pragma unroll
                for (int c = 0; c < NCT; ++c) Po.v[c] = acc[c] + DZo.v[c];
            }
            SEQNB_STAMP(sb + 1);
            lds_barrier();
            xchg_write<NCT>(xb, c0, Po);
            {   // dr = LN2'(dy ; r) ; d_o = dr Wo
                const float* buf = ring.next();
                xchg_read<D>(F, xb);                        // the whole row of dy
                if (l == 0) SEQNB_STAMP(32);
                float mean, rstd, c1, c2;
                ln_bwd_sums<D>(F, Rs, gam2, f.ln_eps, mean, rstd, c1, c2);
                if constexpr (NS == 2) { own_of<D, NCT>(Ro, Rs, part); own_cols<D, NCT>(Go, gam2, f.ln_w[g], part, c0); }
                dbet_f = col_sums_compact<NCT>(Po);
                {
                    PartRegs<NCT> Dg;
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float xh = (Ro.v[c][r] - mean) * rstd;
                            float dg = Po.v[c][r] * xh;
                            asm volatile("" : "+v"(dg));       // (rounded, as strip_ln_bwd)
                            Dg.v[c][r] = dg;
                            float gy = Go.v[c][r] * Po.v[c][r];
                            asm volatile("" : "+v"(gy));
                            Po.v[c][r] = rstd * (gy - c1 - xh * c2);      // dr, own columns (stored under the product)
                        }
                    dgam_f = col_sums_compact<NCT>(Dg);
                }
#pragma unroll
                for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (Rs.v[ct][r] - mean) * rstd;
                        float gy = gam2.v[ct][r] * F.v[ct][r];
                        asm volatile("" : "+v"(gy));          // the rounded product, as strip_ln_bwd keeps it (not contracted into the subtraction)
                        F.v[ct][r] = rstd * (gy - c1 - xh * c2);
                    }
                if (l == 0) SEQNB_STAMP(33);
                // the attention core's saved operands of this wave's head: they come from HBM -- requested under this product.  (They hold 72
                // registers through the matrix loop: with the core's own ~220 the (4, 2) build spills 75 of them, 300 B per lane of scratch.
                // Requested BEHIND the product instead the build spills 23 (fp32) / 3 (bf16) -- measured, round 4: cfg 2 0.3761 ms per step
                // against 0.3763 with the five strip launches, a tie as before; cfg 3 0.658 against 0.665 for the strip-build fused kernel --
                // but the core's contractions then round differently from the strip build's (1 ulp on dq / dk / dv), and the bit-identity
                // of the two builds is worth more than 1 % at one shape: not adopted; "auto" never takes this build at T > 32.)
                with_tiles<WPS>(T, [&](auto nt) { attn_bwd_load_saved<decltype(nt)::value, PAIR>(oa, P.at, g, b, rowbase, PAIR ? w >> 1 : w); });
                if (l == 0) SEQNB_STAMP(34);
                seqn_product<D, NCT, BF>(acc, F, buf, ring, q.wkT[g], W16(q.wkT[g]), c0,
                                         [&](int ct, int j) { part_spread<NCT>(gdr, off_own, Po, ct, j, 1); });
#pragma unroll
                for (int c = 0; c < NCT; ++c) gdo.store4(off_own + c * 64, acc[c]);
            }
        }
        SEQNB_STAMP(sb + 2);
        w_ring_wait();                  // d_o has reached L2 (the Wk slab has landed as well)
        __syncthreads();                // ... and nobody reads the exchange any more
        if (l == 0) SEQNB_STAMP(35);
        ln_partials_put<D, NCT>(S_f, si, c0, dgam_f, dbet_f);
        if (l < top) ln_partials_put<D, NCT>(S_a, si, c0, dgam_a, dbet_a);
        if (l == 0) SEQNB_STAMP(36);
        // ---------------------------------------------------------------- attention core: head w
        with_tiles<WPS>(T, [&](auto nt) {                   // (D = 64: wave w runs head w & 1 of tile w >> 1)
            attn_bwd_load_dout<decltype(nt)::value>(oa, P.at, rowbase, PAIR ? w >> 1 : w);
            attn_bwd_compute<decltype(nt)::value, PAIR>(oa, P.at, rowbase, PAIR ? w >> 1 : w, att_lds, PAIR ? (w & 1) : -1);
        });
        SEQNB_STAMP(sb + 3);
        w_ring_wait();                  // dq / dk / dv have reached L2
        if (l == 0) SEQNB_STAMP(37);
        __syncthreads();
        strip_load<D>(F, GBuf(q.dk, sg.act_bytes), row);      // the next chain's first operand: its round trip under the sums below
        {   // the strips' partial sums -> this sequence's slot (fixed order)
            const int e = threadIdx.x & (2 * D - 1);
            if (threadIdx.x < 2 * D) f.ln_part[(long long)slot * 2 * D + e] = ln_partials_sum<D, WPS>(S_f, e);
            else if (l < top) a.L[l + 1].a.ln_part[(long long)slot * 2 * D + e] = ln_partials_sum<D, WPS>(S_a, e);
        }
        SEQNB_STAMP(sb + 4);
        // ---------------------------------------------------------------- q / k / v + LayerNorm-1 chain: dq, dk, dv, dr -> d x (own columns)
        {
            // (every operand is requested a product ahead of its use.  Tried: one request per slot of the MFMA loop in between instead of
            // eight at once, the attention core's 24 included -- the same microseconds move into the loop: 240.8 -> 252.3 us at cfg 3)
            StripRegs<D> A2, Xs;
            ColVec<D> gam;
            PartRegs<NCT> Dro, Xo, Go;
            f32x4 acc_kv[NCT];
            {   // dk Wk
                const float* buf = ring.next();
                seqn_product<D, NCT, BF>(acc_kv, F, buf, ring, q.wvT[g], W16(q.wvT[g]), c0, [](int, int) {},
                                         [&]() { strip_load<D>(A2, GBuf(q.dv, sg.act_bytes), row); });
            }
            SEQNB_STAMP(sb + 5);
            {   // + dv Wv
                const float* buf = ring.next();
                seqn_product<D, NCT, BF, false>(acc_kv, A2, buf, ring, q.wqT[g], W16(q.wqT[g]), c0, [](int, int) {}, [&]() {
                    strip_load<D>(F, GBuf(q.dq, sg.act_bytes), row);
                    strip_load<D>(Xs, GBuf(q.x, sg.act_bytes), row);                 // LN1 input rows
                    if constexpr (NS != 2) part_load<NCT>(Xo, GBuf(q.x, sg.act_bytes), off_own);
                });
            }
            SEQNB_STAMP(sb + 6);
            {   // dqn = dq Wq + dr ; dx = LN1'(dqn ; x) + (dk Wk + dv Wv)
                const float* buf = ring.next();
                part_load<NCT>(Dro, GBuf(q.dr, sg.act_bytes), off_own);              // residual-path gradient of the normed query
                gam.load(q.ln_w[g]);
                if constexpr (NS != 2) part_cols<NCT>(Go, q.ln_w[g], c0);
                if (l == 0) SEQNB_STAMP(38);
                seqn_product<D, NCT, BF>(acc, F, buf, ring, Pn.f.w2T[g], W16(Pn.f.w2T[g]), c0, [](int, int) {},
                                         [&]() { ffn_prefetch(pre, Pn.f); });        // (layer 0 "prefetches" its own: harmless, dead afterwards)
                PartRegs<NCT> Qo;
#pragma unroll
                for (int c = 0; c < NCT; ++c) Qo.v[c] = Dro.v[c] + acc[c];
                if (l == 0) SEQNB_STAMP(39);
                xchg_write<NCT>(xb, c0, Qo);               // (the exchange's last readers sit behind the attention core's barriers)
                lds_barrier();
                xchg_read<D>(F, xb);                        // the whole row of dqn
                if (l == 0) SEQNB_STAMP(40);
                float mean, rstd, c1, c2;
                ln_bwd_sums<D>(F, Xs, gam, q.ln_eps, mean, rstd, c1, c2);
                if constexpr (NS == 2) { own_of<D, NCT>(Xo, Xs, part); own_cols<D, NCT>(Go, gam, q.ln_w[g], part, c0); }
                dbet_a = col_sums_compact<NCT>(Qo);
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float xh = (Xo.v[c][r] - mean) * rstd;
                        float gy = Go.v[c][r] * Qo.v[c][r];
                        asm volatile("" : "+v"(gy));
                        float dg = Qo.v[c][r] * xh;
                        asm volatile("" : "+v"(dg));
                        Xo.v[c][r] = dg;
                        DZo.v[c][r] = rstd * (gy - c1 - xh * c2);
                    }
                dgam_a = col_sums_compact<NCT>(Xo);
#pragma unroll
                for (int c = 0; c < NCT; ++c) DZo.v[c] += acc_kv[c];
            }
            if (l == 0) part_store<NCT>(GBuf(q.dx, sg.act_bytes), off_own, DZo);
        }
        SEQNB_STAMP(sb + 7);
    }
    // the last chain's LayerNorm partial sums
    __syncthreads();                    // every wave has read its row of dqn
    ln_partials_put<D, NCT>(S_a, si, c0, dgam_a, dbet_a);
    __syncthreads();
    if (threadIdx.x < 2 * D) {
        const int e = threadIdx.x;
        a.L[0].a.ln_part[(long long)slot * 2 * D + e] = ln_partials_sum<D, WPS>(S_a, e);
    }
    SEQNB_STAMP(63);
    w_ring_wait();                      // the last (redundant) weight fetch targets this workgroup's LDS
}

template <int D, int WPS, bool BF> static constexpr size_t seqn_bwd_lds_bytes() {
    return (size_t)((BF ? D * D : 2 * D * D) + seqn_bwd_exchange_floats<D, WPS>()) * sizeof(float);
}

template <int D, int WPS, int NS, bool BF>
static int seqn_bwd_launch_t(const SeqBwdArgs& a, const StripGeom& sg, void* stream) {
    constexpr size_t lds = seqn_bwd_lds_bytes<D, WPS, BF>();
    auto kern = seqn_bwd_kernel<D, WPS, NS, BF>;
    static unsigned long long attr_done = 0;          // per device, as the strip launches (common.h lds_attr_once)
    if (int rc = lds_attr_once((const void*)kern, lds, attr_done)) return rc;
    kern<<<sg.B, 512, lds, (hipStream_t)stream>>>(a, sg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}

// D = 64 (8 heads of 8 dims): four column tiles -- two parts at four strips, four at two, eight waves either way; fp32 products only
int launch_seqn_bwd(const SeqBwdArgs& a, const StripGeom& sg, int D, int mma_bf16, void* stream) {
    const StripFfnBwdArgs& f = a.L[a.n_layers - 1].f;
    if (f.train && spec_bits(f.spec) != 1) return AMID_ERR_UNSUPPORTED;      // part_dropout: the one-bit keep decisions of p = 0.5 (the reference's rate)
    if (sg.T > 64) return AMID_ERR_UNSUPPORTED;
    if (D == 64) {
        if (mma_bf16) return AMID_ERR_UNSUPPORTED;
        if (sg.T > 32) return seqn_bwd_launch_t<64, 4, 2, false>(a, sg, stream);
        return seqn_bwd_launch_t<64, 2, 4, false>(a, sg, stream);            // (T <= 16 too: one strip stays empty)
    }
    if (D != 128) return AMID_ERR_UNSUPPORTED;
#ifdef AMID_DIAG_VARIANTS     // four strips at D 128: measured level with the strip build and over the register budget --
    if (sg.T > 32) return mma_bf16 ? seqn_bwd_launch_t<128, 4, 2, true>(a, sg, stream) : seqn_bwd_launch_t<128, 4, 2, false>(a, sg, stream);
#else                         // built into the diagnostic library only (profiles/tools/build_diag.sh); the product runs seq_bwd_kernel there
    if (sg.T > 32) return AMID_ERR_UNSUPPORTED;
#endif
    if (sg.T > 16) return mma_bf16 ? seqn_bwd_launch_t<128, 2, 4, true>(a, sg, stream) : seqn_bwd_launch_t<128, 2, 4, false>(a, sg, stream);
    return mma_bf16 ? seqn_bwd_launch_t<128, 1, 8, true>(a, sg, stream) : seqn_bwd_launch_t<128, 1, 8, false>(a, sg, stream);
}

}  // namespace amid

#ifdef AMID_STRIP_STAMPS
extern "C" int amid_seqnb_stamps_read(unsigned long long* host) {      // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_seqnb_stamp_buf), sizeof(unsigned long long) * 8 * 64);
}
#endif
