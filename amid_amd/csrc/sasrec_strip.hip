// SASRec encoder layer as register-resident strip GEMM chains (strip_gemm.h), forward and backward.
// Reference arithmetic: Log2feats.forward model_seq.py:371-383, nn.MultiheadAttention's projections as called at :374,
// PointWiseFeedForward model_seq.py:322-326, and their autograd (loss.backward(), train_sr.py:214).  Same operations, operands and
// saved tensors as the row-tile kernels of sasrec_fwd.hip / sasrec_bwd.hip (which stay for bf16 operands and for callers outside
// the fused train step); the attention core between the projections runs in its own launch (attention_mfma.hip).
//
//   strip_qkv_fwd        Qn = LN1(x) ; k = x Wk^T + bk ; v = x Wv^T + bv ; q = Qn Wq^T + bq          (k, v from the UN-normed x, :374)
//   strip_oproj_ffn_fwd  r = Qn + (o Wo^T + bo) (:378) ; y = LN2(r) ; h = relu(drop1(y C1^T + c1)) ; x' = (drop2(h C2^T + c2) + y) * ~tm
//                        [+ the next layer's strip_qkv_fwd on x' without leaving the registers]
//   strip_ffn_bwd        dz = dx' * ~tm ; dpre2 = dz * drop2 ; dpre1 = (dpre2 C2) * relu'(h) * drop1 ; dy = dpre1 C1 + dz ;
//                        dr = LN2'(dy ; r) ; d_o = dr Wo                                  (+ partial sums of d gamma2 / d beta2)
//   strip_qkv_bwd        dx = LN1'(dq Wq + dr ; x) + dk Wk + dv Wv                        (+ partial sums of d gamma1 / d beta1)
//                        [+ the layer below's strip_ffn_bwd on dx without leaving the registers]
// Backward weights arrive TRANSPOSED (wT[in][out], refreshed once per step by the head kernel) so that a data gradient is again
// C[rows, N] = A[rows, K] W'[N, K]^T.  Algorithmic FLOPs per row: 2 D D per projection; MFMA-bound (exact fp32 MFMA).
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "strip_chain.h"
#include "sort_phases.h"
#include "attention_mfma.h"
#include "seq_bwd.h"
#include "scorer_sum.h"
#include <type_traits>

namespace amid {

struct StripQkvArgs {
    const float* x;                     // [2M, D] layer input
    const float* ln_w[2]; const float* ln_b[2];
    const float* w_in[2]; const float* b_in[2];     // [3D, D], [3D]
    float* qn; float* q; float* k; float* v;
    float ln_eps;
};

struct StripOffArgs {
    const float* o; const float* qn;
    const float* w_o[2]; const float* b_o[2]; const float* ln_w[2]; const float* ln_b[2];
    const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
    const unsigned char* tmq;
    float* r; float* y; float* h; float* xo;
    float ln_eps;
    const StepState* st; int train; unsigned spec; float scale; int layer;
};

// ================================================================================================================ forward
// q / k / v of one layer on the strip X (in registers).  The ring's current fetch must be Wk of this layer (started by the caller).
// XSTORE: X is also written to a.x (a fused predecessor produced it: the saved layer input).
template <int D, bool XSTORE>
__device__ __forceinline__ void qkv_fwd_chain(const StripQkvArgs& a, const StripGeom& sg, Ring<D>& ring, const StripRow& row, int g,
                                              const StripRegs<D>& X, const ColVec<D>& lw, const ColVec<D>& lb) {
    constexpr int NT = D / 16;
    const GBuf gx(a.x, sg.act_bytes), gqn(a.qn, sg.act_bytes), gq(a.q, sg.act_bytes), gk(a.k, sg.act_bytes), gv(a.v, sg.act_bytes);
    StripRegs<D> Qn, Kr, Vr;
    ColVec<D> bias;
    strip_layernorm<D>(Qn, X, lw, lb, a.ln_eps);
    STRIP_STAMP(3);
    f32x4 acc[NT];
    {   // k = x Wk^T + bk ; x's and Qn's global copies leave under these MFMAs
        const float* buf = ring.next();
        STRIP_STAMP(4);
        bias.load(a.b_in[g] + D);
        strip_zero<D>(acc);
        strip_mma<D>(acc, X, buf, [&](int ct, int j) {
            ring.fetch(a.w_in[g] + 2LL * D * D, ct, j);
            if constexpr (XSTORE) store_spread<D>(gx, row, X, ct, j);
            store_spread<D>(gqn, row, Qn, ct, j);
        });
        STRIP_STAMP(5);
        add_bias<D>(acc, bias);
        to_regs<D>(Kr, acc);
        STRIP_STAMP(6);
    }
    {   // v = x Wv^T + bv
        const float* buf = ring.next();
        STRIP_STAMP(7);
        bias.load(a.b_in[g] + 2 * D);
        strip_zero<D>(acc);
        strip_mma<D>(acc, X, buf, [&](int ct, int j) { ring.fetch(a.w_in[g], ct, j); store_spread<D>(gk, row, Kr, ct, j); });
        STRIP_STAMP(8);
        add_bias<D>(acc, bias);
        to_regs<D>(Vr, acc);
    }
    {   // q = Qn Wq^T + bq
        const float* buf = ring.next();
        STRIP_STAMP(9);
        bias.load(a.b_in[g]);
        strip_zero<D>(acc);
        strip_mma<D>(acc, Qn, buf, [&](int ct, int j) { store_spread<D>(gv, row, Vr, ct, j); });
        STRIP_STAMP(10);
        add_bias<D>(acc, bias);
        to_regs<D>(Kr, acc);
        strip_store<D>(gq, row, Kr);
        STRIP_STAMP(11);
    }
}

template <int D>
__global__ __launch_bounds__(STRIP_THREADS) void strip_qkv_fwd_kernel(const StripQkvArgs a, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    STRIP_STAMP(0);
    Ring<D> ring(smem);
    ring.first(a.w_in[strip_domain(blockIdx.x)] + 1LL * D * D);
    const StripTile t = strip_tile(sg, blockIdx.x);
    if (!t.live) { w_ring_wait(); return; }
    const StripRow row = strip_row<D>(sg, t);
    STRIP_STAMP(1);
    StripRegs<D> X;
    ColVec<D> lw, lb;
    strip_load<D>(X, GBuf(a.x, sg.act_bytes), row);
    lw.load(a.ln_w[t.g]); lb.load(a.ln_b[t.g]);
    STRIP_STAMP(2);
    qkv_fwd_chain<D, false>(a, sg, ring, row, t.g, X, lw, lb);
}

template <int D, bool NEXT>
__global__ __launch_bounds__(STRIP_THREADS) void strip_oproj_ffn_fwd_kernel(const StripOffArgs a, const StripQkvArgs nx, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = D / 16;
    Ring<D> ring(smem);
    ring.first(a.w_o[strip_domain(blockIdx.x)]);
    const StripTile t = strip_tile(sg, blockIdx.x);
    if (!t.live) { w_ring_wait(); return; }
    const StripRow row = strip_row<D>(sg, t);
    const int g = t.g;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    const GBuf gr(a.r, sg.act_bytes), gy(a.y, sg.act_bytes), gh(a.h, sg.act_bytes), gxo(a.xo, sg.act_bytes);
    StripRegs<D> A, R, Y, H;
    StripTm<D> tm;
    ColVec<D> bias, lw, lb;
    strip_load<D>(A, GBuf(a.o, sg.act_bytes), row);
    strip_load<D>(R, GBuf(a.qn, sg.act_bytes), row);                   // the residual: the NORMED query (model_seq.py:378)
    const bool has_tm = a.tmq != nullptr;
    if (has_tm) strip_tm_load<D>(tm, GBuf(a.tmq, sg.tm_bytes), row);
    bias.load(a.b_o[g]); lw.load(a.ln_w[g]); lb.load(a.ln_b[g]);
    f32x4 acc[NT];
    {   // r = Qn + (o Wo^T + bo) ; y = LN2(r)
        const float* buf = ring.next();
        strip_zero<D>(acc);
        strip_mma<D>(acc, A, buf, [&](int ct, int j) { ring.fetch(a.w1[g], ct, j); });
        add_bias<D>(acc, bias);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) R.v[ct] += acc[ct];
        strip_layernorm<D>(Y, R, lw, lb, a.ln_eps);
    }
    {   // h = relu(drop1(y C1^T + c1))
        const float* buf = ring.next();
        bias.load(a.b1[g]);
        strip_zero<D>(acc);
        strip_mma<D>(acc, Y, buf, [&](int ct, int j) { ring.fetch(a.w2[g], ct, j); store_spread<D>(gr, row, R, ct, j); });
        add_bias<D>(acc, bias);
        to_regs<D>(H, acc);
        if (a.train) strip_dropout<D>(H, seed, site_id(g, a.layer, SITE_FFN1), step, row.local, a.spec, a.scale);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) H.v[ct][r] = fmaxf(H.v[ct][r], 0.f);
    }
    {   // x' = (drop2(h C2^T + c2) + y) * ~tm
        const float* buf = ring.next();
        bias.load(a.b2[g]);
        if constexpr (NEXT) { lw.load(nx.ln_w[g]); lb.load(nx.ln_b[g]); }
        strip_zero<D>(acc);
        strip_mma<D>(acc, H, buf, [&](int ct, int j) {
            if constexpr (NEXT) ring.fetch(nx.w_in[g] + 1LL * D * D, ct, j);
            store_spread<D>(gy, row, Y, ct, j);
            if (ct < NT / 2 && (j & 3) == 3) strip_store_ct<D>(gh, row, H, 2 * ct + (j >> 2));
        });
        add_bias<D>(acc, bias);
        to_regs<D>(A, acc);
        if (a.train) strip_dropout<D>(A, seed, site_id(g, a.layer, SITE_FFN2), step, row.local, a.spec, a.scale);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) A.v[ct] += Y.v[ct];
        if (has_tm) strip_apply_tm<D>(A, tm);
    }
    if constexpr (NEXT) {
        qkv_fwd_chain<D, true>(nx, sg, ring, row, g, A, lw, lb);
    } else {
        strip_store<D>(gxo, row, A);
    }
}

// ================================================================================================================ backward
// LayerNorm backward of the strip: dx = LN'(dy ; x, gamma); this lane's row adds dy * xhat / dy to the column partials
template <int D>
__device__ __forceinline__ void strip_ln_bwd(StripRegs<D>& dx, const StripRegs<D>& dy, const StripRegs<D>& x, const ColVec<D>& gam,
                                             float eps, StripRegs<D>& dgam, StripRegs<D>& dbet) {
    float mean, rstd;
    strip_stats<D>(x, eps, mean, rstd);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (x.v[ct][r] - mean) * rstd;
            float gy = gam.v[ct][r] * dy.v[ct][r];
            asm volatile("" : "+v"(gy));               // the ROUNDED product everywhere below (never contracted into a sum: the N-split build
            s1 += gy;                                  // of the fused backward, sasrec_seqn_bwd.hip, reproduces these bits)
            s2 = fmaf(gy, xh, s2);
            float dg = dy.v[ct][r] * xh;               // first (and only) row of this lane
            asm volatile("" : "+v"(dg));               // (rounded: not contracted into the column sums' first addition)
            dgam.v[ct][r] = dg;
            dbet.v[ct][r] = dy.v[ct][r];
            dx.v[ct][r] = gy;                          // finished below
        }
    }
    const float c1 = row_sum4(s1) * (1.0f / D), c2 = row_sum4(s2) * (1.0f / D);
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (x.v[ct][r] - mean) * rstd;
            dx.v[ct][r] = rstd * (dx.v[ct][r] - c1 - xh * c2);
        }
}

// the loads the feed-forward backward needs first (relu output, "== 0" bits, LayerNorm gain): issued by the caller a slab ahead
template <int D> struct FfnBwdPre { StripRegs<D> Hs; StripTm<D> tm; ColVec<D> gam; };
template <int D>
__device__ __forceinline__ void ffn_bwd_prefetch(FfnBwdPre<D>& p, const StripFfnBwdArgs& a, const StripGeom& sg, const StripRow& row, int g) {
    strip_load<D>(p.Hs, GBuf(a.h, sg.act_bytes), row);
    if (a.tmq != nullptr) strip_tm_load<D>(p.tm, GBuf(a.tmq, sg.tm_bytes), row);
    p.gam.load(a.ln_w[g]);
}

// d x' (DZ, in registers) -> dpre2, dpre1, dr, d_o of this layer; the ring's current fetch must be w2T.  TAIL: a slab (`tail`) is
// fetched under the last MFMA loop (a fused successor's first weight)
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int D, bool TAIL = false, class Hook = NoHook, class RingT = Ring<D>>
__device__ __forceinline__ void ffn_bwd_chain(const StripFfnBwdArgs& a, const StripGeom& sg, RingT& ring, const StripRow& row, int g,
                                              StripRegs<D>& DZ, FfnBwdPre<D>& pre, float* __restrict__ scratch,
                                              const float* __restrict__ tail = nullptr, const Hook& before_last = NoHook()) {
    constexpr int NT = D / 16;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    const GBuf gp2(a.dpre2, sg.act_bytes), gp1(a.dpre1, sg.act_bytes), gdr(a.dr, sg.act_bytes), gdo(a.d_o, sg.act_bytes);
    StripRegs<D> P, Rs;
    StripRegs<D>& Hs = pre.Hs;
    if (a.tmq != nullptr) strip_apply_tm<D>(DZ, pre.tm);
    // dpre2 = dz * drop2
    P = DZ;
    if (a.train) strip_dropout<D>(P, seed, site_id(g, a.layer, SITE_FFN2), step, row.local, a.spec, a.scale);
    f32x4 acc[NT];
    {   // dh = dpre2 C2 ; dpre1 = dh * relu'(h) * drop1   (h > 0 implies the unit was kept by drop1)
        STRIP_STAMP(18);
        const float* buf = ring.next();
        STRIP_STAMP(19);
        strip_load<D>(Rs, GBuf(a.r, sg.act_bytes), row);               // LN2 input rows: needed two slabs from now
        strip_zero<D>(acc);
        strip_product<D>(acc, P, buf, ring, [&](int ct, int j) { ring.fetch(a.w1T[g], ct, j); store_spread<D>(gp2, row, P, ct, j); });
        STRIP_STAMP(20);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) P.v[ct][r] = Hs.v[ct][r] > 0.f ? acc[ct][r] * a.scale : 0.f;
    }
    StripRegs<D> DR, dgam, dbet;
    {   // dy = dpre1 C1 + dz ; dr = LN2'(dy ; r)
        const float* buf = ring.next();
        STRIP_STAMP(21);
        strip_zero<D>(acc);
        strip_product<D>(acc, P, buf, ring, [&](int ct, int j) { ring.fetch(a.woT[g], ct, j); store_spread<D>(gp1, row, P, ct, j); });
        STRIP_STAMP(22);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) Hs.v[ct] = acc[ct] + DZ.v[ct];
        strip_ln_bwd<D>(DR, Hs, Rs, pre.gam, a.ln_eps, dgam, dbet);
        STRIP_STAMP(23);
    }
    {   // d_o = dr Wo
        const float* buf = ring.next();
        STRIP_STAMP(24);
        before_last();                  // (a fused successor requests operands here: they fly under this product)
        strip_zero<D>(acc);
        strip_product<D>(acc, DR, buf, ring, [&](int ct, int j) {
            if constexpr (TAIL) ring.fetch(tail, ct, j);
            store_spread<D>(gdr, row, DR, ct, j);
        });
        STRIP_STAMP(25);
        to_regs<D>(P, acc);
        strip_store<D>(gdo, row, P);
    }
    ln_partials_wave<D>(scratch, dgam, dbet);
    STRIP_STAMP(26);
}

// dq, dk, dv, dr of a layer -> d x (left in DX); the ring's current fetch must be wkT.  TAIL: a slab (`tail`) is fetched behind wqT;
// `before_last()` runs in front of the last MFMA loop (a fused successor issues its first loads there).
template <int D, bool TAIL, class Hook, class RingT>
__device__ __forceinline__ void qkv_bwd_chain(const StripQkvBwdArgs& a, const StripGeom& sg, RingT& ring, const StripRow& row, int g,
                                              StripRegs<D>& DX, float* __restrict__ scratch, const float* __restrict__ tail, const Hook& before_last) {
    constexpr int NT = D / 16;
    StripRegs<D> Dk, Dv, Dq, Drs, Xs;
    ColVec<D> gam;
    strip_load<D>(Dk, GBuf(a.dk, sg.act_bytes), row);
    strip_load<D>(Dv, GBuf(a.dv, sg.act_bytes), row);
    f32x4 acc_kv[NT], acc[NT];
    strip_zero<D>(acc_kv);
    {   // dk Wk          (every operand is requested one slab ahead of its use: the loads fly under the MFMAs in between)
        const float* buf = ring.next();
        strip_load<D>(Dq, GBuf(a.dq, sg.act_bytes), row);
        strip_product<D>(acc_kv, Dk, buf, ring, [&](int ct, int j) { ring.fetch(a.wvT[g], ct, j); });
    }
    {   // + dv Wv
        const float* buf = ring.next();
        strip_load<D>(Drs, GBuf(a.dr, sg.act_bytes), row);             // residual-path gradient of the normed query
        strip_load<D>(Xs, GBuf(a.x, sg.act_bytes), row);               // LN1 input rows
        gam.load(a.ln_w[g]);
        strip_product<D>(acc_kv, Dv, buf, ring, [&](int ct, int j) { ring.fetch(a.wqT[g], ct, j); });
    }
    StripRegs<D> dgam, dbet;
    {   // dqn = dq Wq + dr ; dx = LN1'(dqn ; x) + (dk Wk + dv Wv)
        const float* buf = ring.next();
        before_last();
        strip_zero<D>(acc);
        strip_product<D>(acc, Dq, buf, ring, [&](int ct, int j) { if constexpr (TAIL) ring.fetch(tail, ct, j); });
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) Drs.v[ct] += acc[ct];
        strip_ln_bwd<D>(DX, Drs, Xs, gam, a.ln_eps, dgam, dbet);
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) DX.v[ct] += acc_kv[ct];
    }
    ln_partials_wave<D>(scratch, dgam, dbet);
}

// RIDER: 0, or the phase of the step's index sort (sort_phases.h) that the first rd.plan.nblk workgroups run, on CUs the live tiles
// leave free (this launch: phase 2, the scatter of pass 0)
template <int D, int RIDER, int BF = 0>
__global__ __launch_bounds__(STRIP_THREADS) void strip_ffn_bwd_kernel(const StripFfnBwdArgs a, const StripGeom sg, const SortRider rd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    if constexpr (RIDER != 0) {
        if (bid < rd.plan.nblk) { sort_phase_ct<RIDER>(rd.plan, bid, smem); return; }
        bid -= rd.plan.nblk;
    }
    STRIP_STAMP(16);
    typename RingSel<D, BF>::type ring(smem);
    ring.first(a.w2T[strip_domain(bid)]);
    const StripTile t = strip_tile(sg, bid);
    if (!t.live) { zero_slot<D>(a.ln_part, t.slot); w_ring_wait(); return; }
    const StripRow row = strip_row<D>(sg, t);
    StripRegs<D> DZ;
    FfnBwdPre<D> pre;
    strip_load<D>(DZ, GBuf(a.dxo, sg.act_bytes), row);
    ffn_bwd_prefetch<D>(pre, a, sg, row, t.g);
    STRIP_STAMP(17);
    ffn_bwd_chain<D>(a, sg, ring, row, t.g, DZ, pre, ln_scratch<D>(smem, 0));
    __syncthreads();
    ln_partials_out<D>(ln_scratch<D>(smem, 0), a.ln_part + (long long)t.slot * 2 * D);
    STRIP_STAMP(27);
}

// FFN = true: the layer below's feed-forward / out-projection backward continues on d x in registers (d x is then never stored)
// RIDER: as strip_ffn_bwd_kernel (with FFN: phase 3, pass 1's counts; without: phase 4, the scatter of pass 1)
template <int D, bool FFN, int RIDER, int BF = 0>
__global__ __launch_bounds__(STRIP_THREADS) void strip_qkv_bwd_kernel(const StripQkvBwdArgs a, const StripFfnBwdArgs f, const ScorerSum ss,
                                                                      const StripGeom sg, const SortRider rd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int bid = blockIdx.x;
    if constexpr (RIDER != 0 && FFN) {
        // riders BEHIND the tiles (the last ss.nblk workgroups: they fill the CUs the dead tiles leave at once; in front they would take CUs
        // from live tiles): the scorer's weight gradients from the head's per-sample hidden gradients (scorer_sum.h)
        // (long batches: IN FRONT instead -- scorer_sum.h ScorerSum::front)
        if (ss.front) {
            if (bid < ss.nblk) { scorer_sum_block(ss, bid, (scorer_lds_f4*)smem); return; }
            bid -= ss.nblk;
        } else {
            const int first = (int)gridDim.x - ss.nblk;
            if (bid >= first) { scorer_sum_block(ss, bid - first, (scorer_lds_f4*)smem); return; }
        }
    }
    if constexpr (RIDER != 0) {
        if (bid < rd.plan.nblk) { sort_phase_ct<RIDER>(rd.plan, bid, smem); return; }
        bid -= rd.plan.nblk;
    }
    typename RingSel<D, BF>::type ring(smem);
    ring.first(a.wkT[strip_domain(bid)]);
    const StripTile t = strip_tile(sg, bid);
    if (!t.live) {
        zero_slot<D>(a.ln_part, t.slot);
        if constexpr (FFN) zero_slot<D>(f.ln_part, t.slot);
        w_ring_wait();
        return;
    }
    const StripRow row = strip_row<D>(sg, t);
    StripRegs<D> DX;
    if constexpr (FFN) {
        FfnBwdPre<D> pre;
        qkv_bwd_chain<D, true>(a, sg, ring, row, t.g, DX, ln_scratch<D>(smem, 0), f.w2T[t.g], [&]() { ffn_bwd_prefetch<D>(pre, f, sg, row, t.g); });
        ffn_bwd_chain<D>(f, sg, ring, row, t.g, DX, pre, ln_scratch<D>(smem, 1));
    } else {
        const bool emb = a.emb_tmq != nullptr;                 // (uniform) the embedding layer's backward on the strip: see StripQkvBwdArgs
        StripTm<D> etm;
        qkv_bwd_chain<D, false>(a, sg, ring, row, t.g, DX, ln_scratch<D>(smem, 0), nullptr,
                                [&]() { if (emb) strip_tm_load<D>(etm, GBuf(a.emb_tmq, sg.tm_bytes), row); });
        if (emb) {
            if (a.emb_train) strip_dropout<D>(DX, a.emb_st->seed, site_id(t.g, 0, SITE_EMB), (unsigned)a.emb_st->step, row.local, a.emb_spec, a.emb_scale);
            strip_apply_tm<D>(DX, etm);
        }
        strip_store<D>(GBuf(a.dx, sg.act_bytes), row, DX);
    }
    __syncthreads();
    ln_partials_out<D>(ln_scratch<D>(smem, 0), a.ln_part + (long long)t.slot * 2 * D);
    if constexpr (FFN) ln_partials_out<D>(ln_scratch<D>(smem, 1), f.ln_part + (long long)t.slot * 2 * D);
}


// ================================================================================================ the whole backward of a sequence
// ONE launch for the encoder's data gradients of a train step: per layer, top down, the feed-forward / out-projection chain, the
// attention core's backward (attention_mfma.h) and the q / k / v + LayerNorm-1 chain, for the LIVE sequences only -- the launches of
// strip_ffn_bwd, attn_bwd, strip_qkv_bwd [+ strip_ffn_bwd], attn_bwd, strip_qkv_bwd as one workgroup-long chain per sequence.
// A workgroup owns one sequence (T <= 64: four strips, as the fused forward's WPS = 4 tiling), so the attention core sees all of
// its rows: wave w computes heads w and w + 4.  The chains keep d x in registers from layer to layer; the attention core exchanges
// through global memory (d_o written by the chain, dq / dk / dv read by the next one -- the weight-gradient launch needs those
// copies anyway): a workgroup barrier behind a vmcnt(0) makes a workgroup's stores visible to its own loads (same CU, same L1).
// What it saves is every launch boundary's tail and head -- the last stores, the launch, the first weight slab, the first
// operand loads: the next chain's first slab lands under the attention core and its operands are L2 hits.
template <int D, bool BF = false>
__global__ __launch_bounds__(STRIP_THREADS) void seq_bwd_kernel(const SeqBwdArgs a, const StripGeom sg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int w = wave_id(), lane = lane_id(), m = lane & 15, gq = lane >> 4;
    const int B = sg.B, T = sg.T;
    // workgroup -> live sequence: the live list holds domain 0's batch rows, then domain 1's (contiguous ranges: see seq_fwd_kernel)
    const int n0 = sg.live[B], n1 = B - n0;
    const int g = bid >= n0 ? 1 : 0;
    const int tl = bid - (g ? n0 : 0);
    const int top = a.n_layers - 1;
    Ring<D, BF> ring(smem);
    ring.first(a.L[top].f.w2T[g]);
    const int b = sg.live[bid];
    const int slot = g * B + tl;
    {   // the B slots without a live sequence (domain 0: [n0, B), domain 1: [n1, B)) hold zeros: workgroup j writes dead slot j
        const int dslot = bid < n1 ? n0 + bid : B + n1 + (bid - n1);
        for (int l = 0; l <= top; ++l) { zero_slot<D>(a.L[l].f.ln_part, dslot); zero_slot<D>(a.L[l].a.ln_part, dslot); }
    }
    const int t = w * 16 + m;
    StripRow row;
    row.ok = t < T;
    row.local = b * T + min(t, T - 1);
    const unsigned phys = (unsigned)g * (unsigned)sg.M + (unsigned)(b * T + t);
    row.off = row.ok ? phys * (unsigned)(D * 4) + 16u * (unsigned)gq : STRIP_OOB;
    const long long rowbase = (long long)g * sg.M + (long long)b * T;
    float* att = smem + 2 * D * D + 16 * D;                                       // behind the ring and the two LayerNorm scratch blocks
    float* att_lds = att + w * (ATTN_BWD_LDS_PER_WAVE / 4);                       // this wave's scratch block (attention_mfma.h)

    STRIP_RSTAMP(0);
    STRIP_STAMP(14);
    StripRegs<D> DZ;
    FfnBwdPre<D> pre;
    strip_load<D>(DZ, GBuf(a.L[top].f.dxo, sg.act_bytes), row);
    ffn_bwd_prefetch<D>(pre, a.L[top].f, sg, row, g);
#pragma unroll 1
    for (int l = top; l >= 0; --l) {
        const SeqBwdLayer& P = a.L[l];
        const SeqBwdLayer& Pn = a.L[l > 0 ? l - 1 : 0];
        [[maybe_unused]] const int sb = 1 + 6 * (top - l);                // (diagnostic builds: real-time stamps of workgroup 0)
        // the first head's operands that the forward saved are requested under the chain's last product (they come from HBM: a wave that
        // asked for them behind the barrier would sit out the round trip), the second head's fly under the first head's arithmetic
        AttnBwdOps oa, ob;
        const bool nt4 = T > 48;
        ffn_bwd_chain<D, true>(P.f, sg, ring, row, g, DZ, pre, ln_scratch<D>(smem, 0), P.a.wkT[g], [&]() {
            if (nt4) attn_bwd_load_saved<4>(oa, P.at, g, b, rowbase, w); else attn_bwd_load_saved<3>(oa, P.at, g, b, rowbase, w);
        });
        STRIP_RSTAMP(sb);
        w_ring_wait();                  // d_o has reached L2 (the Wk slab has landed as well: it was requested half a GEMM ago)
        __syncthreads();
        ln_partials_out<D>(ln_scratch<D>(smem, 0), P.f.ln_part + (long long)slot * 2 * D);
        STRIP_RSTAMP(sb + 1);
        if (nt4) {
            attn_bwd_load_dout<4>(oa, P.at, rowbase, w);
            attn_bwd_load_saved<4>(ob, P.at, g, b, rowbase, w + STRIP_WAVES); attn_bwd_load_dout<4>(ob, P.at, rowbase, w + STRIP_WAVES);
            attn_bwd_compute<4>(oa, P.at, rowbase, w, att_lds);
            STRIP_RSTAMP(sb + 2);
            attn_bwd_compute<4>(ob, P.at, rowbase, w + STRIP_WAVES, att_lds);
        } else {
            attn_bwd_load_dout<3>(oa, P.at, rowbase, w);
            attn_bwd_load_saved<3>(ob, P.at, g, b, rowbase, w + STRIP_WAVES); attn_bwd_load_dout<3>(ob, P.at, rowbase, w + STRIP_WAVES);
            attn_bwd_compute<3>(oa, P.at, rowbase, w, att_lds);
            STRIP_RSTAMP(sb + 2);
            attn_bwd_compute<3>(ob, P.at, rowbase, w + STRIP_WAVES, att_lds);
        }
        STRIP_RSTAMP(sb + 3);
        w_ring_wait();                  // dq / dk / dv have reached L2
        __syncthreads();
        STRIP_RSTAMP(sb + 4);
        StripRegs<D> DX;
        // (layer 0 "prefetches" its own w2T into the free buffer: harmless, waited for at the end)
        qkv_bwd_chain<D, true>(P.a, sg, ring, row, g, DX, ln_scratch<D>(smem, 1), Pn.f.w2T[g],
                               [&]() { ffn_bwd_prefetch<D>(pre, Pn.f, sg, row, g); });      // (unconditional: `pre` is then dead across the attention core)
        STRIP_RSTAMP(sb + 5);
        __syncthreads();
        ln_partials_out<D>(ln_scratch<D>(smem, 1), P.a.ln_part + (long long)slot * 2 * D);
        if (l == 0) strip_store<D>(GBuf(P.a.dx, sg.act_bytes), row, DX);
        DZ = DX;
    }
    STRIP_RSTAMP(13);
    STRIP_STAMP(15);
    w_ring_wait();
}

}  // namespace amid

using namespace amid;
using namespace amid_strip_host;

static int make_rider(SortRider& rd, const void* sort_plan, int sort_phase) {
    rd.phase = 0;
    if (sort_plan == nullptr) return AMID_OK;
    if (sort_phase < 1 || sort_phase > 4) return AMID_ERR_ARG;
    rd.plan = *(const SortPlan*)sort_plan;
    rd.phase = sort_phase;
    return AMID_OK;
}

// a strip launch with a sort rider: rd.plan.nblk extra workgroups in front of the tiles'
// (extra: further rider workgroups behind the sort's -- the scorer sums of strip_qkv_bwd_kernel)
template <auto KERNEL, int DVAL, class... Args>
static int launch_strip_rider_x(const StripGeom& sg, const SortRider& rd, int extra, void* stream, const Args&... args) {
    static unsigned long long attr_done = 0;
    // (the rider workgroups run their sort phase on the head of the dynamic LDS: the 4 096-bin scatter needs 48 KB)
    if (rd.phase != 0 && strip_lds_bytes<DVAL>() < (rd.plan.g0.bits > 10 ? sizeof(SortScatterLds<OS_BINS_MAX>) : sizeof(SortScatterLds<1024>))) return AMID_ERR_UNSUPPORTED;
    if (int rc = lds_attr_once((const void*)KERNEL, strip_lds_bytes<DVAL>(), attr_done)) return rc;
    KERNEL<<<2 * sg.tpg + rider_blocks_host(rd) + extra, STRIP_THREADS, strip_lds_bytes<DVAL>(), (hipStream_t)stream>>>(args..., sg, rd);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}
template <auto KERNEL, int DVAL, class... Args>
static int launch_strip_rider(const StripGeom& sg, const SortRider& rd, void* stream, const Args&... args) {
    static unsigned long long attr_done = 0;
    // (the rider workgroups run their sort phase on the head of the dynamic LDS: the 4 096-bin scatter needs 48 KB)
    if (rd.phase != 0 && strip_lds_bytes<DVAL>() < (rd.plan.g0.bits > 10 ? sizeof(SortScatterLds<OS_BINS_MAX>) : sizeof(SortScatterLds<1024>))) return AMID_ERR_UNSUPPORTED;
    if (int rc = lds_attr_once((const void*)KERNEL, strip_lds_bytes<DVAL>(), attr_done)) return rc;
    KERNEL<<<2 * sg.tpg + rider_blocks_host(rd), STRIP_THREADS, strip_lds_bytes<DVAL>(), (hipStream_t)stream>>>(args..., sg, rd);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}

extern "C" int amid_sas_strip_tile_rows(void) { return STRIP_TILE; }

#ifdef AMID_STRIP_STAMPS
extern "C" int amid_strip_stamps_read(unsigned long long* host) {       // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_strip_stamp_buf), sizeof(unsigned long long) * STRIP_STAMP_WAVES * 32);
}
#endif

extern "C" int amid_sas_strip_qkv_fwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* const* w_in,
                                          const float* const* b_in, float ln_eps, int B, int T, int D, const int* live, float* qn, float* q,
                                          float* k, float* v, void* stream) {
    AMID_CHECK_ARG(x && ln_w && ln_b && w_in && b_in && qn && q && k && v);
    StripQkvArgs a;
    a.x = x; a.qn = qn; a.q = q; a.k = k; a.v = v; a.ln_eps = ln_eps;
    for (int g = 0; g < 2; ++g) { a.ln_w[g] = ln_w[g]; a.ln_b[g] = ln_b[g]; a.w_in[g] = w_in[g]; a.b_in[g] = b_in[g]; }
    StripGeom sg;
    if (int e = make_strip_geom(B, T, D, live, &sg)) return e;
    if (D == 128) return launch_strip<strip_qkv_fwd_kernel<128>, 128>(sg, stream, a);
    if (D == 64) return launch_strip<strip_qkv_fwd_kernel<64>, 64>(sg, stream, a);
    return AMID_ERR_UNSUPPORTED;
}

extern "C" int amid_sas_strip_oproj_ffn_fwd_f32(const float* o, const float* qn, const float* const* w_o, const float* const* b_o,
                                                const float* const* ln_w, const float* const* ln_b, const float* const* w1,
                                                const float* const* b1, const float* const* w2, const float* const* b2,
                                                const unsigned char* tmq, float ln_eps, int B, int T, int D, const int* live, int layer,
                                                const void* step_state, int train, float p_drop, float* r, float* y, float* h, float* xo,
                                                const float* const* nln_w, const float* const* nln_b, const float* const* nw_in,
                                                const float* const* nb_in, float* nqn, float* nq, float* nk, float* nv, void* stream) {
    AMID_CHECK_ARG(o && qn && w_o && b_o && ln_w && ln_b && w1 && b1 && w2 && b2 && r && y && h && xo && (!train || step_state));
    const bool next = nln_w != nullptr;
    AMID_CHECK_ARG(!next || (nln_b && nw_in && nb_in && nqn && nq && nk && nv));
    StripOffArgs a;
    a.o = o; a.qn = qn; a.tmq = tmq; a.r = r; a.y = y; a.h = h; a.xo = xo; a.ln_eps = ln_eps;
    a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.spec = drop_spec(p_drop);
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    StripQkvArgs nx = {};
    for (int g = 0; g < 2; ++g) {
        a.w_o[g] = w_o[g]; a.b_o[g] = b_o[g]; a.ln_w[g] = ln_w[g]; a.ln_b[g] = ln_b[g];
        a.w1[g] = w1[g]; a.b1[g] = b1[g]; a.w2[g] = w2[g]; a.b2[g] = b2[g];
        if (next) { nx.ln_w[g] = nln_w[g]; nx.ln_b[g] = nln_b[g]; nx.w_in[g] = nw_in[g]; nx.b_in[g] = nb_in[g]; }
    }
    if (next) { nx.x = xo; nx.qn = nqn; nx.q = nq; nx.k = nk; nx.v = nv; nx.ln_eps = ln_eps; }
    StripGeom sg;
    if (int e = make_strip_geom(B, T, D, live, &sg)) return e;
    if (D == 128 && next) return launch_strip<strip_oproj_ffn_fwd_kernel<128, true>, 128>(sg, stream, a, nx);
    if (D == 128) return launch_strip<strip_oproj_ffn_fwd_kernel<128, false>, 128>(sg, stream, a, nx);
    if (D == 64 && next) return launch_strip<strip_oproj_ffn_fwd_kernel<64, true>, 64>(sg, stream, a, nx);
    if (D == 64) return launch_strip<strip_oproj_ffn_fwd_kernel<64, false>, 64>(sg, stream, a, nx);
    return AMID_ERR_UNSUPPORTED;
}

static void fill_ffn_bwd(StripFfnBwdArgs& a, const float* dxo, const unsigned char* tmq, const float* h, const float* r,
                         const float* const* ln_w, const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps,
                         int layer, const void* step_state, int train, float p_drop, float* dpre2, float* dpre1, float* dr, float* d_o,
                         float* ln_part) {
    a.dxo = dxo; a.tmq = tmq; a.h = h; a.r = r; a.dpre2 = dpre2; a.dpre1 = dpre1; a.dr = dr; a.d_o = d_o; a.ln_part = ln_part;
    a.ln_eps = ln_eps; a.st = (const StepState*)step_state; a.layer = layer;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.spec = drop_spec(p_drop);
    a.scale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int g = 0; g < 2; ++g) { a.ln_w[g] = ln_w[g]; a.w1T[g] = w1T[g]; a.w2T[g] = w2T[g]; a.woT[g] = woT[g]; }
}

// ln_part: [2 * ceil(B T / amid_sas_strip_tile_rows())][2][D]; domain g's partial sums are slots [g * tpg, (g + 1) * tpg)
static int strip_ffn_bwd(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                         const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int B, int T,
                         int D, const int* live, int layer, const void* step_state, int train, float p_drop, float* dpre2,
                         float* dpre1, float* dr, float* d_o, float* ln_part, const void* sort_plan, int sort_phase, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(dxo && h && r && ln_w && w1T && w2T && woT && dpre2 && dpre1 && dr && d_o && ln_part && (!train || step_state));
    if (mma_bf16 && D != 128) return AMID_ERR_UNSUPPORTED;
    StripFfnBwdArgs a;
    fill_ffn_bwd(a, dxo, tmq, h, r, ln_w, w1T, w2T, woT, ln_eps, layer, step_state, train, p_drop, dpre2, dpre1, dr, d_o, ln_part);
    StripGeom sg;
    if (int e = make_strip_geom(B, T, D, live, &sg)) return e;
    SortRider rd;
    if (int e = make_rider(rd, sort_plan, sort_phase)) return e;
    if (rd.phase != 0 && rd.phase != 2) return AMID_ERR_UNSUPPORTED;           // this launch carries phase 2
    if (D == 128 && mma_bf16 == 3) return rd.phase ? launch_strip_rider<strip_ffn_bwd_kernel<128, 2, 3>, 128>(sg, rd, stream, a)
                                                   : launch_strip_rider<strip_ffn_bwd_kernel<128, 0, 3>, 128>(sg, rd, stream, a);
    if (D == 128 && mma_bf16) return rd.phase ? launch_strip_rider<strip_ffn_bwd_kernel<128, 2, 1>, 128>(sg, rd, stream, a)
                                              : launch_strip_rider<strip_ffn_bwd_kernel<128, 0, 1>, 128>(sg, rd, stream, a);
    if (D == 128 && rd.phase) return launch_strip_rider<strip_ffn_bwd_kernel<128, 2>, 128>(sg, rd, stream, a);
    if (D == 128) return launch_strip_rider<strip_ffn_bwd_kernel<128, 0>, 128>(sg, rd, stream, a);
    if (D == 64 && rd.phase) return launch_strip_rider<strip_ffn_bwd_kernel<64, 2>, 64>(sg, rd, stream, a);
    if (D == 64) return launch_strip_rider<strip_ffn_bwd_kernel<64, 0>, 64>(sg, rd, stream, a);
    return AMID_ERR_UNSUPPORTED;
}

extern "C" int amid_sas_strip_ffn_bwd_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                                          const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int B, int T,
                                          int D, const int* live, int layer, const void* step_state, int train, float p_drop, float* dpre2,
                                          float* dpre1, float* dr, float* d_o, float* ln_part, int mma_bf16, void* stream) {
    return strip_ffn_bwd(dxo, tmq, h, r, ln_w, w1T, w2T, woT, ln_eps, B, T, D, live, layer, step_state, train, p_drop, dpre2, dpre1, dr, d_o,
                         ln_part, nullptr, 0, mma_bf16, stream);
}

// ... carrying phase `sort_phase` (1 .. 4) of a sort plan (amid_sort_plan_pack) as extra workgroups in front of the tiles'
extern "C" int amid_sas_strip_ffn_bwd_sort_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r,
                                               const float* const* ln_w, const float* const* w1T, const float* const* w2T,
                                               const float* const* woT, float ln_eps, int B, int T, int D, const int* live, int layer,
                                               const void* step_state, int train, float p_drop, float* dpre2, float* dpre1, float* dr,
                                               float* d_o, float* ln_part, const void* sort_plan, int sort_phase, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr);
    return strip_ffn_bwd(dxo, tmq, h, r, ln_w, w1T, w2T, woT, ln_eps, B, T, D, live, layer, step_state, train, p_drop, dpre2, dpre1, dr, d_o,
                         ln_part, sort_plan, sort_phase, mma_bf16, stream);
}

// fh != NULL: the layer below's feed-forward / out-projection backward (f* arguments) runs on d x in the same launch; dx is then not written
static int strip_qkv_bwd(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                         const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                         float ln_eps, int B, int T, int D, const int* live, float* dx, float* ln_part,
                         const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w,
                         const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer,
                         const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr,
                         float* fd_o, float* fln_part, const void* sort_plan, int sort_phase, int mma_bf16, void* stream,
                         const unsigned char* emb_tmq = nullptr, float emb_p_drop = 0.f, const ScorerSum* scorer = nullptr) {
    AMID_CHECK_ARG(dq && dk && dv && dr && x && ln_w && wqT && wkT && wvT && ln_part);
    if (mma_bf16 && D != 128) return AMID_ERR_UNSUPPORTED;
    const bool ffn = fh != nullptr;
    AMID_CHECK_ARG(emb_tmq == nullptr || (!ffn && (!train || step_state)));
    AMID_CHECK_ARG(ffn || dx);
    AMID_CHECK_ARG(!ffn || (fr && fln_w && fw1T && fw2T && fwoT && fdpre2 && fdpre1 && fdr && fd_o && fln_part && (!train || step_state)));
    StripQkvBwdArgs a;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dr = dr; a.x = x; a.dx = dx; a.ln_part = ln_part; a.ln_eps = ln_eps;
    for (int g = 0; g < 2; ++g) { a.ln_w[g] = ln_w[g]; a.wqT[g] = wqT[g]; a.wkT[g] = wkT[g]; a.wvT[g] = wvT[g]; }
    a.emb_tmq = emb_tmq; a.emb_st = (const StepState*)step_state;
    a.emb_train = (emb_tmq && train && emb_p_drop > 0.f) ? 1 : 0;
    a.emb_spec = drop_spec(emb_p_drop);
    a.emb_scale = a.emb_train ? 1.0f / (1.0f - emb_p_drop) : 1.0f;
    StripFfnBwdArgs f = {};
    if (ffn) fill_ffn_bwd(f, nullptr, tmq, fh, fr, fln_w, fw1T, fw2T, fwoT, ln_eps, flayer, step_state, train, p_drop, fdpre2, fdpre1, fdr, fd_o, fln_part);
    StripGeom sg;
    if (int e = make_strip_geom(B, T, D, live, &sg)) return e;
    SortRider rd;
    if (int e = make_rider(rd, sort_plan, sort_phase)) return e;
    if (rd.phase != 0 && rd.phase != (ffn ? 3 : 4)) return AMID_ERR_UNSUPPORTED;      // phase 3 with the fused feed-forward backward, 4 without
    const bool ride = rd.phase != 0;
    ScorerSum ss = {};
    if (scorer != nullptr) {
        if (!(ride && ffn && D == 128 && (mma_bf16 == 3 || mma_bf16 == 1))) return AMID_ERR_UNSUPPORTED;       // the riders' host: the middle launch on bf16 pieces / one piece
        ss = *scorer;
    }
    if (D == 128 && mma_bf16 == 3 && ffn) return ride ? launch_strip_rider_x<strip_qkv_bwd_kernel<128, true, 3, 3>, 128>(sg, rd, ss.nblk, stream, a, f, ss)
                                                      : launch_strip_rider<strip_qkv_bwd_kernel<128, true, 0, 3>, 128>(sg, rd, stream, a, f, ss);
    if (D == 128 && mma_bf16 == 3) return ride ? launch_strip_rider<strip_qkv_bwd_kernel<128, false, 4, 3>, 128>(sg, rd, stream, a, f, ss)
                                               : launch_strip_rider<strip_qkv_bwd_kernel<128, false, 0, 3>, 128>(sg, rd, stream, a, f, ss);
    if (D == 128 && mma_bf16 && ffn) return ride ? launch_strip_rider_x<strip_qkv_bwd_kernel<128, true, 3, 1>, 128>(sg, rd, ss.nblk, stream, a, f, ss)
                                                 : launch_strip_rider<strip_qkv_bwd_kernel<128, true, 0, 1>, 128>(sg, rd, stream, a, f, ss);
    if (D == 128 && mma_bf16) return ride ? launch_strip_rider<strip_qkv_bwd_kernel<128, false, 4, 1>, 128>(sg, rd, stream, a, f, ss)
                                          : launch_strip_rider<strip_qkv_bwd_kernel<128, false, 0, 1>, 128>(sg, rd, stream, a, f, ss);
    if (D == 128 && ffn) return ride ? launch_strip_rider<strip_qkv_bwd_kernel<128, true, 3>, 128>(sg, rd, stream, a, f, ss)
                                     : launch_strip_rider<strip_qkv_bwd_kernel<128, true, 0>, 128>(sg, rd, stream, a, f, ss);
    if (D == 128) return ride ? launch_strip_rider<strip_qkv_bwd_kernel<128, false, 4>, 128>(sg, rd, stream, a, f, ss)
                              : launch_strip_rider<strip_qkv_bwd_kernel<128, false, 0>, 128>(sg, rd, stream, a, f, ss);
    if (D == 64 && ffn) return ride ? launch_strip_rider<strip_qkv_bwd_kernel<64, true, 3>, 64>(sg, rd, stream, a, f, ss)
                                    : launch_strip_rider<strip_qkv_bwd_kernel<64, true, 0>, 64>(sg, rd, stream, a, f, ss);
    if (D == 64) return ride ? launch_strip_rider<strip_qkv_bwd_kernel<64, false, 4>, 64>(sg, rd, stream, a, f, ss)
                             : launch_strip_rider<strip_qkv_bwd_kernel<64, false, 0>, 64>(sg, rd, stream, a, f, ss);
    return AMID_ERR_UNSUPPORTED;
}

extern "C" int amid_sas_strip_qkv_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                          const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                          float ln_eps, int B, int T, int D, const int* live, float* dx, float* ln_part,
                                          const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w,
                                          const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer,
                                          const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr,
                                          float* fd_o, float* fln_part, int mma_bf16, void* stream) {
    return strip_qkv_bwd(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, B, T, D, live, dx, ln_part, tmq, fh, fr, fln_w, fw1T, fw2T, fwoT, flayer,
                         step_state, train, p_drop, fdpre2, fdpre1, fdr, fd_o, fln_part, nullptr, 0, mma_bf16, stream);
}

// ... carrying phase `sort_phase` (1 .. 4) of a sort plan (amid_sort_plan_pack) as extra workgroups in front of the tiles'
extern "C" int amid_sas_strip_qkv_bwd_sort_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                               const float* const* ln_w, const float* const* wqT, const float* const* wkT,
                                               const float* const* wvT, float ln_eps, int B, int T, int D, const int* live, float* dx,
                                               float* ln_part, const unsigned char* tmq, const float* fh, const float* fr,
                                               const float* const* fln_w, const float* const* fw1T, const float* const* fw2T,
                                               const float* const* fwoT, int flayer, const void* step_state, int train, float p_drop,
                                               float* fdpre2, float* fdpre1, float* fdr, float* fd_o, float* fln_part, const void* sort_plan,
                                               int sort_phase, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr);
    return strip_qkv_bwd(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, B, T, D, live, dx, ln_part, tmq, fh, fr, fln_w, fw1T, fw2T, fwoT, flayer,
                         step_state, train, p_drop, fdpre2, fdpre1, fdr, fd_o, fln_part, sort_plan, sort_phase, mma_bf16, stream);
}

// amid_sas_strip_qkv_bwd_sort_f32 (with the fused feed-forward backward: phase 3 of the sort plan) carrying, as further extra workgroups,
// the scorer's weight gradients summed over the batch from the per-sample hidden gradients of amid_head_fwd_bwd_own_vec_f32 (hidg [B]
// [amid_scorer_vec_floats(NI, hid)]; u [2, B, D]; items [B, NI, D]): dW1 [hid, 2 D], db1 [hid], dW2 [hid], db2 [1].  The launch's live
// tiles leave CUs free at the headline shape; the sums depend on the head launch only.  D = 128, mma_bf16 = 3.
extern "C" int amid_sas_strip_qkv_bwd_sort_scorer_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                                      const float* const* ln_w, const float* const* wqT, const float* const* wkT,
                                                      const float* const* wvT, float ln_eps, int B, int T, int D, const int* live,
                                                      float* ln_part, const unsigned char* tmq, const float* fh, const float* fr,
                                                      const float* const* fln_w, const float* const* fw1T, const float* const* fw2T,
                                                      const float* const* fwoT, int flayer, const void* step_state, int train, float p_drop,
                                                      float* fdpre2, float* fdpre1, float* fdr, float* fd_o, float* fln_part,
                                                      const void* sort_plan, int sort_phase, int mma_bf16, const float* hidg, const float* u,
                                                      const float* items, int NI, int hid, float* dW1, float* db1, float* dW2, float* db2,
                                                      void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr && fh != nullptr && hidg && u && items && NI > 0 && hid > 0 && dW1 && db1 && dW2 && db2);
    AMID_CHECK_ARG(((((unsigned long long)dW1) | ((unsigned long long)u) | ((unsigned long long)items)) & 15) == 0);
    const ScorerSum ss = scorer_sum_args(hidg, u, items, B, NI, D, hid, dW1, db1, dW2, db2);
    return strip_qkv_bwd(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, B, T, D, live, nullptr, ln_part, tmq, fh, fr, fln_w, fw1T, fw2T, fwoT, flayer,
                         step_state, train, p_drop, fdpre2, fdpre1, fdr, fd_o, fln_part, sort_plan, sort_phase, mma_bf16, stream, nullptr, 0.f, &ss);
}

// layer 0's launch of the live-sequence train step with the embedding layer's backward on the strip (StripQkvBwdArgs::emb_tmq): dx = the
// gradient of the gathered rows (dropout keep bits of site SITE_EMB redrawn from step_state, then the "== 0" bits of emb_tmq applied) --
// amid_sas_strip_qkv_bwd(_sort)_f32 without the fused feed-forward followed by amid_embed_bwd_f32's element-wise part, in one launch
// (the position rows' gradient sums: amid_grad_tail_live_f32).  sort_plan may be NULL (no rider); with a plan the launch carries phase 4.
extern "C" int amid_sas_strip_qkv_bwd_emb_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                              const float* const* ln_w, const float* const* wqT, const float* const* wkT,
                                              const float* const* wvT, float ln_eps, int B, int T, int D, const int* live, float* dx,
                                              float* ln_part, const unsigned char* emb_tmq, const void* step_state, int train,
                                              float emb_p_drop, const void* sort_plan, int sort_phase, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(emb_tmq != nullptr && dx != nullptr);
    return strip_qkv_bwd(dq, dk, dv, dr, x, ln_w, wqT, wkT, wvT, ln_eps, B, T, D, live, dx, ln_part, nullptr, nullptr, nullptr, nullptr, nullptr,
                         nullptr, nullptr, 0, step_state, train, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, sort_plan, sort_phase, mma_bf16,
                         stream, emb_tmq, emb_p_drop);
}

// ---- the fused per-sequence backward ------------------------------------------------------------------------------------------------
template <int D> static constexpr size_t seq_bwd_lds_bytes() { return strip_lds_bytes<D>() + (size_t)STRIP_WAVES * ATTN_BWD_LDS_PER_WAVE; }

// 1 when amid_sas_seq_bwd_f32 covers the shape: 8 heads with D = 128 or D = 64, T <= 64 (one sequence per workgroup; T <= 32 and D = 64:
// the N-split build only, which needs p_drop = 0.5 or eval mode and has no bf16 products at D 64), activations within 2 GiB
extern "C" int amid_sas_seq_bwd_supported(int B, int T, int D, int H) {
    return ((D == 128 || D == 64) && H == 8 && T > 0 && T <= 64 && B > 0 && 2LL * B * T * D * 4 <= 0x7FFFFFF0LL) ? 1 : 0;
}

// Which build of the one-launch backward runs: 0 = auto (= 1), 1 = a wave per strip (seq_bwd_kernel), 2 = the N-split build (two waves per
// strip, a wave per head in the attention core: sasrec_seqn_bwd.hip; at T > 32, D 128 that build exists in the diagnostic library only
// -- -DAMID_DIAG_VARIANTS -- and the product answers 2 with the strip build).  v < 0 only queries.  Returns the previous value.
extern "C" int amid_diag_variants(void) {
#ifdef AMID_DIAG_VARIANTS
    return 1;
#else
    return 0;
#endif
}
static int g_seq_bwd_variant = 0;
extern "C" int amid_sas_seq_bwd_variant(int v) {
    const int prev = g_seq_bwd_variant;
    if (v >= 0) g_seq_bwd_variant = v;
    return prev;
}

// The data gradients of the encoder for the live sequences of a train step in one launch (seq_bwd_kernel / seqn_bwd_kernel).  Per-layer arrays hold
// n_layers entries (saved tensors, gradient outputs, LayerNorm partials) or 2 * n_layers ordered [layer][domain] (parameters and
// transposed weights).  dxo: gradient of the last layer's output; dx: gradient of layer 0's input (rows of live sequences only; the
// others are not touched).  ln1_part / ln2_part: per layer [2 B][2][D], domain g's slots are [g B, (g + 1) B), the first n_g of them
// the live sequences' partial sums, the rest zeros.  live: amid_live_list_i32 (required).
extern "C" int amid_sas_seq_bwd_f32(int n_layers, const float* dxo, const unsigned char* tmq, const float* const* h, const float* const* r,
                                    const float* const* x, const float* const* q, const float* const* k, const float* const* v,
                                    const float* const* o, const float* const* stats, const float* const* ln1_w,
                                    const float* const* ln2_w, const float* const* wqT, const float* const* wkT, const float* const* wvT,
                                    const float* const* woT, const float* const* w1T, const float* const* w2T, float ln_eps, int B, int T,
                                    int D, int H, const int* live, const void* step_state, int train, float p_drop, float* const* dpre2,
                                    float* const* dpre1, float* const* dr, float* d_o, float* const* dq, float* const* dk, float* const* dv,
                                    float* dx, float* const* ln1_part, float* const* ln2_part, int mma_bf16, void* stream) {
    AMID_CHECK_ARG(n_layers >= 1 && n_layers <= 2 && dxo && h && r && x && q && k && v && o && stats && ln1_w && ln2_w && wqT && wkT && wvT &&
                   woT && w1T && w2T && live && dpre2 && dpre1 && dr && d_o && dq && dk && dv && dx && ln1_part && ln2_part &&
                   (!train || step_state));
    if (!amid_sas_seq_bwd_supported(B, T, D, H)) return AMID_ERR_UNSUPPORTED;
    SeqBwdArgs a = {};
    a.n_layers = n_layers;
    for (int l = 0; l < n_layers; ++l) {
        SeqBwdLayer& P = a.L[l];
        AMID_CHECK_ARG(h[l] && r[l] && x[l] && q[l] && k[l] && v[l] && o[l] && stats[l] && dpre2[l] && dpre1[l] && dr[l] && dq[l] && dk[l] &&
                       dv[l] && ln1_part[l] && ln2_part[l]);
        for (int g = 0; g < 2; ++g) {
            const int i = 2 * l + g;
            AMID_CHECK_ARG(ln1_w[i] && ln2_w[i] && wqT[i] && wkT[i] && wvT[i] && woT[i] && w1T[i] && w2T[i]);
        }
        fill_ffn_bwd(P.f, l + 1 == n_layers ? dxo : nullptr, tmq, h[l], r[l], ln2_w + 2 * l, w1T + 2 * l, w2T + 2 * l, woT + 2 * l, ln_eps, l,
                     step_state, train, p_drop, dpre2[l], dpre1[l], dr[l], d_o, ln2_part[l]);
        P.a.dq = dq[l]; P.a.dk = dk[l]; P.a.dv = dv[l]; P.a.dr = dr[l]; P.a.x = x[l]; P.a.dx = l == 0 ? dx : nullptr;
        P.a.ln_part = ln1_part[l]; P.a.ln_eps = ln_eps;
        for (int g = 0; g < 2; ++g) {
            P.a.ln_w[g] = ln1_w[2 * l + g]; P.a.wqT[g] = wqT[2 * l + g]; P.a.wkT[g] = wkT[2 * l + g]; P.a.wvT[g] = wvT[2 * l + g];
        }
        AttnArgs& at = P.at;
        at.q = q[l]; at.k = k[l]; at.v = v[l]; at.o = const_cast<float*>(o[l]); at.stats = const_cast<float*>(stats[l]);
        at.d_o = d_o; at.dq = dq[l]; at.dk = dk[l]; at.dv = dv[l];
        at.B = B; at.T = T; at.D = D; at.H = H; at.causal = 1; at.layer = l;
        at.scale = sqrtf(1.0f / (float)(D / H));
        at.st = (const StepState*)step_state;
        at.train = (train && p_drop > 0.f) ? 1 : 0;
        at.thr16 = drop_spec(p_drop);
        at.dscale = at.train ? 1.0f / (1.0f - p_drop) : 1.0f;
        at.stagger_from = -1;
    }
    StripGeom sg;
    if (int e = make_strip_geom(B, T, D, live, &sg)) return e;
    if (g_seq_bwd_variant == 2 || T <= 32 || D != 128) {      // (T > 32 at D 128: auto = the strip build, the two measure the same -- DESIGN.md
        const int rc = launch_seqn_bwd(a, sg, D, mma_bf16, stream);      // section 5.0; T <= 32 and D 64: the N-split build only)
        if (rc != AMID_ERR_UNSUPPORTED || T <= 32 || D != 128) return rc;
    }
    static unsigned long long attr_done[2] = {0, 0};
    if (int rc = mma_bf16 ? lds_attr_once((const void*)seq_bwd_kernel<128, true>, seq_bwd_lds_bytes<128>(), attr_done[1])
                          : lds_attr_once((const void*)seq_bwd_kernel<128, false>, seq_bwd_lds_bytes<128>(), attr_done[0])) return rc;
    if (mma_bf16) seq_bwd_kernel<128, true><<<B, STRIP_THREADS, seq_bwd_lds_bytes<128>(), (hipStream_t)stream>>>(a, sg);
    else seq_bwd_kernel<128, false><<<B, STRIP_THREADS, seq_bwd_lds_bytes<128>(), (hipStream_t)stream>>>(a, sg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}
