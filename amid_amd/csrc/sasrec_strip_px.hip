// The backward strips of the SASRec layer, N-split build on producer-side bf16 pieces (round 6) -- the data gradients of sasrec_strip.hip's
// chains (strip_ffn_bwd_kernel, strip_qkv_bwd_kernel) with the forward's machinery (sasrec_seqn.hip seqn_fwd_px_body, seqn_parts.h):
// a workgroup = a 64-row tile = four 16-row strips x TWO column parts = eight waves, two per SIMD; a wave owns D / 2 output columns of every
// product, the operands made inside a chain cross the strip's two waves through LDS AS bf16 PIECES (xp_write: the next product's operand
// fragments), the transposed weights stream through three 32 KB plane slots (SeqRing3; the three-plane images of amid_step_head_w16_f32 /
// amid_embed_fwd_w16_f32).  Why: in the strip build a SIMD holds ONE wave, whose product is a chain of fragment reads and matrix
// instructions that nothing covers -- 6.5 us per 64 x 128 x 128 product against 1.3 us of matrix issue; the forward's build runs the same
// product in ~ 3 us with two waves per SIMD.  Reference: autograd of Log2feats.forward (model_seq.py:371-383) under loss.backward(),
// train_sr.py:214.  Same operations and operands as the strip build; row sums of the LayerNorm backward are added part by part (the strip
// build adds them column tile by column tile), so the two builds agree to rounding, not bit for bit.
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "strip_chain.h"
#include "seq_fwd.h"
#include "seq_bwd.h"
#include "seqn_parts.h"
#include "sort_phases.h"

namespace amid {

constexpr int PXB_WPS = 4, PXB_NS = 2, PXB_NW = PXB_WPS * PXB_NS, PXB_THREADS = 64 * PXB_NW;
// LDS: [three plane slots][four strips' exchange slots][row sums: 2 x strips x parts x 16][LayerNorm partials: strips x 2 x D]
template <int D> constexpr size_t pxb_lds_floats() {
    return (size_t)3 * (D * D / 2) + (size_t)PXB_WPS * XpStrip<D>::FLOATS + 2 * PXB_NW * 16 + (size_t)PXB_WPS * 2 * D;
}

// this lane's row of the tile: strip si, row m of it
struct PxbRow { unsigned off_own, phys; int local; bool ok; };
template <int D>
__device__ __forceinline__ PxbRow pxb_row(const StripGeom& sg, const StripTile& t, int si, int c0) {
    PxbRow r;
    int v = t.v0 + si * 16 + (lane_id() & 15);
    r.ok = v < t.nv;
    if (!r.ok) v = t.v0;
    if (sg.live != nullptr) {
        const int s = v / sg.T;
        r.local = sg.live[t.s0 + s] * sg.T + (v - s * sg.T);
    } else {
        r.local = v;
    }
    r.phys = (unsigned)t.g * (unsigned)sg.M + (unsigned)r.local;
    r.off_own = r.ok ? r.phys * (unsigned)(D * 4) + 16u * (unsigned)(lane_id() >> 4) + (unsigned)c0 * 64u : STRIP_OOB;
    return r;
}

// the row's two sums of the LayerNorm backward over the WHOLE row: own columns, then the partner's through `stat`
template <int NCT>
__device__ __forceinline__ void pxb_row_sums(float* __restrict__ stat, int si, int part, float s1, float s2, float& t1, float& t2) {
    const int m = lane_id() & 15, gq = lane_id() >> 4;
    s1 = row_sum4(s1); s2 = row_sum4(s2);
    if (gq == 0) {
        lds_st1(stat + (si * PXB_NS + part) * 16 + m, s1);
        lds_st1(stat + (PXB_NW + si * PXB_NS + part) * 16 + m, s2);
    }
    lds_barrier();
    t1 = 0.f; t2 = 0.f;
#pragma unroll
    for (int p = 0; p < PXB_NS; ++p) {
        t1 += *(__attribute__((address_space(3))) float*)(stat + (si * PXB_NS + p) * 16 + m);
        t2 += *(__attribute__((address_space(3))) float*)(stat + (PXB_NW + si * PXB_NS + p) * 16 + m);
    }
}

// column sums over the strip's 16 rows of the own column tiles -> the strip's slice of the scratch [strips][2][D]
template <int D, int NCT>
__device__ __forceinline__ void pxb_ln_partials(float* __restrict__ scratch, int si, int c0, const PartRegs<NCT>& dgam, const PartRegs<NCT>& dbet) {
    const int lane = lane_id();
    float* mine = scratch + si * 2 * D;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        f32x4 a, b;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = col_sum16(dgam.v[c][r]); b[r] = col_sum16(dbet.v[c][r]); }
        if ((lane & 15) == 0) {
            lds_st4(mine + (c0 + c) * 16 + 4 * (lane >> 4), a);
            lds_st4(mine + D + (c0 + c) * 16 + 4 * (lane >> 4), b);
        }
    }
}
template <int D>
__device__ __forceinline__ void pxb_ln_partials_out(const float* __restrict__ scratch, float* __restrict__ part) {
    for (int e = threadIdx.x; e < 2 * D; e += PXB_THREADS)
        part[e] = (scratch[e] + scratch[2 * D + e]) + (scratch[4 * D + e] + scratch[6 * D + e]);
}

// d x' (own columns, in DZ) -> dpre2, dpre1, dr, d_o of the layer: strip_chain.h ffn_bwd_chain on the own column tiles.  The ring's pending
// weight must be w2T; `wnext`: the image the last product announces (a fused successor's first weight, or any valid image).
// ln_stat: the forward's row statistics [2M][4] (LN2's mean, rstd at +2) or nullptr (the statistics are taken from r's row then).
template <int D, int NCT, class Ring>
__device__ __forceinline__ void pxb_ffn_bwd_chain(const StripFfnBwdArgs& a, const float* __restrict__ ln_stat, const StripGeom& sg, Ring& ring,
                                                  const PxbRow& row, int g, int si, int part, int c0, PartRegs<NCT>& DZ, float* __restrict__ xps,
                                                  float* __restrict__ stat, float* __restrict__ lnsc, const unsigned short* __restrict__ wnext) {
    const int lane = lane_id(), gq = lane >> 4;
    const GBuf gp2(a.dpre2, sg.act_bytes), gp1(a.dpre1, sg.act_bytes), gdr(a.dr, sg.act_bytes), gdo(a.d_o, sg.act_bytes);
    PartRegs<NCT> P, Hs, Rs, gam;
    part_load<NCT>(Hs, GBuf(a.h, sg.act_bytes), row.off_own);
    part_load<NCT>(Rs, GBuf(a.r, sg.act_bytes), row.off_own);
    part_cols<NCT>(gam, a.ln_w[g], c0);
    float mean = 0.f, rstd = 0.f;
    if (ln_stat != nullptr && row.ok) { mean = ln_stat[(long long)row.phys * 4 + 2]; rstd = ln_stat[(long long)row.phys * 4 + 3]; }
    if (a.tmq != nullptr) {
        const GBuf gtm(a.tmq, sg.tm_bytes);
        const unsigned tbase = row.ok ? row.phys * (unsigned)(D / 4) + (unsigned)c0 * 4u : STRIP_OOB;
        const int sh = 8 * gq;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const unsigned bits = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(gtm.r, (int)(tbase + 4 * c), 0, 0) >> sh;
            DZ.v[c][0] = (bits & 1u) ? 0.f : DZ.v[c][0];
            DZ.v[c][1] = (bits & 2u) ? 0.f : DZ.v[c][1];
            DZ.v[c][2] = (bits & 4u) ? 0.f : DZ.v[c][2];
            DZ.v[c][3] = (bits & 8u) ? 0.f : DZ.v[c][3];
        }
    }
    // dpre2 = dz * drop2
    P = DZ;
    if (a.train) {
        const uint4 rr2 = rng_call(a.st->seed, (unsigned long long)row.local * D >> 7, site_id(g, a.layer, SITE_FFN2), (unsigned)a.st->step);
        part_dropout<NCT>(P, rr2, c0, a.spec, a.scale, (row.local * D) & 127);
    }
    xp_write<NCT>(xps, c0, P);
    f32x4 acc[NCT];
    {   // dh = dpre2 C2 ; dpre1 = dh * relu'(h) * drop1   (h > 0 implies the unit was kept by drop1)
        ring.next();
        seqn_product_xp<D, NCT>(acc, xps, ring, (const unsigned short*)a.w1T[g], c0, [&](int ct, int j) { part_spread<NCT>(gp2, row.off_own, P, ct, j, 1); });
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) P.v[c][r] = Hs.v[c][r] > 0.f ? acc[c][r] * a.scale : 0.f;
    }
    lds_barrier();                                      // the strip's two waves have read the last fragment of dpre2
    xp_write<NCT>(xps, c0, P);
    PartRegs<NCT> DR, dgam, dbet;
    {   // dy = dpre1 C1 + dz ; dr = LN2'(dy ; r)
        ring.next();
        seqn_product_xp<D, NCT>(acc, xps, ring, (const unsigned short*)a.woT[g], c0, [&](int ct, int j) { part_spread<NCT>(gp1, row.off_own, P, ct, j, 1); });
        if (ln_stat == nullptr) {                       // r's row statistics, part by part (sasrec_seqn.hip's LN2)
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c) sm += (Rs.v[c][0] + Rs.v[c][1]) + (Rs.v[c][2] + Rs.v[c][3]);
            float tot, dummy;
            pxb_row_sums<NCT>(stat, si, part, sm, 0.f, tot, dummy);
            mean = tot * (1.0f / D);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = Rs.v[c][r] - mean; q = fmaf(d, d, q); }
            lds_barrier();
            float qt;
            pxb_row_sums<NCT>(stat, si, part, q, 0.f, qt, dummy);
            rstd = 1.0f / sqrtf(qt * (1.0f / D) + a.ln_eps);
            lds_barrier();
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dy = acc[c][r] + DZ.v[c][r];
                const float xh = (Rs.v[c][r] - mean) * rstd;
                float gy = gam.v[c][r] * dy;
                asm volatile("" : "+v"(gy));            // the rounded product (strip_ln_bwd)
                s1 += gy;
                s2 = fmaf(gy, xh, s2);
                float dg = dy * xh;
                asm volatile("" : "+v"(dg));
                dgam.v[c][r] = dg;
                dbet.v[c][r] = dy;
                DR.v[c][r] = gy;
            }
        }
        float t1, t2;
        pxb_row_sums<NCT>(stat, si, part, s1, s2, t1, t2);
        const float c1 = t1 * (1.0f / D), c2 = t2 * (1.0f / D);
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (Rs.v[c][r] - mean) * rstd;
                DR.v[c][r] = rstd * (DR.v[c][r] - c1 - xh * c2);
            }
    }
    // (the row sums' barrier lies behind the last fragment read of dpre1: the slots are free)
    xp_write<NCT>(xps, c0, DR);
    {   // d_o = dr Wo
        ring.next();
        seqn_product_xp<D, NCT>(acc, xps, ring, wnext, c0, [&](int ct, int j) { part_spread<NCT>(gdr, row.off_own, DR, ct, j, 1); });
#pragma unroll
        for (int c = 0; c < NCT; ++c) P.v[c] = acc[c];
        part_store<NCT>(gdo, row.off_own, P);
    }
    pxb_ln_partials<D, NCT>(lnsc, si, c0, dgam, dbet);
}

// RIDER: 0, or the phase of the step's index sort the first rd.plan.nblk workgroups run (their first four waves; sort_phases.h)
template <int D, int RIDER>
__global__ __launch_bounds__(PXB_THREADS) void strip_ffn_bwd_px_kernel(const StripFfnBwdArgs a, const float* __restrict__ ln_stat, const StripGeom sg,
                                                                       const SortRider rd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NCT = D / 16 / PXB_NS;
    int bid = blockIdx.x;
    if constexpr (RIDER != 0) {
        if (bid < rd.plan.nblk) { if (threadIdx.x < SORT_THREADS) sort_phase_ct<RIDER>(rd.plan, bid, smem); return; }
        bid -= rd.plan.nblk;
    }
    const int w = wave_id(), part = w / PXB_WPS, si = w % PXB_WPS, c0 = part * NCT;
    const int g = strip_domain(bid);
    SeqRing3<D, PXB_NW> ring(smem);
    ring.first((const unsigned short*)a.w2T[g]);
    const StripTile t = strip_tile(sg, bid);
    if (!t.live) { zero_slot<D>(a.ln_part, t.slot); w_ring_wait(); return; }
    float* const xps = smem + 3 * SeqRing3<D, PXB_NW>::SLAB + si * XpStrip<D>::FLOATS;
    float* const stat = smem + 3 * SeqRing3<D, PXB_NW>::SLAB + PXB_WPS * XpStrip<D>::FLOATS;
    float* const lnsc = stat + 2 * PXB_NW * 16;
    const PxbRow row = pxb_row<D>(sg, t, si, c0);
    PartRegs<NCT> DZ;
    part_load<NCT>(DZ, GBuf(a.dxo, sg.act_bytes), row.off_own);
    pxb_ffn_bwd_chain<D, NCT>(a, ln_stat, sg, ring, row, g, si, part, c0, DZ, xps, stat, lnsc, (const unsigned short*)a.woT[g]);
    w_ring_wait();                                      // (the last product announced a plane: its DMA must not outlive the workgroup's LDS)
    __syncthreads();
    pxb_ln_partials_out<D>(lnsc, a.ln_part + (long long)t.slot * 2 * D);
}

}  // namespace amid

using namespace amid;
using namespace amid_strip_host;

static void pxb_fill_ffn_bwd(StripFfnBwdArgs& a, const float* dxo, const unsigned char* tmq, const float* h, const float* r,
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

// amid_sas_strip_ffn_bwd_f32 (mma_bf16 = 3: the weights are three-plane images of the transposes) as the N-split build on pieces.
// ln_stat: the forward's row statistics of this layer ([2 B T][4]: LN2's mean and rstd at +2) or NULL.  sort_plan / sort_phase: optional rider.
extern "C" int amid_sas_strip_ffn_bwd_px_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                                             const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int B, int T,
                                             int D, const int* live, int layer, const void* step_state, int train, float p_drop, float* dpre2,
                                             float* dpre1, float* dr, float* d_o, float* ln_part, const float* ln_stat, const void* sort_plan,
                                             int sort_phase, void* stream) {
    AMID_CHECK_ARG(dxo && h && r && ln_w && w1T && w2T && woT && dpre2 && dpre1 && dr && d_o && ln_part && (!train || step_state));
    if (D != 128) return AMID_ERR_UNSUPPORTED;
    if (train && p_drop > 0.f && spec_bits(drop_spec(p_drop)) != 1) return AMID_ERR_UNSUPPORTED;       // part_dropout: p = 0.5
    StripGeom sg;
    if (int e = make_strip_geom(B, T, D, live, &sg)) return e;
    StripFfnBwdArgs a;
    pxb_fill_ffn_bwd(a, dxo, tmq, h, r, ln_w, w1T, w2T, woT, ln_eps, layer, step_state, train, p_drop, dpre2, dpre1, dr, d_o, ln_part);
    SortRider rd;
    rd.phase = 0;
    if (sort_plan != nullptr) {
        if (sort_phase != 2) return AMID_ERR_UNSUPPORTED;
        rd.plan = *(const SortPlan*)sort_plan;
        rd.phase = sort_phase;
    }
    constexpr size_t lds = pxb_lds_floats<128>() * sizeof(float);
    static_assert(lds >= sizeof(SortScatterLds<OS_BINS_MAX>), "the rider's scatter fits the head of the allocation");
    if (rd.phase != 0) {
        static unsigned long long done = 0;
        if (int rc = lds_attr_once((const void*)strip_ffn_bwd_px_kernel<128, 2>, lds, done)) return rc;
        strip_ffn_bwd_px_kernel<128, 2><<<2 * sg.tpg + rd.plan.nblk, PXB_THREADS, lds, (hipStream_t)stream>>>(a, ln_stat, sg, rd);
    } else {
        static unsigned long long done = 0;
        if (int rc = lds_attr_once((const void*)strip_ffn_bwd_px_kernel<128, 0>, lds, done)) return rc;
        strip_ffn_bwd_px_kernel<128, 0><<<2 * sg.tpg, PXB_THREADS, lds, (hipStream_t)stream>>>(a, ln_stat, sg, rd);
    }
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
