// The building blocks of the N-split kernels (sasrec_seqn.hip: the one-launch encoder forward; sasrec_seqn_bwd.hip: the one-launch
// backward): NS waves share a 16-row strip, each owning D / NS output columns of every product.  Weight ring for NW waves (fp32 images
// by LDS-DMA, bf16 fragment images), the own-column register tiles, the exchange of parts through LDS, the MFMA loops over the own
// column tiles.
#pragma once
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include "bf16_pieces.h"
#include <type_traits>
#include "seq_fwd.h"

namespace amid {

typedef __attribute__((address_space(3))) float lds_f;
typedef float lds_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 lds_ld4(const float* p) {
    const lds_v4 t = *(const __attribute__((address_space(3))) lds_v4*)(p);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void lds_st4(float* p, f32x4 v) { *(__attribute__((address_space(3))) lds_v4*)(p) = lds_v4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void lds_st1(float* p, float v) { *(__attribute__((address_space(3))) float*)(p) = v; }

// ---- weight ring for NW waves (strip_gemm.h's WDma with the wave count as a parameter) ----------------------------------------------
template <int D, int NW> struct WDmaN {
    static constexpr int CPR = D / 4;
    static constexpr int PER_WAVE = D * CPR / 64 / NW;
    static constexpr unsigned STRIDE2 = 2u * (NW * 64 / CPR) * D * 4;      // bytes between pieces k0 and k0 + 2
    unsigned off[2];
    int w;
    __device__ __forceinline__ WDmaN() {
        const int lane = lane_id();
        w = wave_id();
#pragma unroll
        for (int k0 = 0; k0 < 2; ++k0) {
            const int p = (k0 * NW + w) * 64 + lane;
            const int n = p / CPR, pos = p % CPR;
            off[k0] = (unsigned)((n * D + ((pos ^ (n & 15)) * 4)) * 4);
        }
    }
    __device__ __forceinline__ void piece(float* __restrict__ buf, const float* __restrict__ W, int k0) const {
        const unsigned voff = off[k0 & 1] + (unsigned)(k0 >> 1) * STRIDE2;
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(buf + (k0 * NW + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
};

template <int D, int NW> struct SeqRingN {
    static constexpr int NWAVES = NW;
    float* buf; int s; WDmaN<D, NW> dma;
    static constexpr int SLOTS = 4 * (D / 16);
    __device__ __forceinline__ explicit SeqRingN(float* lds) : buf(lds), s(0) {}
    __device__ __forceinline__ void first(const float* __restrict__ W0) {
#pragma unroll
        for (int k0 = 0; k0 < WDmaN<D, NW>::PER_WAVE; ++k0) dma.piece(buf, W0, k0);
    }
    __device__ __forceinline__ float* next() {
        w_ring_wait();
        __syncthreads();
        float* cur = buf + (s & 1) * D * D;
        ++s;
        return cur;
    }
    // slot = ct * 4 + j of the product that reads slab s - 1; pieces go into the other buffer over the first half of the loop
    __device__ __forceinline__ void fetch(const float* __restrict__ W, int ct, int j) const {
        constexpr int PW = WDmaN<D, NW>::PER_WAVE, EVERY = (SLOTS / 2) / PW > 0 ? (SLOTS / 2) / PW : 1;
        const int slot = ct * 4 + j;
        if (slot % EVERY == 0 && slot / EVERY < PW) dma.piece(buf + (s & 1) * D * D, W, slot / EVERY);
    }
};

// ---- bf16 weight images (amid_sas_weights_bf16): row n = D bf16 = D / 8 chunks of 16 bytes; chunk 4 s + g of a row holds the eight k
// values lane group g supplies in k-step s of v_mfma_f32_16x16x32_bf16 when the operand sits in the C layout: k = 32 s + 4 g + r
// (column tile 2 s, elements 0..3) and k = 32 s + 16 + 4 g + r (column tile 2 s + 1, elements 4..7).  In LDS chunk c of row n sits at
// chunk position c ^ (n & 15), applied on the DMA's source address as for the fp32 images: conflict-free ds_read_b128 fragments.
template <int D, int NW> struct SeqRing16 {
    static constexpr int NWAVES = NW;
    static constexpr int CPR = D / 8;                                   // 16-byte chunks per row
    static constexpr int PIECES = D * CPR / 64, PER_WAVE = PIECES / NW;
    static constexpr int SLAB = D * D / 2;                              // floats per slab (32 KB at D = 128)
    float* buf; int s; unsigned off0; int w;
    __device__ __forceinline__ explicit SeqRing16(float* lds) : buf(lds), s(0) {
        w = wave_id();
        const int p = w * 64 + lane_id();
        const int n = p / CPR, pos = p % CPR;
        off0 = (unsigned)(n * D * 2 + ((pos ^ (n & 15)) * 16));
    }
    __device__ __forceinline__ void piece(float* __restrict__ dst, const unsigned short* __restrict__ W, int k0) const {
        const unsigned voff = off0 + (unsigned)k0 * (unsigned)(NW * 64 / CPR) * (unsigned)(D * 2);      // NW * 4 rows further: n & 15 unchanged
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(dst + (k0 * NW + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
    __device__ __forceinline__ void first(const unsigned short* __restrict__ W0) {
#pragma unroll
        for (int k0 = 0; k0 < PER_WAVE; ++k0) piece(buf, W0, k0);
    }
    __device__ __forceinline__ float* next() {
        w_ring_wait();
        __syncthreads();
        float* cur = buf + (s & 1) * SLAB;
        ++s;
        return cur;
    }
    __device__ __forceinline__ void fetch_all(const unsigned short* __restrict__ W) const {
#pragma unroll
        for (int k0 = 0; k0 < PER_WAVE; ++k0) piece(buf + (s & 1) * SLAB, W, k0);
    }
};

// ---- fp32 products on the bf16 matrix cores (compute = "fp32", three bf16 pieces per operand: csrc/bf16_pieces.h has the arithmetic).
// The weights arrive as THREE bf16 fragment images each (planes hi, mid, lo: amid_sas_weights_bf16_planes) = 96 KB per weight, and two
// whole weights do not fit the 128 KB the ring has.  So the ring is four 32 KB plane slots [M][L][H0][H1] and a product walks its
// weight plane by plane -- lo x hi and mid x (mid, hi) of the operand first, ONE barrier, then hi x (lo, mid, hi): the six piece pairs,
// 96 matrix instructions of 16 cycles per wave instead of 128 of 32.  Every plane is requested well ahead of its pass: the hi plane of
// product j + 1 when product j begins (the other H slot is free then), its mid and lo planes behind product j's barrier (M and L are
// free then: a hi pass + the epilogue ahead of their use).  The attention images (64 KB) take H0 + H1 between the q and the
// out-projection products: the q product does not prefetch (`hold_next`), the out-projection requests its own hi plane when it begins
// and waits for it at its barrier.
template <int D, int NW> struct SeqRing16x3 {
    static constexpr int NWAVES = NW;
    static constexpr int CPR = D / 8;
    static constexpr int PIECES = D * CPR / 64, PER_WAVE = PIECES / NW;
    static constexpr int SLAB = D * D / 2;                              // floats per plane slot (32 KB at D = 128)
    float* buf; int s; unsigned off0; int w;
    const unsigned short* cur; const unsigned short* nxt;
    bool cur_hi_ready, nxt_hi_ready, hi_late, hold_next;
    __device__ __forceinline__ explicit SeqRing16x3(float* lds)
        : buf(lds), s(0), cur(nullptr), nxt(nullptr), cur_hi_ready(false), nxt_hi_ready(false), hi_late(false), hold_next(false) {
        w = wave_id();
        const int p = w * 64 + lane_id();
        const int n = p / CPR, pos = p % CPR;
        off0 = (unsigned)(n * D * 2 + ((pos ^ (n & 15)) * 16));
    }
    __device__ __forceinline__ float* mslot() const { return buf; }
    __device__ __forceinline__ float* lslot() const { return buf + SLAB; }
    __device__ __forceinline__ float* hslot(int k) const { return buf + (2 + (k & 1)) * SLAB; }
    __device__ __forceinline__ float* images() const { return buf + 2 * SLAB; }        // H0 + H1
    __device__ __forceinline__ float* hcur() const { return hslot(s - 1); }
    __device__ __forceinline__ void piece(float* __restrict__ dst, const unsigned short* __restrict__ W, int k0) const {
        const unsigned voff = off0 + (unsigned)k0 * (unsigned)(NW * 64 / CPR) * (unsigned)(D * 2);
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(dst + (k0 * NW + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
    __device__ __forceinline__ void plane(float* __restrict__ dst, const unsigned short* __restrict__ Wp) const {
#pragma unroll
        for (int k0 = 0; k0 < PER_WAVE; ++k0) piece(dst, Wp, k0);
    }
    __device__ __forceinline__ void first(const unsigned short* __restrict__ W0) {
        cur = W0; cur_hi_ready = true;
        plane(hslot(0), W0); plane(mslot(), W0 + (size_t)D * D); plane(lslot(), W0 + (size_t)2 * D * D);
    }
    __device__ __forceinline__ float* next() {              // product s begins: every wave is past product s - 1 (and the images)
        w_ring_wait();
        __syncthreads();
        ++s;
        return hslot(s - 1);
    }
    __device__ __forceinline__ void begin(const unsigned short* __restrict__ Wnext) {
        nxt = Wnext;
        hi_late = !cur_hi_ready;
        if (hi_late) plane(hslot(s - 1), cur);             // (the product behind the attention core: its H slot held an image)
        nxt_hi_ready = !hold_next;
        if (nxt_hi_ready) plane(hslot(s), nxt);
        hold_next = false;
    }
    // behind the lo / mid passes: M and L are free for the next weight's planes
    __device__ __forceinline__ void mid_sync() {
        if (hi_late) w_ring_wait();
        __syncthreads();
        plane(mslot(), nxt + (size_t)D * D);
        plane(lslot(), nxt + (size_t)2 * D * D);
        cur = nxt; cur_hi_ready = nxt_hi_ready;
    }
};

typedef __bf16 seqn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 seqn_bf16x2 __attribute__((ext_vector_type(2)));
typedef float seqn_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned seqn_pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(seqn_f32x2{a, b}, seqn_bf16x2));       // v_cvt_pk_bf16_f32: round to nearest even
}
// acc[c] += A W^T over the own column tiles with bf16 operands: 4 k-steps of 32; the operand's eight values of step s are the lane's
// elements of column tiles 2 s and 2 s + 1
template <int D, int NCT>
__device__ __forceinline__ void part_mma16(f32x4 (&acc)[NCT], const StripRegs<D>& A, const float* __restrict__ buf, int c0) {
    constexpr int KS = D / 32;
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const float* rowp = buf + (c0 * 16 + i) * (D / 2);                  // a row = D bf16 = D / 2 floats
    amid_v4u a16[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        a16[s] = amid_v4u{seqn_pack2(A.v[2 * s][0], A.v[2 * s][1]), seqn_pack2(A.v[2 * s][2], A.v[2 * s][3]),
                          seqn_pack2(A.v[2 * s + 1][0], A.v[2 * s + 1][1]), seqn_pack2(A.v[2 * s + 1][2], A.v[2 * s + 1][3])};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float4 wf[NCT];
#pragma unroll
        for (int c = 0; c < NCT; ++c) wf[c] = lds_ld4(rowp + c * 16 * (D / 2) + 4 * ((4 * s + g) ^ i));
#pragma unroll
        for (int c = 0; c < NCT; ++c)
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(seqn_bf16x8, wf[c]), __builtin_bit_cast(seqn_bf16x8, a16[s]),
                                                             acc[c], 0, 0, 0);
    }
}

// acc[c] += A W^T over the own column tiles, fp32 operands as three bf16 pieces each, six piece pairs (SeqRing16x3)
template <int D, int NCT, class Ring>
__device__ __forceinline__ void part_mma16x6(f32x4 (&acc)[NCT], const StripRegs<D>& A, Ring& ring, int c0) {
    constexpr int KS = D / 32;
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int rowo = (c0 * 16 + i) * (D / 2);                          // a row = D bf16 = D / 2 floats
    auto frag = [&](const float* plane, int c, int s) { return lds_ld4(plane + rowo + c * 16 * (D / 2) + 4 * ((4 * s + g) ^ i)); };
    auto mma = [&](const float4& wf, const amid_v4u& a16, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(seqn_bf16x8, wf), __builtin_bit_cast(seqn_bf16x8, a16), c, 0, 0, 0);
    };
    const float* mbuf = ring.mslot();
    const float* lbuf = ring.lslot();
    // (fragments are read one column tile ahead, not a k-step's worth at once: the kernel sits at the register limit, and a build that
    // held four tiles' fragments beside the three pieces spilled -- 16.5 k cycles for the last product instead of 8.9 k)
    // pass 1: the lo plane against the operand's hi piece, the mid plane against (mid, hi)
    float4 wm = frag(mbuf, 0, 0), wl = frag(lbuf, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const WgSplit2 p0 = wg_split3(A.v[2 * s][0], A.v[2 * s][1]), p1 = wg_split3(A.v[2 * s][2], A.v[2 * s][3]);
        const WgSplit2 p2 = wg_split3(A.v[2 * s + 1][0], A.v[2 * s + 1][1]), p3 = wg_split3(A.v[2 * s + 1][2], A.v[2 * s + 1][3]);
        const amid_v4u ah = amid_v4u{p0.hi, p1.hi, p2.hi, p3.hi}, am = amid_v4u{p0.mid, p1.mid, p2.mid, p3.mid};
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const float4 cm = wm, cl = wl;
            const int cn = c + 1 < NCT ? c + 1 : 0, sn = c + 1 < NCT ? s : s + 1;
            if (sn < KS) { wm = frag(mbuf, cn, sn); wl = frag(lbuf, cn, sn); }
            acc[c] = mma(cl, ah, acc[c]); acc[c] = mma(cm, am, acc[c]); acc[c] = mma(cm, ah, acc[c]);
        }
        __builtin_amdgcn_sched_barrier(0);                 // (keeps a k-step's pieces and fragments from being hoisted over the previous one's)
    }
    const float* hbuf = ring.hcur();
    ring.mid_sync();
    // pass 2: the hi plane against the operand's three pieces
    float4 wf = frag(hbuf, 0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const WgSplit2 p0 = wg_split3(A.v[2 * s][0], A.v[2 * s][1]), p1 = wg_split3(A.v[2 * s][2], A.v[2 * s][3]);
        const WgSplit2 p2 = wg_split3(A.v[2 * s + 1][0], A.v[2 * s + 1][1]), p3 = wg_split3(A.v[2 * s + 1][2], A.v[2 * s + 1][3]);
        const amid_v4u ah = amid_v4u{p0.hi, p1.hi, p2.hi, p3.hi}, am = amid_v4u{p0.mid, p1.mid, p2.mid, p3.mid},
                       al = amid_v4u{p0.lo, p1.lo, p2.lo, p3.lo};
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const float4 cf = wf;
            const int cn = c + 1 < NCT ? c + 1 : 0, sn = c + 1 < NCT ? s : s + 1;
            if (sn < KS) wf = frag(hbuf, cn, sn);
            acc[c] = mma(cf, al, acc[c]); acc[c] = mma(cf, am, acc[c]); acc[c] = mma(cf, ah, acc[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- the own columns of a strip ---------------------------------------------------------------------------------------------------
template <int NCT> struct PartRegs { f32x4 v[NCT]; };

template <int NCT>
__device__ __forceinline__ void part_load(PartRegs<NCT>& x, const GBuf& g, unsigned off_own) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) x.v[c] = g.load4(off_own + c * 64);
}
template <int NCT>
__device__ __forceinline__ void part_store(const GBuf& g, unsigned off_own, const PartRegs<NCT>& x) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) g.store4(off_own + c * 64, x.v[c]);
}
// one column tile per call, from inside an MFMA loop: tensor `x` leaves in groups j == phase of the first NCT k tiles
template <int NCT>
__device__ __forceinline__ void part_spread(const GBuf& g, unsigned off_own, const PartRegs<NCT>& x, int ct, int j, int phase) {
    if (ct < NCT && j == phase) g.store4(off_own + ct * 64, x.v[ct]);
}
template <int NCT>
__device__ __forceinline__ void part_cols(PartRegs<NCT>& v, const float* __restrict__ p, int c0) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) v.v[c] = col4(p, c0 + c);
}

// exchange: own parts -> xb[column tile][lane]; after a barrier every wave of the strip reads all D / 16 tiles
template <int NCT>
__device__ __forceinline__ void xchg_write(float* __restrict__ xb, int c0, const PartRegs<NCT>& x) {
    const int lane = lane_id();
#pragma unroll
    for (int c = 0; c < NCT; ++c) lds_st4(xb + ((c0 + c) * 64 + lane) * 4, x.v[c]);
}
template <int D>
__device__ __forceinline__ void xchg_read(StripRegs<D>& full, const float* __restrict__ xb) {
    const int lane = lane_id();
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
        const float4 t = lds_ld4(xb + (ct * 64 + lane) * 4);
        full.v[ct] = f32x4{t.x, t.y, t.z, t.w};
    }
}

// acc[c] += sum_k A[.][k] W[(c0 + c) * 16 + .][k]: strip_mma for the own column tiles.  4 groups per k tile (one per element r of the
// operand quad): NCT MFMAs on different accumulators + one fragment read of the next k tile; hook(ct, j) behind group j.
template <int D, int NCT, class Hook>
__device__ __forceinline__ void part_mma(f32x4 (&acc)[NCT], const StripRegs<D>& A, const float* __restrict__ buf, int c0, const Hook& hook) {
    constexpr int NT = D / 16;
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int xl = g ^ i;
    const float* rowp = buf + (c0 * 16 + i) * D;
    float4 wf[2][NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) wf[0][c] = lds_ld4(rowp + c * 16 * D + 4 * xl);
    AMID_STRIP_FENCE();
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const float* nxt = rowp + 4 * (((ct + 1) * 4) ^ xl);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int c = 0; c < NCT; ++c) {
                const float4 w = wf[ct & 1][c];
                const float wr = j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w;
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, A.v[ct][j], acc[c], 0, 0, 0);
            }
            // the next k tile's fragments: two behind group 0, one behind groups 1 and 2 -- the last one has a whole group (128 cycles)
            // + three MFMAs in front of its use instead of three MFMAs
            if (ct + 1 < NT) {
                if (j == 0) {
                    wf[(ct + 1) & 1][0] = lds_ld4(nxt);
                    if (NCT > 1) wf[(ct + 1) & 1][1] = lds_ld4(nxt + 16 * D);
                } else if (j + 1 < NCT) {
                    wf[(ct + 1) & 1][j + 1] = lds_ld4(nxt + (j + 1) * 16 * D);
                }
            }
            hook(ct, j);
            AMID_STRIP_FENCE();
        }
    }
}

// keep multipliers of the own columns of row `local` at `site` (p = 0.5: the row's ONE Philox call, requested ahead: `rr`)
// (f0: the row's first field inside the call -- 0 at D = 128, where a call is a row; 0 or 64 at D = 64, two rows per call)
template <int NCT>
__device__ __forceinline__ void part_dropout(PartRegs<NCT>& x, const uint4 rr, int c0, unsigned spec, float scale, int f0 = 0) {
    const int g4 = 4 * (lane_id() >> 4);
    const bool all = spec_thr(spec) == 0;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const int f = f0 + (c0 + c) * 16 + g4;
        const unsigned wlo = (f & 64) ? rr.z : rr.x, whi = (f & 64) ? rr.w : rr.y;
        const unsigned w = ((f & 32) ? whi : wlo) >> (f & 31);
#pragma unroll
        for (int r = 0; r < 4; ++r) x.v[c][r] = (all || ((w >> r) & 1u)) ? x.v[c][r] * scale : 0.f;
    }
}

// one product of the chain on the own column tiles: fp32 -- strip MFMA loop with the next slab's DMA pieces and the deferred stores
// (`stores(ct, j)`) in its groups; bf16 -- the next slab requested up front, 16 MFMAs, the deferred stores behind them
struct NoLate { __device__ __forceinline__ void operator()() const {} };
// `late()` runs half way through the loop: loads the epilogue needs (LayerNorm gains: 64 registers) are requested there --
// early enough to land under the remaining MFMAs, late enough not to be carried through the whole loop (vector-memory instructions do
// not cross the loop's scheduling fences)
// ZERO = false: the product is added to what `acc` holds (the MFMA chain goes on: same bits as one product over the joined k range)
template <int D, int NCT, bool BF, bool ZERO = true, class Ring, class Stores, class Late = NoLate>
__device__ __forceinline__ void seqn_product(f32x4 (&acc)[NCT], const StripRegs<D>& A, const float* __restrict__ buf, Ring& ring,
                                             const float* __restrict__ wn32, const unsigned short* __restrict__ wn16, int c0, const Stores& stores,
                                             const Late& late = NoLate()) {
    if constexpr (ZERO) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (std::is_same<Ring, SeqRing16x3<D, Ring::NWAVES>>::value) {
        ring.begin(wn16);
        late();
        part_mma16x6<D, NCT>(acc, A, ring, c0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { stores(ct, 1); stores(ct, 3); }
    } else if constexpr (BF) {
        ring.fetch_all(wn16);
        late();
        part_mma16<D, NCT>(acc, A, buf, c0);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) { stores(ct, 1); stores(ct, 3); }
    } else {
        part_mma<D, NCT>(acc, A, buf, c0, [&](int ct, int j) {
            ring.fetch(wn32, ct, j);
            stores(ct, j);
            if (ct == D / 16 / 2 && j == 0) late();
        });
    }
}

// ---- the same products with the operand's pieces made ONCE, by the wave that produced the operand (seqn_fwd_px_kernel) ----------------
// In SeqRing16x3's build every wave splits the whole row of its strip into pieces itself, in both passes: eight splits of eight values
// per product, more vector issue slots than the 96 matrix instructions leave free, and a 32-register operand on top of the pieces.
// Here the exchange between the column parts of a strip carries bf16 PIECES: a wave splits its own D / NS columns once per product and
// writes whole operand fragments ([strip][k-step][piece][lane] x 16 bytes: the eight values a lane supplies in a k-step are its elements
// of column tiles 2 s and 2 s + 1), the products read fragments -- no split, no whole-row operand in registers.  48 KB of exchange
// (WPS = 4) beside THREE 32 KB plane slots [M][L][H]: with one H slot a product's hi plane is requested when the product begins and
// awaited behind its first pass; its mid / lo planes were requested behind the previous product's first pass.
template <int D, int NW> struct SeqRing3 {
    static constexpr int NWAVES = NW;
    static constexpr int CPR = D / 8;
    static constexpr int PIECES = D * CPR / 64, PER_WAVE = PIECES / NW;
    static constexpr int SLAB = D * D / 2;
    float* buf; unsigned off0; int w;
    const unsigned short* cur; const unsigned short* nxt;
    bool hi_ready, rest_ready, rest_late, hold_next;
    bool one;                            // ONE piece per operand (compute = "bf16" on the folded step): only the hi planes are streamed
    __device__ __forceinline__ explicit SeqRing3(float* lds)
        : buf(lds), cur(nullptr), nxt(nullptr), hi_ready(false), rest_ready(false), rest_late(false), hold_next(false), one(false) {
        w = wave_id();
        const int p = w * 64 + lane_id();
        const int n = p / CPR, pos = p % CPR;
        off0 = (unsigned)(n * D * 2 + ((pos ^ (n & 15)) * 16));
    }
    __device__ __forceinline__ float* mslot() const { return buf; }
    __device__ __forceinline__ float* lslot() const { return buf + SLAB; }
    __device__ __forceinline__ float* hslot() const { return buf + 2 * SLAB; }
    __device__ __forceinline__ float* images() const { return buf; }                 // M + L
    __device__ __forceinline__ void piece(float* __restrict__ dst, const unsigned short* __restrict__ W, int k0) const {
        const unsigned voff = off0 + (unsigned)k0 * (unsigned)(NW * 64 / CPR) * (unsigned)(D * 2);
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(dst + (k0 * NW + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
    __device__ __forceinline__ void plane(float* __restrict__ dst, const unsigned short* __restrict__ Wp) const {
#pragma unroll
        for (int k0 = 0; k0 < PER_WAVE; ++k0) piece(dst, Wp, k0);
    }
    __device__ __forceinline__ void first(const unsigned short* __restrict__ W0) {
        cur = W0; hi_ready = true; rest_ready = true;
        plane(hslot(), W0);
        if (!one) { plane(mslot(), W0 + (size_t)D * D); plane(lslot(), W0 + (size_t)2 * D * D); }
    }
    // Round 5: a product walks its HI plane first.  Schedule of product i (weight cur):
    //   next()      everything requested so far has landed and every wave is past product i - 1 (and past the attention images, which live in
    //               M + L between the q product and the out-projection): the mid / lo planes of cur are requested NOW and land under the
    //               hi pass; the hi plane was requested half a product ago (mid_sync of product i - 1) and is here
    //   pass H      hi plane x the operand's (lo, mid, hi)
    //   mid_sync()  mid / lo have landed, every wave is done with H: the NEXT weight's hi plane is requested into H and lands under the
    //   pass M/L    lo x hi, mid x (mid, hi)
    // Round 4 walked mid / lo first: the product behind the attention core then found its mid / lo planes only requested when it began (M + L
    // held the images) and sat out the whole DMA -- 2.7 - 9 k cycles per layer (profiles/r04_seqn_stamps.txt "stats + o parts out + ring
    // wait").  With the hi plane first that product's first pass runs on a plane requested before the attention core.
    __device__ __forceinline__ void next() {
        w_ring_wait();
        __syncthreads();
        if (!hi_ready) plane(hslot(), cur);                  // (never in steady state: kept for a ring that is started without first())
        rest_late = !hi_ready;
        if (!rest_ready && !one) { plane(mslot(), cur + (size_t)D * D); plane(lslot(), cur + (size_t)2 * D * D); }
        hold_next = false;
    }
    __device__ __forceinline__ void begin(const unsigned short* __restrict__ Wnext) { nxt = Wnext; }
    __device__ __forceinline__ void pre_pass1() {            // (only when the hi plane was requested by next() itself)
        if (rest_late) { w_ring_wait(); __syncthreads(); }
    }
    __device__ __forceinline__ void mid_sync() {             // behind the hi pass: mid / lo have landed; H takes the next weight's hi plane
#ifndef AMID_XNOWAIT
        w_ring_wait();
#endif
        __syncthreads();
        plane(hslot(), nxt);
        cur = nxt; hi_ready = true; rest_ready = false;
    }
};

// a strip's exchange slots of pieces: [k-step][piece][lane] x 16 bytes
template <int D> struct XpStrip { static constexpr int FLOATS = (D / 32) * 3 * 64 * 4; };
template <int NCT>
__device__ __forceinline__ void xp_write(float* __restrict__ xps, int c0, const PartRegs<NCT>& x) {
    static_assert(NCT % 2 == 0, "a wave owns whole k-steps (pairs of column tiles)");
    const int lane = lane_id();
#pragma unroll
    for (int e = 0; e < NCT / 2; ++e) {
        const int s = c0 / 2 + e;
        const WgSplit2 p0 = wg_split3(x.v[2 * e][0], x.v[2 * e][1]), p1 = wg_split3(x.v[2 * e][2], x.v[2 * e][3]);
        const WgSplit2 p2 = wg_split3(x.v[2 * e + 1][0], x.v[2 * e + 1][1]), p3 = wg_split3(x.v[2 * e + 1][2], x.v[2 * e + 1][3]);
        lds_st4(xps + ((s * 3 + 0) * 64 + lane) * 4, __builtin_bit_cast(f32x4, amid_v4u{p0.hi, p1.hi, p2.hi, p3.hi}));
        lds_st4(xps + ((s * 3 + 1) * 64 + lane) * 4, __builtin_bit_cast(f32x4, amid_v4u{p0.mid, p1.mid, p2.mid, p3.mid}));
        lds_st4(xps + ((s * 3 + 2) * 64 + lane) * 4, __builtin_bit_cast(f32x4, amid_v4u{p0.lo, p1.lo, p2.lo, p3.lo}));
    }
}
__device__ __forceinline__ amid_v4u xp_frag(const float* __restrict__ xps, int s, int p) {
    const float4 t = lds_ld4(xps + ((s * 3 + p) * 64 + lane_id()) * 4);
    return __builtin_bit_cast(amid_v4u, t);
}
// the strip's whole row back in fp32 (hi + mid + lo is the value exactly): LayerNorm statistics
template <int D>
__device__ __forceinline__ void xp_row(StripRegs<D>& full, const float* __restrict__ xps) {
#pragma unroll
    for (int s = 0; s < D / 32; ++s) {
        const amid_v4u h = xp_frag(xps, s, 0), m = xp_frag(xps, s, 1), l = xp_frag(xps, s, 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = (__uint_as_float(h[j] << 16) + __uint_as_float(m[j] << 16)) + __uint_as_float(l[j] << 16);
            const float hi = (__uint_as_float(h[j] & 0xffff0000u) + __uint_as_float(m[j] & 0xffff0000u)) + __uint_as_float(l[j] & 0xffff0000u);
            full.v[2 * s + (j >> 1)][2 * (j & 1)] = lo;
            full.v[2 * s + (j >> 1)][2 * (j & 1) + 1] = hi;
        }
    }
}

// acc[c] += A W^T over the own column tiles: operand fragments from the strip's exchange slots, weight planes from SeqRing3.
// The fragment reads are placed by hand (strip_gemm.h lds_frag_issue / lds_frag_wait): left to the compiler every read sat just in front
// of its first use (lgkmcnt(1) / (0) in front of most matrix instructions).  A step = one column tile of one k-step: two weight fragments
// (pass 1: mid, lo planes) or one (pass 2: hi plane), three matrix instructions.  Weight reads run PD steps ahead; LDS operations return
// in order, so a step waits until only the reads issued behind its own are outstanding.
// ONE: the hi plane against the operand's hi piece only -- bf16 products (operands rounded to nearest even), a sixth of the matrix instructions
template <int D, int NCT, class Ring, bool ONE = false>
__device__ __forceinline__ void part_mma_xp(f32x4 (&acc)[NCT], const float* __restrict__ xps, Ring& ring, int c0) {
    constexpr int KS = D / 32, NSTEP = KS * NCT, CT_BYTES = 16 * (D / 2) * 4, PLANE_BYTES = Ring::SLAB * 4;
    constexpr int PD1 = FRAG_AHEAD_2, PD2 = FRAG_AHEAD_1;
    static_assert(PD1 < NSTEP && PD2 < NSTEP && 2 * PD1 + 2 < 16, "read-ahead against the step count and the lgkmcnt field");
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    auto lds_addr = [](const float* p) { return lds_offset(p); };
    auto mma = [&](const f32x4& wf, const f32x4& a16, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(seqn_bf16x8, wf), __builtin_bit_cast(seqn_bf16x8, a16), c, 0, 0, 0);
    };
    unsigned fo[KS];                                          // this lane's fragment of k-step s in a plane, own column tile 0
#pragma unroll
    for (int s = 0; s < KS; ++s) fo[s] = (unsigned)((((c0 * 16 + i) * (D / 2)) + 4 * ((4 * s + g) ^ i)) * 4);
    const unsigned xa = lds_addr(xps) + (unsigned)lane * 16u;  // operand fragments: [k-step][piece][lane] x 16 bytes
    const unsigned mbase = lds_addr(ring.mslot());
    static_assert(CT_BYTES * (NCT - 1) + PLANE_BYTES < 65536, "the lo plane is reached through the offset field");
    const unsigned hbase = mbase + 2u * PLANE_BYTES;
    ring.pre_pass1();
    // a step = (own column tile c, k-step s), s fastest: a column tile's twelve matrix instructions per pass run back to back on its
    // accumulator (profiles/r04_mfma_rate_probe.txt: a chain on one accumulator issues faster than instructions that change accumulator;
    // 77.7 -> 76.1 us by HIP events on one box); the operand's fragments of all four k-steps are read up front
    {   // first pass: the hi plane against the operand's three pieces (the plane was requested half a product ago)
        f32x4 wf[PD2 + 1], ah[KS], am[KS], al[KS];
        static_for<KS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            lds_frag_issue<(s * 3 + 0) * 1024>(ah[s], xa);
            lds_frag_issue<(s * 3 + 1) * 1024>(am[s], xa);
            lds_frag_issue<(s * 3 + 2) * 1024>(al[s], xa);
        });
        auto issue = [&](auto U) {
            constexpr int u = decltype(U)::value, c = u / KS, s = u % KS;
            lds_frag_issue<c * CT_BYTES>(wf[u % (PD2 + 1)], hbase + fo[s]);
        };
        static_for<PD2>(issue);
        static_for<NSTEP>([&](auto T) {
            constexpr int t = decltype(T)::value, c = t / KS, s = t % KS, k = t % (PD2 + 1);
            if constexpr (t + PD2 < NSTEP) issue(std::integral_constant<int, t + PD2>{});
            constexpr int last = t + PD2 < NSTEP ? t + PD2 : NSTEP - 1;
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wf[k]), "+v"(ah[s]), "+v"(am[s]), "+v"(al[s]) : "n"(last - t));
            if constexpr (!ONE) { acc[c] = mma(wf[k], al[s], acc[c]); acc[c] = mma(wf[k], am[s], acc[c]); }
            acc[c] = mma(wf[k], ah[s], acc[c]);
        });
    }
    ring.mid_sync();
    if constexpr (!ONE) {   // second pass: the lo plane against the operand's hi piece, the mid plane against (mid, hi)
        f32x4 wm[PD1 + 1], wl[PD1 + 1], ah[KS], am[KS];
        static_for<KS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            lds_frag_issue<(s * 3 + 0) * 1024>(ah[s], xa);
            lds_frag_issue<(s * 3 + 1) * 1024>(am[s], xa);
        });
        auto issue = [&](auto U) {
            constexpr int u = decltype(U)::value, c = u / KS, s = u % KS, k = u % (PD1 + 1);
            lds_frag_issue<c * CT_BYTES>(wm[k], mbase + fo[s]);
            lds_frag_issue<c * CT_BYTES + PLANE_BYTES>(wl[k], mbase + fo[s]);
        };
        static_for<PD1>(issue);
        static_for<NSTEP>([&](auto T) {
            constexpr int t = decltype(T)::value, c = t / KS, s = t % KS, k = t % (PD1 + 1);
            if constexpr (t + PD1 < NSTEP) issue(std::integral_constant<int, t + PD1>{});
            constexpr int last = t + PD1 < NSTEP ? t + PD1 : NSTEP - 1;
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(wm[k]), "+v"(wl[k]), "+v"(ah[s]), "+v"(am[s]) : "n"(2 * (last - t)));
            acc[c] = mma(wl[k], ah[s], acc[c]); acc[c] = mma(wm[k], am[s], acc[c]); acc[c] = mma(wm[k], ah[s], acc[c]);
        });
    }
}

// one product of seqn_fwd_px_kernel: the next weight is announced, the deferred stores leave behind the matrix instructions
template <int D, int NCT, bool ONE = false, class Ring, class Stores>
__device__ __forceinline__ void seqn_product_xp(f32x4 (&acc)[NCT], const float* __restrict__ xps, Ring& ring, const unsigned short* __restrict__ wn16,
                                                int c0, const Stores& stores) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    ring.begin(wn16);
#ifndef AMID_EXP_STORES_LAST
    // the deferred stores (an earlier product's result) leave IN FRONT of the matrix instructions: the ring's waits are vmcnt(0), and stores
    // issued behind the product were the youngest operations every next() then sat out (a store's acknowledgement takes 1 - 2 k cycles);
    // issued here they have the whole hi pass to complete before mid_sync's wait
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { stores(ct, 1); stores(ct, 3); }
#endif
    part_mma_xp<D, NCT, Ring, ONE>(acc, xps, ring, c0);
#ifdef AMID_EXP_STORES_LAST
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { stores(ct, 1); stores(ct, 3); }
#endif
}

// the own column tiles of a per-column vector held whole: two parts -- selects on the wave-uniform part index; more parts -- loaded
// (select chains over four or eight candidates end up as scratch arrays)
template <int D, int NCT>
__device__ __forceinline__ void own_cols(PartRegs<NCT>& o, const ColVec<D>& full, const float* __restrict__ p, int part, int c0) {
    if constexpr ((D / 16) / NCT == 2) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) o.v[c] = part ? full.v[NCT + c] : full.v[c];
    } else {
        part_cols<NCT>(o, p, c0);
    }
}

}  // namespace amid
