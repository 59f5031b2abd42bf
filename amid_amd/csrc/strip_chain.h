// The shared scaffolding of the strip kernels (sasrec_strip.hip: the SASRec layer; bert_strip.hip: the BERT4Rec block): the two-slab weight
// ring over strip_gemm.h's LDS-DMA, the deferred stores, the LayerNorm-gradient partial sums of a 64-row tile, the tile geometry and
// the launch helper.
#pragma once
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"
#include <type_traits>

namespace amid {

template <int D>
__device__ __forceinline__ void add_bias(f32x4 (&acc)[D / 16], const ColVec<D>& b) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) acc[ct] += b.v[ct];
}
template <int D>
__device__ __forceinline__ void to_regs(StripRegs<D>& dst, const f32x4 (&acc)[D / 16]) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) dst.v[ct] = acc[ct];
}

// a running two-slab ring.  next() returns the slab whose fetch was started one slab ago: it waits for this wave's DMAs, then meets
// the other waves at the workgroup barrier -- the slab has landed for everybody, and everybody is done reading the OTHER buffer,
// which the pieces issued from inside the coming MFMA loop (fetch()) overwrite.
template <int D, bool BF = false> struct Ring {
    static constexpr bool BF16 = BF;                    // bf16 fragment images (strip_gemm.h WDma16): slabs of D D / 2 floats
    static constexpr int SLAB = BF ? D * D / 2 : D * D;
    using Dma = typename std::conditional<BF, WDma16<D>, WDma<D>>::type;
    float* buf; int s; Dma dma;
    __device__ __forceinline__ explicit Ring(float* lds) : buf(lds), s(0) {}
    __device__ __forceinline__ void first(const float* __restrict__ W0) { dma.all(buf, W0); }
    __device__ __forceinline__ const float* next() {
        w_ring_wait();
        __syncthreads();
        const float* cur = buf + (s & 1) * SLAB;
        ++s;
        return cur;
    }
    // group (ct, j) of the current MFMA loop: this wave's share of slab W's DMA, one piece every few groups (address arithmetic and
    // issue slide under the matrix work instead of standing in front of the loop)
    __device__ __forceinline__ void fetch(const float* __restrict__ W, int ct, int j) const {
        // all of them in the FIRST half of the loop: a piece takes a couple of thousand cycles to land, and the next slab starts
        // with a wait for every one of them
        constexpr int SLOTS = 8 * (D / 16), EVERY = (SLOTS / 2) / Dma::PER_WAVE;
        const int slot = ct * 8 + j;
        if (slot % EVERY == 0 && slot / EVERY < Dma::PER_WAVE) dma.piece(buf + (s & 1) * SLAB, W, slot / EVERY);
    }
};

// The ring of the products on bf16 pieces (strip_gemm.h strip_mma16x6): a weight is three 32 KB fragment images (planes hi, mid, lo,
// [3][D][D] bf16 behind one pointer), two whole weights do not fit the ring's 128 KB: four plane slots [M][L][H0][H1], every plane
// requested well ahead of its pass -- the next weight's hi plane when a product begins (the other H slot is free then), its mid / lo
// planes behind the product's barrier (M and L are free then).  The next weight is whatever the product's hooks fetch().
template <int D> struct RingP3 {
    static constexpr bool BF16 = true, P3 = true;
    static constexpr int SLAB = D * D / 2;
    float* buf; int s; WDma16<D> dma;
    const float* cur; const float* nxt;
    bool cur_hi_ready, hi_late, have_next;
    __device__ __forceinline__ explicit RingP3(float* lds) : buf(lds), s(0), cur(nullptr), nxt(nullptr), cur_hi_ready(false), hi_late(false), have_next(false) {}
    __device__ __forceinline__ float* mslot() const { return buf; }
    __device__ __forceinline__ float* lslot() const { return buf + SLAB; }
    __device__ __forceinline__ float* hslot(int k) const { return buf + (2 + (k & 1)) * SLAB; }
    __device__ __forceinline__ const float* hcur() const { return hslot(s - 1); }
    __device__ __forceinline__ void first(const float* __restrict__ W0) {
        cur = W0; cur_hi_ready = true;
        dma.all(hslot(0), W0); dma.all(mslot(), W0 + SLAB); dma.all(lslot(), W0 + 2 * SLAB);
    }
    __device__ __forceinline__ const float* next() {
        w_ring_wait();
        __syncthreads();
        ++s;
        have_next = false;
        return hslot(s - 1);
    }
    __device__ __forceinline__ void fetch(const float* __restrict__ W, int ct, int j) { if (ct == 0 && j == 0) { nxt = W; have_next = true; } }
    __device__ __forceinline__ void begin() {
        hi_late = !cur_hi_ready;
        if (hi_late) dma.all(hslot(s - 1), cur);
        if (have_next) dma.all(hslot(s), nxt);
    }
    __device__ __forceinline__ void mid_sync() {
        if (hi_late) w_ring_wait();
        __syncthreads();
        if (have_next) { dma.all(mslot(), nxt + SLAB); dma.all(lslot(), nxt + 2 * SLAB); cur = nxt; cur_hi_ready = true; }
        else cur_hi_ready = false;
    }
};
// MODE 0: fp32 matrix instructions, 1: operands rounded to bf16, 3: fp32 operands as three bf16 pieces
template <int D, int MODE> struct RingSel { using type = Ring<D, MODE != 0>; };
template <int D> struct RingSel<D, 3> { using type = RingP3<D>; };
template <class R> struct ring_is_p3 { static constexpr bool value = false; };
template <int D> struct ring_is_p3<RingP3<D>> { static constexpr bool value = true; };
// one product of a chain on whichever ring the kernel was built with
template <int D, bool SPREAD = false, class RingT, class Hook>
__device__ __forceinline__ void strip_product(f32x4 (&acc)[D / 16], const StripRegs<D>& A, const float* __restrict__ buf, RingT& ring, const Hook& hook) {
    if constexpr (ring_is_p3<RingT>::value) strip_mma16x6<D, RingT, Hook, SPREAD>(acc, A, ring, hook);
    else strip_mma_sel<D, RingT::BF16>(acc, A, buf, hook);
}

// stores of a finished strip leave under the FIRST half of the next MFMA loop, one column tile every fourth group: by the end of the
// loop they have long been acknowledged, so the ring's vmcnt(0) in front of the next slab costs nothing
template <int D>
__device__ __forceinline__ void store_spread(const GBuf& g, const StripRow& row, const StripRegs<D>& x, int ct, int j) {
    constexpr int NT = D / 16;
    if (ct < NT / 2 && (j & 3) == 1) strip_store_ct<D>(g, row, x, 2 * ct + (j >> 2));
}

// column sums of the strip's 16 rows (DPP inside each row of 16 lanes) -> this wave's slice of the LDS scratch [4 waves][2][D]
template <int D>
__device__ __forceinline__ void ln_partials_wave(float* __restrict__ scratch, const StripRegs<D>& dgam, const StripRegs<D>& dbet) {
    const int lane = lane_id(), w = wave_id();
    float* mine = scratch + w * 2 * D;
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
        f32x4 a, b;
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[r] = col_sum16(dgam.v[ct][r]); b[r] = col_sum16(dbet.v[ct][r]); }
        if ((lane & 15) == 0) {
            st4(mine + ct * 16 + 4 * (lane >> 4), make_float4(a[0], a[1], a[2], a[3]));
            st4(mine + D + ct * 16 + 4 * (lane >> 4), make_float4(b[0], b[1], b[2], b[3]));
        }
    }
}
// after a workgroup barrier: the four waves' slices -> part[2][D] in global memory (fixed order)
template <int D>
__device__ __forceinline__ void ln_partials_out(const float* __restrict__ scratch, float* __restrict__ part) {
    for (int e = threadIdx.x; e < 2 * D; e += STRIP_THREADS)
        part[e] = (scratch[e] + scratch[2 * D + e]) + (scratch[4 * D + e] + scratch[6 * D + e]);
}

// LDS: [ring: 2 slabs][LayerNorm-partial scratch A: 4 x 2 x D][scratch B: 4 x 2 x D]
template <int D> __device__ __forceinline__ float* ln_scratch(float* smem, int which) { return smem + 2 * D * D + which * 8 * D; }

template <int D>
__device__ __forceinline__ void zero_slot(float* __restrict__ part, int slot) {
    for (int e = threadIdx.x; e < 2 * D; e += STRIP_THREADS) part[(long long)slot * 2 * D + e] = 0.f;
}


}  // namespace amid

// ---- host side ------------------------------------------------------------------------------------------------------------------------
namespace amid_strip_host {
using namespace amid;

template <int D> static constexpr size_t strip_lds_bytes() { return (size_t)(2 * D * D + 16 * D) * sizeof(float); }

static int make_strip_geom(int B, int T, int D, const int* live, StripGeom* sg) {
    if (B <= 0 || T <= 0) return AMID_ERR_ARG;
    const long long bytes = 2LL * B * T * D * 4;
    if (bytes > 0x7FFFFFF0LL) return AMID_ERR_UNSUPPORTED;          // buffer descriptors: 32-bit offsets, out-of-range marker at 2 GiB
    sg->B = B; sg->T = T; sg->M = B * T;
    sg->act_bytes = (unsigned)bytes; sg->tm_bytes = (unsigned)(bytes / 16);
    sg->tpg = (sg->M + STRIP_TILE - 1) / STRIP_TILE;
    sg->live = live;
    return AMID_OK;
}

template <auto KERNEL, int DVAL, class... Args>
static int launch_strip(const StripGeom& sg, void* stream, const Args&... args) {
    static unsigned long long attr_done = 0;
    if (int rc = lds_attr_once((const void*)KERNEL, strip_lds_bytes<DVAL>(), attr_done)) return rc;
    KERNEL<<<2 * sg.tpg, STRIP_THREADS, strip_lds_bytes<DVAL>(), (hipStream_t)stream>>>(args..., sg);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? AMID_OK : (int)e;
}

}  // namespace amid_strip_host
