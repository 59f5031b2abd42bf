// Register-resident "strip" GEMM chains for the SASRec encoder on gfx950 (sasrec_strip.hip).
//
// The row-tile kernels of tile_gemm.h stage every activation tile through LDS between two chained GEMMs (accumulators -> C image ->
// row pass -> A image -> next MFMA loop), with three workgroup barriers per GEMM and the matrix pipe idle in every phase that is not
// the MFMA loop: 0.25-0.44 of the fp32 matrix peak (profiles/r01h_*).  Here an activation never leaves the register file between
// two GEMMs of a chain:
//
//   * a WAVE owns a strip of 16 rows and ALL D columns.  Lane (m, g) = (lane & 15, lane >> 4) holds row m of the strip, and its
//     register v[ct][r] holds column ct * 16 + g * 4 + r  (ct < D / 16, r < 4)  -- the "C layout": exactly what
//     v_mfma_f32_16x16x4_f32 issued as C^T = W . A^T leaves in the accumulators (reg r of lane (m, g) = C[m][ct * 16 + g * 4 + r]).
//   * the SAME registers are the second operand of the next GEMM: an MFMA step sums over 4 values of k, one per lane group g, and
//     a sum over k does not care which k goes where as long as both operands agree -- step (ct, r) takes k = ct * 16 + g * 4 + r,
//     i.e. register v[ct][r] as it stands, and the weight operand of lane (n, g) is W[n][ct * 16 + g * 4 + r]: four consecutive
//     floats of a K-contiguous weight row, one ds_read_b128 per four MFMAs.
//   * so bias / dropout / relu / residual / LayerNorm / masks run on the accumulators in place (a row's LayerNorm statistic is
//     an in-lane sum of 32 values + two cross-lane steps over the 4 lanes of the row), and the result IS the next operand.  No
//     A image, no C image, no barrier inside a chain; LDS only holds the weights.
//   * weights stream through a two-slab LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no staging registers): slab s + 1
//     lands while slab s multiplies; ONE workgroup barrier per slab.  The image is [n][D] floats without padding; bank conflicts
//     are removed by an XOR swizzle of the 16-byte chunk index with n & 15, applied on the DMA's per-lane SOURCE address (the DMA
//     writes LDS linearly).
//
// A workgroup is 4 waves (one per SIMD, up to 512 registers each) = 64 consecutive rows of one domain; rows are independent, so
// tiles need not align with sequences.  In the train step only the LIVE sequences are walked (the loss multiplies the other
// domain's terms of every sample by zero, train_sr.py:205-211): tiles cover "virtual" rows -- the live sequences of a domain back
// to back -- and a lane maps its virtual row to the row of the [2, B, T] activation layout once per kernel (StripRow).
#pragma once
#include "common.h"
#include "bf16_pieces.h"
#include "rng.h"
#include <type_traits>
#include <utility>

#define STRIP_STAMP_WAVES 4

namespace amid {

// diagnostic builds only (profiles/tools/strip_stamps.py compiles sasrec_strip.hip with -DAMID_STRIP_STAMPS into its own library):
// s_memtime stamps of workgroup 0, one row of 32 per wave, in a buffer no kernel reads
#ifdef AMID_STRIP_STAMPS
static __device__ unsigned long long amid_strip_stamp_buf[STRIP_STAMP_WAVES * 32];      // one per translation unit
#define STRIP_STAMP(i) do { if (blockIdx.x == 0 && lane_id() == 0) amid_strip_stamp_buf[wave_id() * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STRIP_RSTAMP(i) do { if (blockIdx.x == 0 && lane_id() == 0) amid_strip_stamp_buf[wave_id() * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STRIP_STAMP(i) do { } while (0)
#define STRIP_RSTAMP(i) do { } while (0)
#endif

constexpr int STRIP_WAVES = 4;
constexpr int STRIP_THREADS = 64 * STRIP_WAVES;
constexpr int STRIP_TILE = 16 * STRIP_WAVES;        // rows per workgroup

template <int D> struct StripRegs {                 // a [16, D] strip in the C layout
    static constexpr int NT = D / 16;
    f32x4 v[NT];
};

// ---- geometry ------------------------------------------------------------------------------------------------------------------
struct StripGeom {
    int M, B, T;            // rows per domain (B * T)
    unsigned act_bytes;     // bytes of a [2M, D] fp32 activation tensor (buffer descriptors; <= 2 GiB), tm_bytes = act_bytes / 16
    unsigned tm_bytes;
    int tpg;                // tiles per domain in the worst case: ceil(M / STRIP_TILE); the launch has 2 * tpg workgroups
    const int* live;        // optional [B + 1]: batch rows of domain 0's live sequences (ascending), then domain 1's; live[B] = n0.
                            // nullptr: every sequence of both domains
};

struct StripTile { int g, slot, v0, nv, s0; bool live; };     // domain, partial slot, first virtual row, virtual rows of the domain, first entry of the live list

// Workgroup -> (domain, tile) straight from the workgroup index: even workgroups walk domain 0, odd ones domain 1, so the weight
// pointers and the first slab's DMA need no load of the live list (two dependent global loads that used to stand in front of
// everything else); a workgroup whose tile lies past its domain's live rows finds out AFTER it has started the DMA -- it waits for
// the DMA (the LDS it targets dies with the workgroup) and leaves.  With the domains interleaved the live tiles of an even split
// are the first workgroups dispatched.
__device__ __forceinline__ int strip_domain(int tile) { return tile & 1; }
__device__ __forceinline__ StripTile strip_tile(const StripGeom& sg, int tile) {
    StripTile t;
    t.g = tile & 1;
    const int tl = tile >> 1;
    t.slot = t.g * sg.tpg + tl;
    t.v0 = tl * STRIP_TILE;
    t.s0 = 0;
    t.nv = sg.M;
    if (sg.live != nullptr) {
        const int n0 = sg.live[sg.B];
        t.nv = (t.g ? sg.B - n0 : n0) * sg.T;
        t.s0 = t.g ? n0 : 0;
    }
    t.live = t.v0 < t.nv;
    return t;
}

// A [2M, D] fp32 activation tensor seen through a buffer descriptor: per-lane 32-bit byte offsets, and the hardware's bounds check
// drops the accesses of the lanes whose row lies past the domain (their offset is STRIP_OOB): loads return zeros, stores vanish --
// no exec-mask branch anywhere in a chain, so every MFMA loop stays ONE scheduling region.  Tensors are limited to 2 GiB.
constexpr unsigned STRIP_OOB = 0x80000000u;
typedef unsigned amid_v4u __attribute__((ext_vector_type(4)));
struct GBuf {
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ GBuf(const void* p, unsigned bytes) : r(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000)) {}
    __device__ __forceinline__ f32x4 load4(unsigned off) const { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); }
    __device__ __forceinline__ amid_v4u load4u(unsigned off) const { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); }
    __device__ __forceinline__ void store4(unsigned off, f32x4 v) const { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(amid_v4u, v), r, (int)off, 0, 0); }
};

struct StripRow {           // this lane's row
    unsigned off;           // byte offset of the row in a [2M, D] fp32 tensor + this lane's 16 bytes of column tile 0; STRIP_OOB past the domain
    int local;              // row inside the domain (b * T + t): dropout counters
    bool ok;
};
__device__ __forceinline__ float4 ld4_global(const float* p) {
    const amid_gv4 t = *(const __attribute__((address_space(1))) amid_gv4*)(p);
    return make_float4(t.x, t.y, t.z, t.w);
}

template <int D>
__device__ __forceinline__ StripRow strip_row(const StripGeom& sg, const StripTile& t) {
    StripRow r;
    int v = t.v0 + wave_id() * 16 + (lane_id() & 15);
    r.ok = v < t.nv;
    if (!r.ok) v = t.v0;                            // a live tile's first row always exists
    if (sg.live != nullptr) {
        const int s = v / sg.T;
        r.local = sg.live[t.s0 + s] * sg.T + (v - s * sg.T);
    } else {
        r.local = v;
    }
    const unsigned phys = (unsigned)t.g * (unsigned)sg.M + (unsigned)r.local;
    r.off = r.ok ? phys * (unsigned)(D * 4) + 16u * (unsigned)(lane_id() >> 4) : STRIP_OOB;
    return r;
}

// ---- strip <-> global ---------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void strip_load(StripRegs<D>& x, const GBuf& g, const StripRow& row) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) x.v[ct] = g.load4(row.off + ct * 64);
}
template <int D>
__device__ __forceinline__ void strip_store(const GBuf& g, const StripRow& row, const StripRegs<D>& x) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) g.store4(row.off + ct * 64, x.v[ct]);
}
// one column tile of the strip (deferred stores inside an MFMA loop)
template <int D>
__device__ __forceinline__ void strip_store_ct(const GBuf& g, const StripRow& row, const StripRegs<D>& x, int ct) {
    g.store4(row.off + ct * 64, x.v[ct]);
}

// a per-column vector (bias, LayerNorm gain): the lane's 4 columns of tile ct
__device__ __forceinline__ f32x4 col4(const float* __restrict__ p, int ct) {
    const float4 v = ld4_global(p + ct * 16 + 4 * (lane_id() >> 4));
    return f32x4{v.x, v.y, v.z, v.w};
}

// a whole per-column vector in the strip's column layout, requested ahead of its use
template <int D> struct ColVec {
    f32x4 v[D / 16];
    __device__ __forceinline__ void load(const float* __restrict__ p) {
#pragma unroll
        for (int ct = 0; ct < D / 16; ++ct) v[ct] = col4(p, ct);
    }
};

// all-reduce over the 4 lanes (m, 0..3) that share a row, i.e. over lane ^ 16 and lane ^ 32: two half-exchanges on the VALU
// (v_permlane16_swap: odd rows of 16 lanes of the first operand <-> even rows of the second; v_permlane32_swap: upper half <-> lower
// half) instead of two ds_bpermute round trips through the LDS crossbar -- with one wave per SIMD nothing hides those
// (written as inline asm: with the builtins hipcc 7.2 treats the two results of a swap as equal when both inputs carry the same
// value and emits v_add v1, v1, v1 behind it; the two wait states a VALU write needs in front of a v_permlane read sit in the string)
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float row_sum4(float v) {
    float a = v, b = v;
    swap16(a, b);
    a += b; b = a;
    swap32(a, b);
    return a + b;
}
__device__ __forceinline__ float row_max4(float v) {
    float a = v, b = v;
    swap16(a, b);
    a = fmaxf(a, b); b = a;
    swap32(a, b);
    return fmaxf(a, b);
}
// sum over the 16 lanes (0..15, g) that share a column quad: DPP inside the row of 16
__device__ __forceinline__ float col_sum16(float v) { return group_sum<16>(v); }

template <int D>
__device__ __forceinline__ void strip_stats(const StripRegs<D>& x, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) s += (x.v[ct][0] + x.v[ct][1]) + (x.v[ct][2] + x.v[ct][3]);
    mean = row_sum4(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = x.v[ct][r] - mean; q = fmaf(d, d, q); }
    }
    rstd = 1.0f / sqrtf(row_sum4(q) * (1.0f / D) + eps);
}

// y = LayerNorm(x) (biased variance, eps inside the sqrt: torch.nn.LayerNorm as used at model_seq.py:342-353)
template <int D>
__device__ __forceinline__ void strip_layernorm(StripRegs<D>& y, const StripRegs<D>& x, const float* __restrict__ w, const float* __restrict__ b, float eps) {
    float mean, rstd;
    strip_stats<D>(x, eps, mean, rstd);
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
        const f32x4 ww = col4(w, ct), bb = col4(b, ct);
#pragma unroll
        for (int r = 0; r < 4; ++r) y.v[ct][r] = (x.v[ct][r] - mean) * rstd * ww[r] + bb[r];
    }
}

template <int D>
__device__ __forceinline__ void strip_layernorm(StripRegs<D>& y, const StripRegs<D>& x, const ColVec<D>& w, const ColVec<D>& b, float eps) {
    float mean, rstd;
    strip_stats<D>(x, eps, mean, rstd);
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) y.v[ct][r] = (x.v[ct][r] - mean) * rstd * w.v[ct][r] + b.v[ct][r];
}

// ---- dropout on a strip ---------------------------------------------------------------------------------------------------------
// keep multipliers of this lane's 32 (D = 128) columns of row `local` at `site`.  p = 0.5 (1 bit per decision, rng.h): ONE Philox
// call covers 128 consecutive elements = the whole row at D = 128 (two rows at D = 64); other p: one call per column quad.
template <int D>
__device__ __forceinline__ void strip_dropout(StripRegs<D>& x, unsigned long long seed, unsigned site, unsigned step, int local, unsigned spec, float scale) {
    const int g4 = 4 * (lane_id() >> 4);
    if (spec_bits(spec) == 1) {
        const unsigned long long e0 = (unsigned long long)local * D;
        const uint4 rr = rng_call(seed, e0 >> 7, site, step);
        const int f0 = (int)(e0 & 127);                       // 0 (D = 128) or 0 / 64 (D = 64)
        const bool all = spec_thr(spec) == 0;
#pragma unroll
        for (int ct = 0; ct < D / 16; ++ct) {
            const int f = f0 + ct * 16 + g4;                  // 4 consecutive fields, never straddling a word
            const unsigned w = rng_word(rr, f >> 5) >> (f & 31);
#pragma unroll
            for (int r = 0; r < 4; ++r) x.v[ct][r] = (all || ((w >> r) & 1u)) ? x.v[ct][r] * scale : 0.f;
        }
    } else {
#pragma unroll
        for (int ct = 0; ct < D / 16; ++ct) {
            const float4 m = dropout_mult4(seed, site, step, (unsigned long long)local * D + ct * 16 + g4, spec, scale);
            x.v[ct][0] *= m.x; x.v[ct][1] *= m.y; x.v[ct][2] *= m.z; x.v[ct][3] *= m.w;
        }
    }
}

// the feature-level "== 0" bits of the row (tmq [2M, D / 4]: one byte per column quad, embed.hip): zero the flagged elements.
// The row's D / 4 bytes arrive as D / 64 16-byte loads; column tile ct, lane group g <-> byte 4 ct + g.
template <int D> struct StripTm { unsigned w[D / 16]; };
template <int D>
__device__ __forceinline__ void strip_tm_load(StripTm<D>& tm, const GBuf& g, const StripRow& row) {
    const unsigned base = row.ok ? (row.off - 16u * (unsigned)(lane_id() >> 4)) >> 4 : STRIP_OOB >> 4;     // row * D / 4 bytes
#pragma unroll
    for (int q = 0; q < D / 64; ++q) {
        const amid_v4u v = g.load4u(base + 16 * q);
        tm.w[4 * q] = v.x; tm.w[4 * q + 1] = v.y; tm.w[4 * q + 2] = v.z; tm.w[4 * q + 3] = v.w;
    }
}
template <int D>
__device__ __forceinline__ void strip_apply_tm(StripRegs<D>& x, const StripTm<D>& tm) {
    const int sh = 8 * (lane_id() >> 4);
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) {
        const unsigned bits = tm.w[ct] >> sh;
        x.v[ct][0] = (bits & 1u) ? 0.f : x.v[ct][0];
        x.v[ct][1] = (bits & 2u) ? 0.f : x.v[ct][1];
        x.v[ct][2] = (bits & 4u) ? 0.f : x.v[ct][2];
        x.v[ct][3] = (bits & 8u) ? 0.f : x.v[ct][3];
    }
}

// ---- weight ring ------------------------------------------------------------------------------------------------------------------
// W [D out-features][D in-features] row-major in global memory -> LDS image [n][D], the 16-byte chunk c of row n stored at chunk
// position c ^ (n & 15).  One DMA wave-instruction fills 1 KiB of LDS linearly (64 lanes x 16 B); lane L of piece k fills chunk
// position (64 k + L) % (D / 4) of row (64 k + L) / (D / 4), so it must FETCH chunk position ^ (row & 15).
// A ds_read_b128 is served in 4 groups of 16 lanes; the 16 lanes of a group read 16 different n (mod 16) at chunk 4 ct + g with two
// values of g: positions (4 ct + g) ^ i cover all 16 slots of the 256-byte bank window exactly once (strip_mma below).
// this wave's pieces are k = 4 k0 + wave, k0 < WDma::PER_WAVE; the per-lane source offsets repeat with period 2 in k0 (the row advances
// by 256 / CPR per k0 step: n & 15 flips bit 3 at D = 128 and does not change at D = 64), so two offsets + a stride describe them all
template <int D> struct WDma {
    static constexpr int CPR = D / 4;                 // 16-byte chunks per row
    static constexpr int PER_WAVE = D * CPR / 64 / STRIP_WAVES;
    static constexpr unsigned STRIDE2 = 2u * (256 / CPR) * D * 4;      // bytes between pieces k0 and k0 + 2
    unsigned off[2];
    int w;
    __device__ __forceinline__ WDma() {
        const int lane = lane_id();
        w = wave_id();
#pragma unroll
        for (int k0 = 0; k0 < 2; ++k0) {
            const int p = (k0 * STRIP_WAVES + w) * 64 + lane;
            const int n = p / CPR, pos = p % CPR;
            off[k0] = (unsigned)((n * D + ((pos ^ (n & 15)) * 4)) * 4);
        }
    }
    // One piece as INLINE ASM: beside a compiler-visible LDS-DMA hipcc waits lgkmcnt(0) in front of every MFMA group that follows a
    // fragment read (it cannot order the DMA's LDS write against ds_reads by a count), i.e. right behind the read it has just issued:
    // a full LDS latency per k tile.  The asm is invisible to its counters -- completion is waited for by w_ring_wait() (vmcnt(0))
    // + the workgroup barrier in front of the slab's first read; a compiler wait vmcnt(N) for an older ordinary load only becomes
    // stricter by the uncounted pieces.  M0 (the DMA's LDS base) is saved and restored inside the statement.
    __device__ __forceinline__ void piece(float* __restrict__ buf, const float* __restrict__ W, int k0) const {
        const unsigned voff = off[k0 & 1] + (unsigned)(k0 >> 1) * STRIDE2;
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(buf + (k0 * STRIP_WAVES + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
    __device__ __forceinline__ void all(float* __restrict__ buf, const float* __restrict__ W) const {
#pragma unroll
        for (int k0 = 0; k0 < PER_WAVE; ++k0) piece(buf, W, k0);
    }
};
// every DMA (and every other vector-memory operation) of this wave has completed
__device__ __forceinline__ void w_ring_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct NoDeferred { __device__ __forceinline__ void operator()(int, int) const {} };

// acc[co] += sum_k A[.][k] W[co * 16 + .][k] over the whole K = D of the slab in `buf`; `deferred(ct)` is called once per k tile,
// behind its MFMAs (stores of an earlier epilogue drain under the matrix work)
// Issue order, pinned: hipcc's scheduler, left alone, sinks every fragment read to just in front of its four MFMAs, waits for it with
// lgkmcnt(0) and issues the four as one dependent chain on one accumulator (40-cycle dependent latency against the 32-cycle issue
// rate, plus an exposed LDS latency per read: measured 2x the time).  So the loop is written as groups of NT / 2 MFMAs on DIFFERENT
// accumulators + one fragment read of the NEXT k tile, with a scheduling barrier behind every group that MFMA and LDS instructions
// may not cross (VALU / SALU / VMEM may: epilogue arithmetic and stores of neighbouring code still slide under the matrix work).
#define AMID_STRIP_FENCE() __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x400)
// `hook(ct, j)` runs behind group j of k tile ct (8 groups per k tile): deferred stores of an earlier epilogue, the DMA pieces of the
// next weight slab -- vector-memory instructions stay where they are written (they may not cross the fences either: a load placed
// in front of the loop is in flight during the loop, a store placed in group j is issued there)
template <int D, class Hook = NoDeferred>
__device__ __forceinline__ void strip_mma(f32x4 (&acc)[D / 16], const StripRegs<D>& A, const float* __restrict__ buf, const Hook& hook = NoDeferred()) {
    constexpr int NT = D / 16, HALF = NT / 2;
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const int xl = g ^ i;
    const float* rowp = buf + i * D;
    float4 wf[2][NT];
#pragma unroll
    for (int co = 0; co < NT; ++co) wf[0][co] = ld4(rowp + co * 16 * D + 4 * xl);          // k tile 0: chunk (0 ^ xl)
    AMID_STRIP_FENCE();
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        const float* nxt = rowp + 4 * (((ct + 1) * 4) ^ xl);
        // 8 groups per k tile: (element r, accumulators co = half * HALF ..); group j also reads fragment j of k tile ct + 1
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = j >> 1, c0 = (j & 1) * HALF;
#pragma unroll
            for (int c = 0; c < HALF; ++c) {
                const float4 w = wf[ct & 1][c0 + c];
                const float wr = r == 0 ? w.x : r == 1 ? w.y : r == 2 ? w.z : w.w;
                acc[c0 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, A.v[ct][r], acc[c0 + c], 0, 0, 0);
            }
            if (ct + 1 < NT && j < NT) wf[(ct + 1) & 1][j] = ld4(nxt + j * 16 * D);
            hook(ct, j);
            AMID_STRIP_FENCE();
        }
    }
}

// ---- bf16 matrix products on a strip (BASELINE.json configs[2]) -------------------------------------------------------------------------
// Weights arrive as bf16 FRAGMENT IMAGES (amid_sas_weights_bf16): row n = D bf16 = D / 8 chunks of 16 bytes, chunk 4 s + g = the eight k
// lane group g supplies in k-step s of v_mfma_f32_16x16x32_bf16 when the operand sits in the C layout: k = 32 s + 4 g + 0..3 (column tile
// 2 s) and 32 s + 16 + 4 g + 0..3 (column tile 2 s + 1).  A slab is D D / 2 floats (32 KB at D = 128); in LDS chunk c of row n sits at
// chunk position c ^ (n & 15), applied on the DMA's source address: conflict-free ds_read_b128 fragments.  The operand's values are
// rounded to bf16 (nearest even) on the fly, accumulation stays fp32.
template <int D> struct WDma16 {
    static constexpr int CPR = D / 8;
    static constexpr int PER_WAVE = D * CPR / 64 / STRIP_WAVES;
    unsigned off0;
    int w;
    __device__ __forceinline__ WDma16() {
        w = wave_id();
        const int p = w * 64 + lane_id();
        const int n = p / CPR, pos = p % CPR;
        off0 = (unsigned)(n * D * 2 + ((pos ^ (n & 15)) * 16));
    }
    __device__ __forceinline__ void piece(float* __restrict__ buf, const float* __restrict__ W, int k0) const {
#ifdef AMID_EXP_NO_DMA      // (variant libraries only: what a launch costs when its weight stream is free -- results are garbage)
        return;
#endif
        const unsigned voff = off0 + (unsigned)k0 * (unsigned)(STRIP_WAVES * 64 / CPR) * (unsigned)(D * 2);      // 16 rows further: n & 15 unchanged
        const unsigned lds = __builtin_amdgcn_readfirstlane(
            lds_offset(buf + (k0 * STRIP_WAVES + w) * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(W), "s"(lds) : "memory");
    }
    __device__ __forceinline__ void all(float* __restrict__ buf, const float* __restrict__ W) const {
#pragma unroll
        for (int k0 = 0; k0 < PER_WAVE; ++k0) piece(buf, W, k0);
    }
};

typedef __bf16 strip_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 strip_bf16x2 __attribute__((ext_vector_type(2)));
typedef float strip_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned strip_pack2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(strip_f32x2{a, b}, strip_bf16x2));
}
// acc[co] += A W^T with bf16 operands; `hook(ct, j)` is called for every slot of the fp32 loop IN FRONT of the 4 x D / 16 MFMAs (the
// next slab's DMA pieces and the deferred stores the callers place in the loop: there is no long loop to spread them under)
template <int D, class Hook = NoDeferred>
__device__ __forceinline__ void strip_mma16(f32x4 (&acc)[D / 16], const StripRegs<D>& A, const float* __restrict__ buf, const Hook& hook = NoDeferred()) {
    constexpr int NT = D / 16, KS = D / 32;
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int j = 0; j < 8; ++j) hook(ct, j);
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    const float* rowp = buf + i * (D / 2);
    amid_v4u a16[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        a16[s] = amid_v4u{strip_pack2(A.v[2 * s][0], A.v[2 * s][1]), strip_pack2(A.v[2 * s][2], A.v[2 * s][3]),
                          strip_pack2(A.v[2 * s + 1][0], A.v[2 * s + 1][1]), strip_pack2(A.v[2 * s + 1][2], A.v[2 * s + 1][3])};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        float4 wf[NT];
#pragma unroll
        for (int co = 0; co < NT; ++co) wf[co] = ld4(rowp + co * 16 * (D / 2) + 4 * ((4 * s + g) ^ i));
#pragma unroll
        for (int co = 0; co < NT; ++co)
            acc[co] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(strip_bf16x8, wf[co]), __builtin_bit_cast(strip_bf16x8, a16[s]),
                                                              acc[co], 0, 0, 0);
    }
}
// acc[co] += A W^T, fp32 operands as three bf16 pieces each (csrc/bf16_pieces.h), six piece pairs: 192 matrix instructions of 16 cycles
// instead of 256 of 32.  The weight is three fragment images (planes hi, mid, lo) walked plane by plane through strip_chain.h's
// RingP3: lo x hi and mid x (mid, hi) of the operand, the ring's barrier, then hi x (lo, mid, hi).  A whole-row strip splits its operand
// ONCE (every wave multiplies its own 16 rows by the whole weight: 176 vector instructions beside 192 matrix instructions).
// Hand-placed LDS fragment reads: the compiler, at this kernel's register pressure, sinks every ds_read to just in front of its first
// use -- read, a full LDS round trip, three matrix instructions, read ... (a wave is alone on its SIMD in the strip kernels: nobody fills
// the wait; measured 7 - 11 k cycles per product against 3 072 of matrix issue, and sched_group_barrier did not move it).  So the reads
// are inline asm issued PD steps ahead and the wait is an asm that RETURNS the fragment: its users cannot be scheduled in front of it.
// LDS operations return in order: lgkmcnt(N) = everything but the N youngest is back (operations the compiler adds only make it wait more).
// steps (one column tile of one k-step: three matrix instructions) a fragment read runs ahead of its use, for passes that read two
// fragments per step / one: depths 1 ... 8 measure the same in the step; -DAMID_FRAG_AHEAD_2 / _1 for A/B builds
#ifndef AMID_FRAG_AHEAD_2
#define AMID_FRAG_AHEAD_2 3
#endif
#ifndef AMID_FRAG_AHEAD_1
#define AMID_FRAG_AHEAD_1 4
#endif
constexpr int FRAG_AHEAD_2 = AMID_FRAG_AHEAD_2, FRAG_AHEAD_1 = AMID_FRAG_AHEAD_1;
template <int OFF> __device__ __forceinline__ void lds_frag_issue(f32x4& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int YOUNGER> __device__ __forceinline__ void lds_frag_wait(f32x4& d) {
    static_assert(YOUNGER >= 0 && YOUNGER < 16, "lgkmcnt field");
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(YOUNGER));
}
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(const F& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(const F& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// SPREAD: the hook's 8 D / 16 slots run one per step INSIDE the two passes (slot 0 -- the ring's fetch() registration -- in front):
// BERT4Rec's chains carry the dropout counters' Philox rounds in their hooks (bert_strip.hip KeepGen: ~84 vector cycles per slot), which
// a lone wave issues beside its matrix instructions when they stand between them and in front of them when they stand in front.
template <int D, class RingT, class Hook = NoDeferred, bool SPREAD = false>
__device__ __forceinline__ void strip_mma16x6(f32x4 (&acc)[D / 16], const StripRegs<D>& A, RingT& ring, const Hook& hook = NoDeferred()) {
    constexpr int NT = D / 16, KS = D / 32;
    static_assert(!SPREAD || KS * NT * 2 == 8 * NT, "one hook slot per step of the two passes");
    if constexpr (SPREAD) {
        hook(0, 0);
    } else {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int j = 0; j < 8; ++j) hook(ct, j);       // (deferred stores; the ring learns the next weight from its fetch() calls)
    }
    ring.begin();
    const int lane = lane_id();
    const int i = lane & 15, g = lane >> 4;
    auto mma = [&](const f32x4& wf, const amid_v4u& a16, const f32x4& c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(strip_bf16x8, wf), __builtin_bit_cast(strip_bf16x8, a16), c, 0, 0, 0);
    };
    amid_v4u ah[KS], am[KS], al[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const WgSplit2 p0 = wg_split3(A.v[2 * s][0], A.v[2 * s][1]), p1 = wg_split3(A.v[2 * s][2], A.v[2 * s][3]);
        const WgSplit2 p2 = wg_split3(A.v[2 * s + 1][0], A.v[2 * s + 1][1]), p3 = wg_split3(A.v[2 * s + 1][2], A.v[2 * s + 1][3]);
        ah[s] = amid_v4u{p0.hi, p1.hi, p2.hi, p3.hi}; am[s] = amid_v4u{p0.mid, p1.mid, p2.mid, p3.mid}; al[s] = amid_v4u{p0.lo, p1.lo, p2.lo, p3.lo};
    }
    // byte offset of this lane's fragment of k-step s inside a plane (row i of column tile 0; a column tile further = 16 rows = +4 KB)
    unsigned fo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) fo[s] = (unsigned)((i * (D / 2) + 4 * ((4 * s + g) ^ i)) * 4);
    STRIP_STAMP(28);
    constexpr int NSTEP = KS * NT, CT_BYTES = 16 * (D / 2) * 4, PLANE_BYTES = RingT::SLAB * 4;
    constexpr int PD1 = FRAG_AHEAD_2, PD2 = FRAG_AHEAD_1;  // steps a read runs ahead of its matrix instructions
    auto lds_addr = [](const float* p) { return lds_offset(p); };
    {   // pass 1: lo x hi, mid x (mid, hi) -- the lo plane sits one slot (32 KB) behind the mid plane: one address, two offsets
        static_assert(CT_BYTES * (NT - 1) + PLANE_BYTES < 65536, "the lo plane is reached through the offset field");
        const unsigned base = lds_addr(ring.mslot());
        // a step = (column tile co, k-step s), s fastest: the twelve matrix instructions of a column tile run back to back on ONE
        // accumulator -- measured (profiles/tools/probe/mfma_rate_probe.hip, a wave alone on its SIMD): 16.5 cycles per instruction in a
        // chain on one accumulator, 18.1 when the accumulator changes every one to three instructions
        f32x4 wm[PD1 + 1], wl[PD1 + 1];
        auto issue = [&](auto U) {
            constexpr int u = decltype(U)::value, k = u % (PD1 + 1);
            lds_frag_issue<(u / KS) * CT_BYTES>(wm[k], base + fo[u % KS]);
            lds_frag_issue<(u / KS) * CT_BYTES + PLANE_BYTES>(wl[k], base + fo[u % KS]);
        };
        static_for<PD1>(issue);
        static_for<NSTEP>([&](auto T) {
            constexpr int t = decltype(T)::value, co = t / KS, s = t % KS, k = t % (PD1 + 1);
            if constexpr (t + PD1 < NSTEP) issue(std::integral_constant<int, t + PD1>{});      // (into the slot step t - 1 is done with)
            constexpr int last = t + PD1 < NSTEP ? t + PD1 : NSTEP - 1;
            lds_frag_wait<2 * (last - t) + 1>(wm[k]);
            acc[co] = mma(wm[k], am[s], acc[co]);
            lds_frag_wait<2 * (last - t)>(wl[k]);
            acc[co] = mma(wl[k], ah[s], acc[co]); acc[co] = mma(wm[k], ah[s], acc[co]);
            if constexpr (SPREAD && t > 0) hook(t / 8, t % 8);                                  // slots 1 .. NSTEP - 1
        });
    }
    STRIP_STAMP(29);
    const unsigned hbase = lds_addr(ring.hcur());
    ring.mid_sync();
    STRIP_STAMP(30);
    {   // pass 2: hi x (lo, mid, hi)
        f32x4 wf[PD2 + 1];
        auto issue = [&](auto U) {
            constexpr int u = decltype(U)::value;
            lds_frag_issue<(u / KS) * CT_BYTES>(wf[u % (PD2 + 1)], hbase + fo[u % KS]);
        };
        static_for<PD2>(issue);
        static_for<NSTEP>([&](auto T) {
            constexpr int t = decltype(T)::value, co = t / KS, s = t % KS, k = t % (PD2 + 1);
            if constexpr (t + PD2 < NSTEP) issue(std::integral_constant<int, t + PD2>{});
            constexpr int last = t + PD2 < NSTEP ? t + PD2 : NSTEP - 1;
            lds_frag_wait<last - t>(wf[k]);
            acc[co] = mma(wf[k], al[s], acc[co]); acc[co] = mma(wf[k], am[s], acc[co]); acc[co] = mma(wf[k], ah[s], acc[co]);
            if constexpr (SPREAD) hook((NSTEP + t) / 8, (NSTEP + t) % 8);                       // slots NSTEP .. 2 NSTEP - 1
        });
    }
    STRIP_STAMP(31);
}

// fp32 or bf16 products, chosen at compile time
template <int D, bool BF, class Hook = NoDeferred>
__device__ __forceinline__ void strip_mma_sel(f32x4 (&acc)[D / 16], const StripRegs<D>& A, const float* __restrict__ buf, const Hook& hook = NoDeferred()) {
    if constexpr (BF) strip_mma16<D, Hook>(acc, A, buf, hook); else strip_mma<D, Hook>(acc, A, buf, hook);
}

template <int D>
__device__ __forceinline__ void strip_zero(f32x4 (&acc)[D / 16]) {
#pragma unroll
    for (int ct = 0; ct < D / 16; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
}

}  // namespace amid
