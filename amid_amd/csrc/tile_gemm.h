// fp32 MFMA row-tile GEMM primitives shared by the fused encoder kernels (sasrec_fwd.hip / sasrec_bwd.hip).
//
// Every dense op of the encoder is C[rows, N] = A[rows, K] * W[N, K]^T with K, N in {64, 128} per
// slab and rows = B*T*2 (tens of thousands): weights are tiny (<= 64 KB), activations stream.  A
// workgroup (512 threads = 8 waves, one per CU) owns a tile of up to TILE_ROWS = 112 activation rows
// (7 MFMA row tiles of 16) and ALL N columns of the slab, so that row-wise epilogues (bias, dropout,
// residual, LayerNorm, masks) run on complete rows while the tile is still on chip.  112 rather than
// 128: at the headline shape (B 256, T 50, two domains) 25 600 rows / 256 CUs = exactly 100 rows per
// CU; 16-row MFMA tiles waste 11 % there, 32-row tiles would waste 22 %.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain), issued as C^T = W * A^T:
//   first operand : lane l holds W[n = l & 15][k = l >> 4]      second: lane l holds A[m = l & 15][k = l >> 4]
//   result        : reg r of lane l is C[m = l & 15][n = (l >> 4) * 4 + r]
// so a lane ends up with FOUR CONSECUTIVE COLUMNS of one row: accumulators leave as 16-byte stores
// (to LDS or straight to global), a quarter of the instructions of the C = A W^T orientation.
// Both operands come from K-contiguous LDS images ([row][K] for A, [n][K] for W) with one
// ds_read_b128 per 4 MFMAs: lane (i, g) reads k = 16*kk + 4*g .. +3 and issue j of the group uses
// element j, i.e. the four lane groups of issue j cover k = 16*kk + {0,4,8,12} + j (same assignment
// on both operands; a sum over k does not care about the order).
// LDS row stride K + 8 floats: (K + 8) * 4 B = 2 * 256 + 32 (K = 128) or 256 + 32 (K = 64) puts the
// 16 lanes of every ds_read_b128 lane group on 16 distinct 16-B slots (slot = 2 i + g mod 16).
//
// Latency: every global -> LDS staging and every row-wise epilogue input goes through REGISTER
// prefetch (TileRegs / WRegs): all loads of a phase are issued back to back, before the MFMA loop
// that precedes their use, instead of one dependent load per row-loop iteration (the first version of
// these kernels spent ~2/3 of its time in such load->use chains: profiles/r01).
#pragma once
#include "common.h"

namespace amid {

// The row-tile height is a build parameter: sasrec_fwd.hip / sasrec_bwd.hip are compiled three times, with 7 MFMA row tiles (112
// rows, the headline shape's 100 rows per CU), with 5 (80 rows) and with 3 (48 rows: every accumulator tile is always computed, so
// at seq_len 20 -- the mybank shape, 40 rows per CU -- the 112-row build spends 64 % of its matrix work on zero rows).  The extra
// builds live in namespaces amid_rt5 / amid_rt3 and export the same entry points with the suffix _rt5 / _rt3 (csrc/Makefile); the
// host picks by rows_per_tile.
#ifndef AMID_TILE_RT
#define AMID_TILE_RT 7
#endif
#ifndef AMID_ENTRY_SUFFIX
#define AMID_ENTRY_SUFFIX
#endif
#define AMID_CAT2(a, b) a##b
#define AMID_CAT(a, b) AMID_CAT2(a, b)
#define AMID_ENTRY(name) AMID_CAT(name, AMID_ENTRY_SUFFIX)
constexpr int TILE_RT = AMID_TILE_RT;
constexpr int TILE_ROWS = 16 * TILE_RT;   // 112 (or 48)
constexpr int GEMM_THREADS = 512;

template <int K> struct TileCfg {
    static constexpr int LDK = K + 8;                 // A / W image row stride (floats)
    // rows of the A image: every accumulator tile of every wave is computed unconditionally (no per-tile branch in
    // the MFMA loop), so the image covers 16 * ACC * WR rows: 112 at D = 128, 128 at D = 64 (48 / 64 in the _rt3 build; rows past the tile are zero)
    static constexpr int ROWS = (K >= 128) ? TILE_ROWS : 32 * ((TILE_RT + 1) / 2);
    static constexpr int A_FLOATS = ROWS * LDK;
    // second region: the W slab [K rows][LDK], reused as the C image [ROWS][K + 4]
    static constexpr int W_FLOATS = (K * LDK > ROWS * (K + 4)) ? K * LDK : ROWS * (K + 4);
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- bf16 matrix-core mode (BASELINE.json configs[2]: "bf16 with fp32 ref tolerance check") -------------------------------------
// BF = true: the MFMA operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) when they are staged into LDS and the
// products run on v_mfma_f32_16x16x32_bf16 (16x the fp32 matrix rate), accumulating in fp32; everything outside the matrix products
// (LayerNorm, softmax, residuals, epilogues, what is stored in HBM, Adam) stays fp32.  LDS images become [row][K + 8] bf16: the
// 16-byte fragment of lane (i, g) is k = 32 kk + 8 g .. + 7, and the 272-byte row stride puts the 16 lanes of a lane group on 16
// distinct 16-byte slots ((17 i + g) mod 16).  The C image stays fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const bf16x2 v = __builtin_convertvector(f32x2v{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ uint2 pack_bf16x4(float4 v) { return make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
// four consecutive k of image row r (A or W image, K + 8 elements per row)
template <int K, bool BF>
__device__ __forceinline__ void store_a4(float* __restrict__ img, int r, int sub, float4 v) {
    if constexpr (BF) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(img) + r * (K + 8) + 4 * sub) = pack_bf16x4(v);
    else st4(img + r * (K + 8) + 4 * sub, v);
}

// wave -> (column tile, row-tile group) for an N-column slab: NT = N/16 column tiles, WR = 8/NT row groups
template <int N> struct WaveMap {
    static constexpr int NT = N / 16;
    static constexpr int WR = 8 / NT;
    static constexpr int ACC = (TILE_RT + WR - 1) / WR;     // accumulator tiles per wave: 7 (N=128) or 4 (N=64)
};

// Row-pass geometry over an N-column tile: QPR lanes own one row (a float4 each); the block covers
// RPP rows per pass; a thread visits rows first_row() + i * RPP, i < NR, always with the same column quad.
template <int N> struct RowPass {
    static constexpr int QPR = N / 4;
    static constexpr int RPP = GEMM_THREADS / QPR;
    static constexpr int NR = (TILE_ROWS + RPP - 1) / RPP;  // 7 (N=128) or 4 (N=64)
    __device__ static __forceinline__ int sub() { return threadIdx.x % QPR; }
    __device__ static __forceinline__ int first_row() { return threadIdx.x / QPR; }
};

// a thread's share of a [rows, N] tile held in registers (row i of the thread = first_row() + i * RPP)
template <int N> struct TileRegs { float4 v[RowPass<N>::NR]; };

// issue every load of the thread's share; rows >= nrows read as zero
template <int N>
__device__ __forceinline__ void load_tile(TileRegs<N>& t, const float* __restrict__ g, long long row0, int nrows, int ldg) {
    using RP = RowPass<N>;
    const int sub = RP::sub(), r0 = RP::first_row();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = r0 + i * RP::RPP;
        t.v[i] = (r < nrows) ? ld4(g + (row0 + r) * ldg + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// the same through a tile-row -> global-row map (tiles over scattered rows: the backward of the live sequences only)
template <int N, class RowFn>
__device__ __forceinline__ void load_tile_rows(TileRegs<N>& t, const float* __restrict__ g, RowFn rowf, int nrows, int ldg) {
    using RP = RowPass<N>;
    const int sub = RP::sub(), r0 = RP::first_row();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) {
        const int r = r0 + i * RP::RPP;
        t.v[i] = (r < nrows) ? ld4(g + rowf(r) * ldg + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// registers -> LDS image [TileCfg::ROWS][N + 8]; every image row is written (NR * RPP == ROWS), rows past the tile as zeros
template <int N, bool BF = false>
__device__ __forceinline__ void tile_to_lds(float* __restrict__ As, const TileRegs<N>& t) {
    using RP = RowPass<N>;
    static_assert(RP::NR * RP::RPP == TileCfg<N>::ROWS, "row pass must cover the A image exactly");
    const int sub = RP::sub(), r0 = RP::first_row();
#pragma unroll
    for (int i = 0; i < RP::NR; ++i) store_a4<N, BF>(As, r0 + i * RP::RPP, sub, t.v[i]);
}

// W slab [N rows (output features)][K] in registers / LDS
template <int K, int N> struct WRegs {
    static constexpr int QPR = K / 4, RPP = GEMM_THREADS / QPR, NR = N / RPP;
    float4 v[NR];
};
template <int K, int N>
__device__ __forceinline__ void load_w(WRegs<K, N>& t, const float* __restrict__ w, int ldw) {
    using W = WRegs<K, N>;
    const int sub = threadIdx.x % W::QPR, r0 = threadIdx.x / W::QPR;
#pragma unroll
    for (int i = 0; i < W::NR; ++i) t.v[i] = ld4(w + (long long)(r0 + i * W::RPP) * ldw + 4 * sub);
}
template <int K, int N, bool BF = false>
__device__ __forceinline__ void w_to_lds(float* __restrict__ Ws, const WRegs<K, N>& t) {
    using W = WRegs<K, N>;
    const int sub = threadIdx.x % W::QPR, r0 = threadIdx.x / W::QPR;
#pragma unroll
    for (int i = 0; i < W::NR; ++i) store_a4<K, BF>(Ws, r0 + i * W::RPP, sub, t.v[i]);
}

// ---- deferred epilogue stores -------------------------------------------------------------------------------------------
// A CU drains global stores at only ~8 B per clock: in a row pass that writes a [rows, N] tensor every wave sits at a full store
// queue with the matrix pipe idle (s_memtime: 7 500 cycles to ISSUE seven st4).  Many of those tensors (LN
// outputs, relu outputs, dgrad inputs) are ALSO written to the A image for the next GEMM -- so the row pass skips the global store
// and the next mma_tile issues it, one row slot per k-step between its MFMAs, re-reading the values from the A image: the stores
// drain under the matrix work at no register cost.  fp32 images only (the bf16 images hold rounded values).
// Every store is UNCONDITIONAL so that the compiler can count the stores in flight and wait for older loads with
// s_waitcnt vmcnt(k > 0): a dead slot (row >= nrows) re-stores a live element -- the thread's first row, or row 0 of the tile for
// a thread without live rows: same address, same value as its owner writes.
template <int N> struct ImageRowsPending {
    const float* img;          // A image [ROWS][N + 8]
    float* out;                // global [.., ld], already offset to the tile's first row
    int ld, nrows;
    static constexpr int COUNT = RowPass<N>::NR;
    __device__ __forceinline__ void issue(int i) const {
        using RP = RowPass<N>;
        const int r0 = RP::first_row(), sub = RP::sub();
        const int r = r0 + i * RP::RPP;
        const int row = (r < nrows) ? r : ((r0 < nrows) ? r0 : 0);
        st4_global(out + (long long)row * ld + 4 * sub, ld4(img + row * (N + 8) + 4 * sub));
    }
};
struct NoPending {
    static constexpr int COUNT = 0;
    __device__ __forceinline__ void issue(int) const {}
};

// MFMA loop over the whole K of the slab.  Software-pipelined by hand: the operand fragments of k-step kk+1 are read
// from LDS while the MFMAs of step kk issue, and inside a step the issue order is element-major / tile-minor, so
// back-to-back MFMAs hit DIFFERENT accumulators (a 16x16x4 f32 MFMA issues every 32 cycles but needs 40 before a
// dependent one).  No per-tile branch: all ACC tiles are always computed (rows past the tile are zeros in the image).
// The first version (read -> wait -> 4 dependent MFMAs per tile, a branch per tile) ran the matrix pipe at ~37 %.
template <int K, int N>
__device__ __forceinline__ void mma_tile_bf16(const float* __restrict__ As, const float* __restrict__ Ws, f32x4 (&acc)[WaveMap<N>::ACC]) {
    using WM = WaveMap<N>;
    constexpr int LDB = K + 8, ACC = WM::ACC, KK = K / 32;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int i = lane & 15, g = lane >> 4;
    const unsigned short* wp = reinterpret_cast<const unsigned short*>(Ws) + (ct * 16 + i) * LDB + 8 * g;
    const unsigned short* ap = reinterpret_cast<const unsigned short*>(As) + (rg * 16 + i) * LDB + 8 * g;
    bf16x8 b_cur = *reinterpret_cast<const bf16x8*>(wp), a_cur[ACC];
#pragma unroll
    for (int t = 0; t < ACC; ++t) a_cur[t] = *reinterpret_cast<const bf16x8*>(ap + t * WM::WR * 16 * LDB);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        bf16x8 b_nxt = b_cur, a_nxt[ACC];
#pragma unroll
        for (int t = 0; t < ACC; ++t) a_nxt[t] = a_cur[t];
        if (kk + 1 < KK) {
            b_nxt = *reinterpret_cast<const bf16x8*>(wp + 32 * (kk + 1));
#pragma unroll
            for (int t = 0; t < ACC; ++t) a_nxt[t] = *reinterpret_cast<const bf16x8*>(ap + t * WM::WR * 16 * LDB + 32 * (kk + 1));
        }
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_cur, a_cur[t], acc[t], 0, 0, 0);
        b_cur = b_nxt;
#pragma unroll
        for (int t = 0; t < ACC; ++t) a_cur[t] = a_nxt[t];
    }
}

template <int K, int N, class P0, class P1>
__device__ __forceinline__ void mma_tile_f32(const float* __restrict__ As, const float* __restrict__ Ws, f32x4 (&acc)[WaveMap<N>::ACC],
                                             const P0& p0, const P1& p1) {
    using WM = WaveMap<N>;
    constexpr int LDK = TileCfg<K>::LDK, ACC = WM::ACC, KK = K / 16;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int i = lane & 15, g = lane >> 4;
    const float* wp = Ws + (ct * 16 + i) * LDK + 4 * g;
    const float* ap = As + (rg * 16 + i) * LDK + 4 * g;
    float4 b_cur = ld4(wp), a_cur[ACC];
#pragma unroll
    for (int t = 0; t < ACC; ++t) a_cur[t] = ld4(ap + t * WM::WR * 16 * LDK);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        float4 b_nxt = b_cur, a_nxt[ACC];
#pragma unroll
        for (int t = 0; t < ACC; ++t) a_nxt[t] = a_cur[t];
        if (kk + 1 < KK) {
            b_nxt = ld4(wp + 16 * (kk + 1));
#pragma unroll
            for (int t = 0; t < ACC; ++t) a_nxt[t] = ld4(ap + t * WM::WR * 16 * LDK + 16 * (kk + 1));
        }
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = mfma16(b_cur.x, a_cur[t].x, acc[t]);
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = mfma16(b_cur.y, a_cur[t].y, acc[t]);
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = mfma16(b_cur.z, a_cur[t].z, acc[t]);
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = mfma16(b_cur.w, a_cur[t].w, acc[t]);
        // pin the issue order: the ACC + 1 fragment reads of step kk + 1 are spread behind the 4 * ACC MFMAs of step kk
        // (left alone, the scheduler sinks each read to just before its first use and waits on it)
        if (kk + 1 < KK) {
#pragma unroll
            for (int t = 0; t < ACC + 1; ++t) {
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        {   // deferred stores of the previous epilogue(s): one row slot (a few at K = 64) per k-step, behind this step's MFMAs
            constexpr int C0 = (P0::COUNT + KK - 1) / KK, C1 = (P1::COUNT + KK - 1) / KK, PER = C0 > C1 ? C0 : C1;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (kk * PER + u < P0::COUNT) p0.issue(kk * PER + u);
                if (kk * PER + u < P1::COUNT) p1.issue(kk * PER + u);
            }
        }
        b_cur = b_nxt;
#pragma unroll
        for (int t = 0; t < ACC; ++t) a_cur[t] = a_nxt[t];
    }
}

// p0 / p1: stores a previous row pass left to this loop (ImageRowsPending; fp32 images only -- ignored in the bf16 mode, whose
// callers keep storing directly)
template <int K, int N, bool BF = false, class P0 = NoPending, class P1 = NoPending>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Ws, f32x4 (&acc)[WaveMap<N>::ACC],
                                         const P0& p0 = NoPending(), const P1& p1 = NoPending()) {
    if constexpr (BF) mma_tile_bf16<K, N>(As, Ws, acc);
    else mma_tile_f32<K, N>(As, Ws, acc, p0, p1);
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[WaveMap<N>::ACC]) {
#pragma unroll
    for (int t = 0; t < WaveMap<N>::ACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// accumulators -> LDS C image [TILE_ROWS][ldc]: one 16-byte store per tile per lane
template <int N>
__device__ __forceinline__ void acc_to_lds(float* __restrict__ Cs, int ldc, const f32x4 (&acc)[WaveMap<N>::ACC]) {
    using WM = WaveMap<N>;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int m = lane & 15, n = ct * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < WM::ACC; ++t) {
        const int rt = rg + t * WM::WR;
        st4(Cs + (rt * 16 + m) * ldc + n, make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]));
    }
}

// accumulators (+ bias) -> global [rows, ldo] directly: 16-byte stores, 64 contiguous bytes per row per wave instruction
template <int N>
__device__ __forceinline__ void acc_to_global(float* __restrict__ out, long long row0, int nrows, int ldo, const float* __restrict__ bias,
                                              const f32x4 (&acc)[WaveMap<N>::ACC]) {
    using WM = WaveMap<N>;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int m = lane & 15, n = ct * 16 + (lane >> 4) * 4;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ld4(bias + n);
#pragma unroll
    for (int t = 0; t < WM::ACC; ++t) {
        const int rt = rg + t * WM::WR;
        const int r = rt * 16 + m;
        if (r < nrows)
            st4(out + (row0 + r) * ldo + n, make_float4(acc[t][0] + b.x, acc[t][1] + b.y, acc[t][2] + b.z, acc[t][3] + b.w));
    }
}
template <int N, class RowFn>
__device__ __forceinline__ void acc_to_global_rows(float* __restrict__ out, RowFn rowf, int nrows, int ldo, const float* __restrict__ bias,
                                              const f32x4 (&acc)[WaveMap<N>::ACC]) {
    using WM = WaveMap<N>;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int m = lane & 15, n = ct * 16 + (lane >> 4) * 4;
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = ld4(bias + n);
#pragma unroll
    for (int t = 0; t < WM::ACC; ++t) {
        const int rt = rg + t * WM::WR;
        const int r = rt * 16 + m;
        if (r < nrows)
            st4(out + rowf(r) * ldo + n, make_float4(acc[t][0] + b.x, acc[t][1] + b.y, acc[t][2] + b.z, acc[t][3] + b.w));
    }
}

// LayerNorm statistics of one row spread over QPR lanes (biased variance, eps inside the sqrt)
template <int QPR>
__device__ __forceinline__ void row_stats(float4 x, int n, float eps, float& mean, float& rstd) {
    mean = group_sum<QPR>(f4hsum(x)) * (1.0f / n);
    const float4 d = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
    const float var = group_sum<QPR>(f4hsum(f4mul(d, d))) * (1.0f / n);
    rstd = 1.0f / sqrtf(var + eps);
}

__device__ __forceinline__ float4 ln_apply(float4 x, float mean, float rstd, float4 w, float4 b) {
    return make_float4((x.x - mean) * rstd * w.x + b.x, (x.y - mean) * rstd * w.y + b.y, (x.z - mean) * rstd * w.z + b.z,
                       (x.w - mean) * rstd * w.w + b.w);
}

}  // namespace amid
