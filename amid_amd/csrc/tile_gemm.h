// fp32 MFMA row-tile GEMM primitives shared by the fused encoder kernels (linear.hip).
//
// Every dense op of the encoder is C[rows, N] = A[rows, K] * W[N, K]^T with K, N in {64, 128} per
// slab and rows = B*T*2 (tens of thousands): weights are tiny (<= 64 KB), activations stream.  A
// workgroup (512 threads = 8 waves, one per CU) owns a tile of up to TILE_ROWS = 112 activation rows
// (7 MFMA row tiles of 16) and ALL N columns of the slab, so that row-wise epilogues (bias, dropout,
// residual, LayerNorm, masks) run on complete rows while the tile is still on chip.  112 rather than
// 128: at the headline shape (B 256, T 50, two domains) 25 600 rows / 256 CUs = exactly 100 rows per
// CU; 16-row MFMA tiles waste 11 % there, 32-row tiles would waste 22 %.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  Operand maps (cdna_hip_programming.md §3):
//   A: lane l holds A[i = l & 15][k = l >> 4]   B: lane l holds B[k = l >> 4][j = l & 15]
//   C: reg r of lane l is C[row = (l >> 4) * 4 + r][col = l & 15]
// Both operands come from K-contiguous LDS images ([row][K] for A, [n][K] for W) with one
// ds_read_b128 per 4 MFMAs: lane (i, g) reads k = 16*kk + 4*g .. +3 and issue j of the group uses
// element j, i.e. the four lane groups of issue j cover k = 16*kk + {0,4,8,12} + j.  A and B use the
// same assignment, and a sum over k does not care about the order.
// LDS row stride K + 8 floats: (K + 8) * 4 B = 2 * 256 + 32 (K = 128) or 256 + 32 (K = 64) puts the
// 16 lanes of every ds_read_b128 lane group on 16 distinct 16-B slots (slot = 2 i + g mod 16).
#pragma once
#include "common.h"

namespace amid {

constexpr int TILE_ROWS = 112;
constexpr int TILE_RT = TILE_ROWS / 16;   // 7
constexpr int GEMM_THREADS = 512;

template <int K> struct TileCfg {
    static constexpr int LDK = K + 8;                 // A / W image row stride (floats)
    static constexpr int A_FLOATS = TILE_ROWS * LDK;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// wave -> (column tile, row-tile group) for an N-column slab: NT = N/16 column tiles, WR = 8/NT row groups
template <int N> struct WaveMap {
    static constexpr int NT = N / 16;
    static constexpr int WR = 8 / NT;
    static constexpr int ACC = (TILE_RT + WR - 1) / WR;     // accumulator tiles per wave: 7 (N=128) or 4 (N=64)
};

// Copy `nrows` rows x K columns (global row stride ldg floats, first column k0) into an LDS image
// [TILE_ROWS][K+8]; rows >= nrows are zero-filled so the MFMAs may run on whole 16-row tiles.
template <int K>
__device__ __forceinline__ void stage_rows(float* __restrict__ As, const float* __restrict__ g, long long row0, int nrows, int ldg, int k0,
                                           int rows_to_fill) {
    constexpr int QPR = K / 4;
    constexpr int RPP = GEMM_THREADS / QPR;
    const int sub = threadIdx.x % QPR, rl = threadIdx.x / QPR;
    for (int r = rl; r < rows_to_fill; r += RPP) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nrows) v = ld4(g + (row0 + r) * ldg + k0 + 4 * sub);
        st4(As + r * TileCfg<K>::LDK + 4 * sub, v);
    }
}

// W slab: N rows (output features) x K columns, global row stride ldw
template <int K, int N>
__device__ __forceinline__ void stage_weights(float* __restrict__ Ws, const float* __restrict__ w, int ldw, int k0) {
    constexpr int QPR = K / 4;
    constexpr int RPP = GEMM_THREADS / QPR;
    const int sub = threadIdx.x % QPR, rl = threadIdx.x / QPR;
#pragma unroll 4
    for (int r = rl; r < N; r += RPP) st4(Ws + r * TileCfg<K>::LDK + 4 * sub, ld4(w + (long long)r * ldw + k0 + 4 * sub));
}

template <int K, int N>
__device__ __forceinline__ void mma_tile(const float* __restrict__ As, const float* __restrict__ Ws, f32x4 (&acc)[WaveMap<N>::ACC], int nrt) {
    using WM = WaveMap<N>;
    constexpr int LDK = TileCfg<K>::LDK;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int i = lane & 15, g = lane >> 4;
    const float* wp = Ws + (ct * 16 + i) * LDK + 4 * g;
    const float* ap = As + i * LDK + 4 * g;
#pragma unroll 2
    for (int kk = 0; kk < K / 16; ++kk) {
        const float4 b = ld4(wp + 16 * kk);
#pragma unroll
        for (int t = 0; t < WM::ACC; ++t) {
            const int rt = rg + t * WM::WR;
            if (rt < nrt) {
                const float4 a = ld4(ap + rt * 16 * LDK + 16 * kk);
                acc[t] = mfma16(a.x, b.x, acc[t]);
                acc[t] = mfma16(a.y, b.y, acc[t]);
                acc[t] = mfma16(a.z, b.z, acc[t]);
                acc[t] = mfma16(a.w, b.w, acc[t]);
            }
        }
    }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[WaveMap<N>::ACC]) {
#pragma unroll
    for (int t = 0; t < WaveMap<N>::ACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// accumulators -> LDS C image [TILE_ROWS][ldc]
template <int N>
__device__ __forceinline__ void acc_to_lds(float* __restrict__ Cs, int ldc, const f32x4 (&acc)[WaveMap<N>::ACC], int nrt) {
    using WM = WaveMap<N>;
    const int w = wave_id(), lane = lane_id();
    const int ct = w % WM::NT, rg = w / WM::NT;
    const int col = ct * 16 + (lane & 15), rbase = (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < WM::ACC; ++t) {
        const int rt = rg + t * WM::WR;
        if (rt < nrt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(rt * 16 + rbase + r) * ldc + col] = acc[t][r];
        }
    }
}

// Row-pass geometry over an N-column tile: QPR lanes own one row (a float4 each); the block covers
// RPP rows per pass.  A thread keeps the same column quad for every row it visits.
template <int N> struct RowPass {
    static constexpr int QPR = N / 4;
    static constexpr int RPP = GEMM_THREADS / QPR;
    __device__ static __forceinline__ int sub() { return threadIdx.x % QPR; }
    __device__ static __forceinline__ int first_row() { return threadIdx.x / QPR; }
};

// LayerNorm statistics of one row spread over QPR lanes (biased variance, eps inside the sqrt)
template <int QPR>
__device__ __forceinline__ void row_stats(float4 x, int n, float eps, float& mean, float& rstd) {
    mean = group_sum<QPR>(f4hsum(x)) * (1.0f / n);
    const float4 d = make_float4(x.x - mean, x.y - mean, x.z - mean, x.w - mean);
    const float var = group_sum<QPR>(f4hsum(f4mul(d, d))) * (1.0f / n);
    rstd = 1.0f / sqrtf(var + eps);
}

}  // namespace amid
