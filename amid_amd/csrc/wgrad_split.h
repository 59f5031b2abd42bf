// Weight-gradient tile dW[128][128] = dY^T X on the bf16 matrix cores at fp32 accuracy -- shared by SASRec's (sasrec_bwd.hip) and
// BERT4Rec's (bert.hip) weight-gradient launches.
#pragma once
#include "common.h"
#include "tile_gemm.h"
#include "bf16_pieces.h"

namespace amid {

__device__ __forceinline__ float f4comp_w(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// compute = "fp32" (mma mode 2 / 3 of amid_sas_wgrad*_f32, amid_bert_wgrad_mode_f32): every operand element is split into three bf16
// pieces x = hi + mid + lo (each rounded to nearest even; 3 x 8 significand bits plus the remainders' signs cover
// fp32's 24, so the sum is exact), and dY^T X = sum over the piece pairs of exact bf16 products accumulated in fp32.  NTERM = 9 keeps
// every pair; NTERM = 6 drops mid*lo, lo*mid, lo*lo (each <= 2^-24 of |x y|, below the rounding of the fp32 chain it replaces).
// v_mfma_f32_16x16x32_bf16 issues in 16 cycles against 8 x 32 for the same 16 x 16 x 32 block on v_mfma_f32_16x16x4_f32: 144 (96)
// cycles instead of 256.  A chunk = 32 rows = one k-step, staged transposed as three bf16 planes per operand ([column][32 rows] =
// 64 bytes per column, its four 16-byte k-groups XOR-swizzled by the column: slot = g ^ pi(column's index in its tile >> 2), pi =
// (0, 2, 3, 1)).  ds_read_b128 is served in FOUR lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS),
// not in quarters: with that swizzle each group's 16 fragments cover the 64 banks once.  (Round 4's first layout -- 80 bytes per
// column, no swizzle, derived for contiguous quarters -- read every fragment in 8 LDS cycles instead of 4: SQ_LDS_BANK_CONFLICT was
// half of SQ_LDS_IDX_ACTIVE, and the fragment reads of a chunk held the LDS longer than its matrix instructions hold the pipes.)
// A wave's 4-byte staging writes (lane = column quad & 3, row pair): see wr_j below.  56 KB of LDS, two
// workgroups per CU as sas_wgrad_kernel.
// Two chunks of loads stay in flight per thread (a chunk's 48 matrix instructions per wave are too short to hide a load).
// Measured (MI355X, 2 layers, B 256 x T 50 with the live-row hint, 21 splits, replayed): 46 us with six pairs, 56 with nine, against 69
// for sas_wgrad_kernel and 32 for bf16-rounded operands (that one is bound by the 157 MB of operands); error of the summed partials
// against the fp64 product 3.8e-7 of the largest entry for six and nine pairs alike, 4.4e-7 for the fp32 instructions
// (profiles/tools/probe/wgrad_split_probe.py).  cfg 2 step 0.3725 -> 0.3558 ms.
constexpr int WGS_ROWS = 32;
constexpr int WGS_COL_BYTES = 64;
__device__ __forceinline__ int wgs_pi(int x) { return (0x78 >> (2 * x)) & 3; }           // (0, 2, 3, 1)
__device__ __forceinline__ f32x4 wg_mma16(const wg_v4u& x, const wg_v4u& y, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, x), __builtin_bit_cast(wg_bf16x8, y), c, 0, 0, 0);
}

constexpr int WGS_LIVE_MAX = 1024;                     // live sequences a split's window may hold (LDS, one int each)
constexpr size_t WGS_LDS_FIXED = (size_t)6 * 128 * WGS_COL_BYTES + (size_t)16 * 128 * sizeof(float);      // + the live window's ints

// the rows a launch walks: M rows per domain in `splits` ranges, or -- row_domain != nullptr -- only the live sequences' (see WgradArgs)
struct WgsRows { int M, splits, rows_per_split; const long long* row_domain; int B, T; };
// optional: the X operand is a LayerNorm output that was not stored (seq_fwd.h SeqLayer::ln_stat) -- xin points at the LayerNorm's INPUT rows,
// stat at the rows' (mean, rstd) pairs (row stride stat_ld floats), gam / bet at the gain and bias: the staged value is
// (x - mean) rstd gam + bet, the forward's own expression
struct WgsLn { const float* stat; int stat_ld; const float* gam; const float* bet; };

// One workgroup (eight waves) = one 128 x 128 tile over the rows of (domain g, split): acc = this wave's 16 x 128 block (rows 16 w ..,
// MFMA C layout, eight column tiles); the column sums of dY go to bias_out [128].  dy / xin point at the tile's first column.
template <int NTERM, bool HINT>
__device__ __forceinline__ void wgrad_split_tile(float* smem, const float* __restrict__ dy, int ldy, const float* __restrict__ xin, int ldx,
                                                 int g, int split, const WgsRows& a, f32x4 (&acc)[8], float* __restrict__ bias_out,
                                                 const WgsLn ln = WgsLn{nullptr, 0, nullptr, nullptr}) {
    constexpr int D = 128;
    static_assert(NTERM == 6 || NTERM == 9 || NTERM == 1, "piece pairs kept (1: the hi pieces only -- bf16 products, compute = \"bf16\" on the folded step)");
    constexpr int NTn = D / 16;
    constexpr int PLANE = D * WGS_COL_BYTES;
    char* const Yt = reinterpret_cast<char*>(smem);                // planes hi, mid, lo
    char* const Xt = Yt + 3 * PLANE;
    float* const scratch = reinterpret_cast<float*>(Xt + 3 * PLANE);                       // [16][D] bias-sum scratch
    int* live = reinterpret_cast<int*>(scratch + 16 * D);
    const int w = wave_id(), lane = lane_id();
    int local_beg = split * a.rows_per_split;
    int local_end = min(a.M, local_beg + a.rows_per_split);
    int sq0 = 0;
    if constexpr (HINT) {                // wave 0 counts the domain's live sequences, then lists the window of them this split walks (as sas_wgrad_kernel)
        __shared__ int hd[3];
        if (w == 0) {
            int nl = 0;
            for (int c = 0; c < a.B; c += 64) nl += __popcll(__ballot(c + lane < a.B && ((a.row_domain[c + lane] != 0 ? 1 : 0) == g)));
            const int mv = nl * a.T, rps = (mv + a.splits - 1) / a.splits;
            const int lb = min(mv, split * rps), le = min(mv, lb + rps);
            const int s0 = lb / a.T, s1 = le > lb ? (le - 1) / a.T : s0 - 1;
            int n = 0;
            for (int c = 0; c < a.B && n <= s1; c += 64) {
                const int b = c + lane;
                const bool f = b < a.B && ((a.row_domain[b] != 0 ? 1 : 0) == g);
                const unsigned long long m = __ballot(f);
                const int k = n + __popcll(m & ((1ull << lane) - 1ull));
                if (f && k >= s0 && k <= s1) live[k - s0] = b;
                n += __popcll(m);
            }
            if (lane == 0) { hd[0] = lb; hd[1] = le; hd[2] = s0; }
        }
        __syncthreads();
        local_beg = hd[0]; local_end = hd[1]; sq0 = hd[2];
    }
    const int i = lane & 15, gq = lane >> 4;
    const int cq = 4 * w + (lane & 3), rr = lane >> 2;            // column quad 0..31; rows 2 rr, 2 rr + 1 of the chunk
#pragma unroll
    for (int t = 0; t < NTn; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pyA[2], pxA[2], pyB[2], pxB[2];              // two chunks in flight: a chunk's MFMA phase alone is too short to cover a load's latency
    float2 lsA[2], lsB[2];                              // (WgsLn: the rows' statistics travel with them)
    const bool has_ln = ln.stat != nullptr;             // (block-uniform)
    // (WgsLn: the gain and bias of this thread's four columns are read from LDS in every chunk -- the bias-sum scratch is idle until the
    // chunks are done --: held in registers next to the statistics they took the kernel over its 128 and 24 bytes per lane into scratch)
    float* const lnv = scratch;                         // [2][D]
    const float inv_T = HINT ? 1.0f / (float)a.T : 0.f;
    // every load of chunk c0, unconditionally and branch-free (a branch around a load makes the compiler wait for it on the spot): rows
    // beyond the split's range re-read its last row and are zeroed when the chunk is staged
    auto fetch = [&](int c0, float4 (&py)[2], float4 (&px)[2], float2 (&ls)[2]) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int v = min(c0 + 2 * rr + k, local_end - 1);
            long long row = (long long)g * a.M + v;
            if constexpr (HINT) {                      // virtual row -> (live sequence, position) -> row of the domain
                int sq = (int)(((float)v + 0.5f) * inv_T), pos = v - sq * a.T;          // v < 2^23: off by one at most
                const int lo = pos < 0 ? 1 : 0, hi = pos >= a.T ? 1 : 0;
                sq += hi - lo; pos += (lo - hi) * a.T;
                row = (long long)g * a.M + (long long)live[sq - sq0] * a.T + pos;
            }
            py[k] = ld4(dy + row * ldy + 4 * cq);
            px[k] = ld4(xin + row * ldx + 4 * cq);
            ls[k] = has_ln ? *reinterpret_cast<const float2*>(ln.stat + row * ln.stat_ld) : make_float2(0.f, 1.f);
        }
    };
    // this thread's 4-byte slot in column 4 cq (+ j columns): k-group rr >> 2 of the column, swizzled (columns 4 cq .. 4 cq + 3 share pi)
    const int wr_off = 4 * cq * WGS_COL_BYTES + 16 * ((rr >> 2) ^ wgs_pi(cq & 3)) + 4 * (rr & 3);
    // A column is 64 bytes = 16 banks, so column n lies in bank group n & 3: if every lane wrote its column 4 cq + j in step j, the four column
    // quads of a wave would meet in ONE group -- 32 lanes of a pass on 16 banks, every staging write 2-way (23 % of the LDS cycles of the
    // launch were conflicts).  Lane (cq & 3 = c) writes its columns in the order j + c instead: the pass covers 32 distinct banks (round 6).
    const int wrot = lane & 3;
    int wr_j[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wr_j[j] = wr_off + ((j + wrot) & 3) * WGS_COL_BYTES;
    auto rot4 = [&](const float4& v) {                  // component j of the result = component (j + wrot) & 3 of v
        float4 r = v;
        if (wrot & 1) r = make_float4(r.y, r.z, r.w, r.x);
        if (wrot & 2) r = make_float4(r.z, r.w, r.x, r.y);
        return r;
    };
    const int rd_sw = 16 * (gq ^ wgs_pi(i >> 2));
    const int rd_y = (w * 16 + i) * WGS_COL_BYTES + rd_sw, rd_x = i * WGS_COL_BYTES + rd_sw;
    auto chunk = [&](int c0, float4 (&py)[2], float4 (&px)[2], float2 (&ls)[2]) {       // stage chunk c0 (in py / px), refill them with chunk c0 + 2, multiply
        __syncthreads();                               // previous chunk fully consumed
        if (has_ln) {
            const float4 lgam = *reinterpret_cast<const float4*>(lnv + 4 * cq), lbet = *reinterpret_cast<const float4*>(lnv + D + 4 * cq);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                px[k] = make_float4((px[k].x - ls[k].x) * ls[k].y * lgam.x + lbet.x, (px[k].y - ls[k].x) * ls[k].y * lgam.y + lbet.y,
                                    (px[k].z - ls[k].x) * ls[k].y * lgam.z + lbet.z, (px[k].w - ls[k].x) * ls[k].y * lgam.w + lbet.w);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // the rows past the split's end, by selection.  (History: zeroing by a multiplication with a 0 / 1 factor -- v_pk_mul_f32 / v_pk_fma_f32 with op_sel on the
            // freshly loaded pairs -- gave wrong sums in 239 of 240 launches whenever two workgroups shared a CU and right ones with
            // one per CU; not explained.  tests/test_gpu_kernels.py repeats the two-per-CU launch against the fp64 product.)
#ifdef AMID_WGS_ZERO_BY_MUL      // diagnostic builds only (profiles/tools/probe/wgrad_opsel_repro.sh): the form that miscompared
            const float keep = (c0 + 2 * rr + k < local_end) ? 1.f : 0.f;
            py[k] = f4scale(py[k], keep); px[k] = f4scale(px[k], keep);
#else
            if (c0 + 2 * rr + k >= local_end) { py[k] = make_float4(0.f, 0.f, 0.f, 0.f); px[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
#endif
        }
        bsum = f4add(bsum, f4add(py[0], py[1]));
        const float4 ry0 = rot4(py[0]), ry1 = rot4(py[1]), rx0 = rot4(px[0]), rx1 = rot4(px[1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const WgSplit2 sy = wg_split3(f4comp_w(ry0, j), f4comp_w(ry1, j));
            const WgSplit2 sx = wg_split3(f4comp_w(rx0, j), f4comp_w(rx1, j));
            char* const yq = Yt + wr_j[j];
            char* const xq = Xt + wr_j[j];
            *reinterpret_cast<unsigned*>(yq) = sy.hi; *reinterpret_cast<unsigned*>(xq) = sx.hi;
            if constexpr (NTERM != 1) {
                *reinterpret_cast<unsigned*>(yq + PLANE) = sy.mid; *reinterpret_cast<unsigned*>(yq + 2 * PLANE) = sy.lo;
                *reinterpret_cast<unsigned*>(xq + PLANE) = sx.mid; *reinterpret_cast<unsigned*>(xq + 2 * PLANE) = sx.lo;
            }
        }
        __syncthreads();
        fetch(c0 + 2 * WGS_ROWS, py, px, ls);           // unconditionally (a skipped refill would cost a register copy and a wait here)
        if constexpr (NTERM == 1) {
            const wg_v4u y0 = *reinterpret_cast<const wg_v4u*>(Yt + rd_y);
#pragma unroll
            for (int t = 0; t < NTn; ++t) {
                const wg_v4u x0 = *reinterpret_cast<const wg_v4u*>(Xt + rd_x + t * 16 * WGS_COL_BYTES);
                acc[t] = wg_mma16(y0, x0, acc[t]);
            }
        } else {
        const wg_v4u y0 = *reinterpret_cast<const wg_v4u*>(Yt + rd_y), y1 = *reinterpret_cast<const wg_v4u*>(Yt + PLANE + rd_y),
                     y2 = *reinterpret_cast<const wg_v4u*>(Yt + 2 * PLANE + rd_y);
#pragma unroll
        for (int t = 0; t < NTn; ++t) {
            wg_v4u x[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) x[p] = *reinterpret_cast<const wg_v4u*>(Xt + rd_x + t * 16 * WGS_COL_BYTES + p * PLANE);
            f32x4 c = acc[t];
            if constexpr (NTERM == 9) { c = wg_mma16(y2, x[2], c); c = wg_mma16(y2, x[1], c); c = wg_mma16(y1, x[2], c); }
            c = wg_mma16(y2, x[0], c); c = wg_mma16(y0, x[2], c); c = wg_mma16(y1, x[1], c);
            c = wg_mma16(y1, x[0], c); c = wg_mma16(y0, x[1], c); c = wg_mma16(y0, x[0], c);
            acc[t] = c;
        }
        }
    };
    if (has_ln && threadIdx.x < 64) {                  // (visible behind the first chunk's barrier)
        const int c = threadIdx.x & 31, which = threadIdx.x >> 5;
        const float4 v = ld4((which ? ln.bet : ln.gam) + 4 * c);
        *reinterpret_cast<float4*>(lnv + which * D + 4 * c) = v;
    }
    if (local_beg < local_end) { fetch(local_beg, pyA, pxA, lsA); fetch(local_beg + WGS_ROWS, pyB, pxB, lsB); }
    for (int c0 = local_beg; c0 < local_end; c0 += 2 * WGS_ROWS) {      // an odd chunk count multiplies one chunk of zeros
        chunk(c0, pyA, pxA, lsA);
        chunk(c0 + WGS_ROWS, pyB, pxB, lsB);
    }
    __syncthreads();
    st4(scratch + rr * D + 4 * cq, bsum);              // [16][D]
    __syncthreads();
    for (int e = threadIdx.x; e < D; e += GEMM_THREADS) {
        float sum = 0.f;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) sum += scratch[k * D + e];
        bias_out[e] = sum;
    }
}

}  // namespace amid
