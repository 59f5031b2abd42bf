// K2 on the matrix cores: causal multi-head attention for the SASRec shape (head dim 16, T <= 64), forward and
// backward.  Reference: softmax((q sqrt(1/hd)) k^T + causal(-inf)) -> dropout(p) -> . v inside nn.MultiheadAttention as
// called at model_seq.py:374, and its autograd.  (attention.hip keeps the general VALU kernels: any T, hd 8/16/32, key masks.)
//
// One workgroup per (domain, batch row), one wave per head.  Every product is a chain of v_mfma_f32_16x16x4_f32 on
// 16x16 tiles; operands are loaded straight from global/L2 into their MFMA lane layout (no LDS staging):
//   S   = Qs K^T   A-operand = K rows (lane i = key, k = 4 g + j), B-operand = Qs rows  -> lane (m, g) holds
//                  S[query m][key 4 g + r], r = 0..3, of the tile: a query's keys sit in 4 registers x 4 lane groups x
//                  (T/16) tiles, so the softmax max / sum is 15 in-lane ops + two wavefront shuffles (xor 16, xor 32).
//   O   = P~ V     contraction over keys: the P~ registers of step (tile, r) ARE the B-operand (lane group g supplies key
//                  4 g + r) -- no data movement; the A-operand is V^T, fetched as dwords V[key 4 g + r][d = lane & 15].
//   backward, phase 1 (lanes = queries): dP~ = dO V^T the same way, delta = dO . O (two shuffles), dS = P (dP - delta),
//                  dQ = dS K with dS as B-operand and K^T dwords as A-operand.
//   backward, phase 2 (lanes = keys): S^T = K Qs^T and dP~^T = V dO^T recomputed in the transposed layout, so that
//                  dV = P~^T dO and dK = dS^T Qs again take their own registers as B-operand.  Row statistics
//                  (max, 1/sum, delta) and the 64-bit dropout keep word of every query row travel through LDS.
// Causality skips the tiles above the diagonal (10 of 16 tiles at T = 50..64).
// Dropout: p = 0.5 needs ONE Philox call per query row (1 bit per key, rng.h); lane (m, g) computes the row 16 g + m,
// i.e. a single call sequence per wave covers all 64 rows; a row's word reaches its query tile with two shuffles.
#include "attention_mfma.h"

namespace amid {

// a wave's unit of work: a head of 16 dims, or a 16-column tile = two heads of 8 dims (attention_mfma.h PAIR)
__device__ __forceinline__ bool attn_pairs(const AttnArgs& a) { return a.D / a.H == 8; }
__device__ __forceinline__ int attn_units(const AttnArgs& a) { return attn_pairs(a) ? a.H / 2 : a.H; }

__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(const AttnArgs a) {
    const int hw = blockDim.x >> 6, parts = attn_units(a) / hw;
    int seq = blockIdx.x / parts;
    const int part = blockIdx.x - seq * parts;
    if (a.live != nullptr) seq = (seq >= a.live[a.B] ? a.B : 0) + a.live[seq];      // slot -> (g, b) of the live list
    const int g = seq / a.B, b = seq - g * a.B;
    if (attn_pairs(a)) attn_fwd_head<true>(a, g, b, (long long)seq * a.T, part * hw + wave_id());
    else attn_fwd_head<false>(a, g, b, (long long)seq * a.T, part * hw + wave_id());
}

__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, H = a.H;
    // a workgroup = `hw` heads of one sequence (hw = blockDim / 64): with 4 heads per workgroup two workgroups fit a CU at
    // this kernel's ~190 VGPRs, and the second residents of the first wave of workgroups start late (below), so that from then
    // on one workgroup's loads run under the other's MFMAs (one 8-head workgroup per CU ran load -> compute -> load -> compute)
    // (head dim 8: a wave per HEAD here -- the two heads of a tile run their passes on two waves, each loading the tile)
    const int hw = blockDim.x >> 6, parts = H / hw;
    const int slot = blockIdx.x / parts, part = blockIdx.x - slot * parts;
    bool live = true;
    const int seq = a.live != nullptr ? (slot >= a.live[a.B] ? a.B : 0) + a.live[slot]
                  : a.row_domain != nullptr ? live_rows_remap(a.row_domain, a.B, slot, live) : slot;
    const int g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int wv = wave_id(), unit = part * hw + wv, lane = lane_id();
    const bool pairs = attn_pairs(a);
    const int h = pairs ? unit >> 1 : unit, e_only = pairs ? unit & 1 : -1;       // column tile; the tile's head in pair mode
    float* lds = smem + wv * (ATTN_BWD_LDS_PER_WAVE / 4);                                 // this wave's scratch block (attention_mfma.h)
    if (!live) {                                                                       // no gradient reaches this sequence: exact zeros
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e_only <= 0) {                                                             // (pair mode: the tile's first wave writes its zeros)
            for (int i = lane; i < T * (AHD / 4); i += 64) {
                const long long off = (rowbase + i / (AHD / 4)) * D + h * AHD + 4 * (i % (AHD / 4));
                st4(a.dq + off, z); st4(a.dk + off, z); st4(a.dv + off, z);
            }
        }
        return;
    }
    if (a.stagger_from >= 0 && (int)blockIdx.x >= a.stagger_from && (int)blockIdx.x < 2 * a.stagger_from) {
#pragma unroll 1
        for (int i = 0; i < a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);       // 127 x 64 clocks each
    }

    if (pairs) {
        switch ((T + 15) >> 4) {
            case 1: attn_bwd_head<1, true>(a, g, b, rowbase, h, lds, e_only); break;
            case 2: attn_bwd_head<2, true>(a, g, b, rowbase, h, lds, e_only); break;
            case 3: attn_bwd_head<3, true>(a, g, b, rowbase, h, lds, e_only); break;
            default: attn_bwd_head<4, true>(a, g, b, rowbase, h, lds, e_only); break;
        }
        return;
    }
    switch ((T + 15) >> 4) {
        case 1: attn_bwd_head<1>(a, g, b, rowbase, h, lds); break;
        case 2: attn_bwd_head<2>(a, g, b, rowbase, h, lds); break;
        case 3: attn_bwd_head<3>(a, g, b, rowbase, h, lds); break;
        default: attn_bwd_head<4>(a, g, b, rowbase, h, lds); break;
    }
}

}  // namespace amid

using namespace amid;

// called by the entry points in attention.hip when the shape fits (causal, head dim 16, T <= 64, no key mask)
static int cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount; else n_cu = 256; }
    return n_cu;
}

int amid_attn_mfma_fwd_launch(const void* args, void* stream) {
    AttnArgs a = *(const AttnArgs*)args;
    const int units = (a.D / a.H == 8) ? a.H / 2 : a.H;        // waves per sequence: heads of 16 dims, or pairs of heads of 8
    const int hw = (units % 4 == 0) ? 4 : units;               // ... per workgroup
    const int parts = units / hw, grid = (a.live != nullptr ? 1 : 2) * a.B * parts;
    a.stagger_from = -1;                                       // measured: any stagger only delays the forward kernel (20.6 -> 22.5+ us)
    a.stagger_sleeps = 0;
    attn_fwd_mfma_kernel<<<grid, hw * 64, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

int amid_attn_mfma_bwd_launch(const void* args, void* stream) {
    AttnArgs a = *(const AttnArgs*)args;
    const int units = a.H;                                     // a wave per head (head dim 8: the two heads of a tile on two waves)
    const int hw = (units % 4 == 0) ? 4 : units;               // waves per workgroup
    const int parts = units / hw, grid = (a.live != nullptr ? 1 : 2) * a.B * parts;
    // workgroups are handed out one per CU first, so with more than 256 of them [256, 512) are the second residents of the CUs:
    // they wait ~8 us (their neighbour's load phase) before requesting their own operands
    const int n_cu = cu_count();
    a.stagger_from = (grid >= 2 * n_cu) ? n_cu : -1;
    // measured at B 256, T 50 (4 key tiles) with the first backward kernel (two passes, 80 loads per head): 0 sleeps: 52.2 us, 1: 50.4,
    // 2: 48.2, 3: 53.5.  The one-pass kernel's load phase is a third as long: the wait no longer pays (live launch in the step:
    // 0 sleeps 0.3875 ms/step, 1: 0.3894, 2: 0.3891) -- kept as a mechanism, set to 0
    a.stagger_sleeps = 0;
    const size_t lds = (size_t)hw * ATTN_BWD_LDS_PER_WAVE;
    attn_bwd_mfma_kernel<<<grid, hw * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
