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

__global__ __launch_bounds__(512) void attn_fwd_mfma_kernel(const AttnArgs a) {
    const int hw = blockDim.x >> 6, parts = a.H / hw;
    int seq = blockIdx.x / parts;
    const int part = blockIdx.x - seq * parts;
    if (a.live != nullptr) seq = (seq >= a.live[a.B] ? a.B : 0) + a.live[seq];      // slot -> (g, b) of the live list
    const int g = seq / a.B, b = seq - g * a.B;
    attn_fwd_head(a, g, b, (long long)seq * a.T, part * hw + wave_id());
}

__global__ __launch_bounds__(512) void attn_bwd_mfma_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, H = a.H;
    // a workgroup = `hw` heads of one sequence (hw = blockDim / 64): with 4 heads per workgroup two workgroups fit a CU at
    // this kernel's ~190 VGPRs, and the second residents of the first wave of workgroups start late (below), so that from then
    // on one workgroup's loads run under the other's MFMAs (one 8-head workgroup per CU ran load -> compute -> load -> compute)
    const int hw = blockDim.x >> 6, parts = H / hw;
    const int slot = blockIdx.x / parts, part = blockIdx.x - slot * parts;
    bool live = true;
    const int seq = a.live != nullptr ? (slot >= a.live[a.B] ? a.B : 0) + a.live[slot]
                  : a.row_domain != nullptr ? live_rows_remap(a.row_domain, a.B, slot, live) : slot;
    const int g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int wv = wave_id(), h = part * hw + wv, lane = lane_id();
    const int m = lane & 15, gq = lane >> 4;
    const int NT = (T + 15) >> 4;
    const int col4 = h * AHD + 4 * gq, colm = h * AHD + m;
    float4* rstat = reinterpret_cast<float4*>(smem) + wv * 64;                          // [hw][64] (max, 1/sum, delta, -)
    unsigned long long* keepw = reinterpret_cast<unsigned long long*>(smem + hw * 64 * 4) + wv * 64;   // [hw][64]
    if (!live) {                                                                       // no gradient reaches this sequence: exact zeros
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = lane; i < T * (AHD / 4); i += 64) {
            const long long off = (rowbase + i / (AHD / 4)) * D + h * AHD + 4 * (i % (AHD / 4));
            st4(a.dq + off, z); st4(a.dk + off, z); st4(a.dv + off, z);
        }
        return;
    }
    if (a.stagger_from >= 0 && (int)blockIdx.x >= a.stagger_from && (int)blockIdx.x < 2 * a.stagger_from) {
#pragma unroll 1
        for (int i = 0; i < a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);       // 127 x 64 clocks each
    }

    // Every operand of BOTH phases is requested here, up front (one exposure of the memory latency per workgroup instead of
    // two): row fragments of Q, K, V, dO, O (float4 along the head dim) and the transposed dword fragments K^T (phase 1), Q^T,
    // dO^T (phase 2).  The first version reloaded Q, dO, K, V after the barrier between the phases.
    float4 kfr[4], vfr[4], qfr[4], dofr[4];
    float qts[4][4], dots[4][4];
    {
        float kt[4][4];
        float4 ofr[4];
        float2 str[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            kfr[t] = ld4_row(a.k, rowbase, t * 16 + m, T, D, col4);
            vfr[t] = ld4_row(a.v, rowbase, t * 16 + m, T, D, col4);
            qfr[t] = f4scale(ld4_row(a.q, rowbase, t * 16 + m, T, D, col4), a.scale);
            dofr[t] = ld4_row(a.d_o, rowbase, t * 16 + m, T, D, col4);
            ofr[t] = ld4_row(a.o, rowbase, t * 16 + m, T, D, col4);
            str[t] = *reinterpret_cast<const float2*>(a.stats + ((rowbase + min(t * 16 + m, T - 1)) * H + h) * 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) kt[t][r] = ld1_row(a.k, rowbase, t * 16 + 4 * gq + r, T, D, colm);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                qts[t][r] = ld1_row(a.q, rowbase, t * 16 + 4 * gq + r, T, D, colm) * a.scale;
                dots[t][r] = ld1_row(a.d_o, rowbase, t * 16 + 4 * gq + r, T, D, colm);
            }
        // ---------------- phase 1: lanes = queries -> dQ; row stats + keep words to LDS ----------------
        unsigned long long kw_own = ~0ull;
        if (a.train) {
            const int qrow = min(gq * 16 + m, T - 1);
            kw_own = row_keep_word(a.st->seed, site_id(g, a.layer, SITE_ATTN), (unsigned)a.st->step,
                                   (unsigned long long)(b * H + h) * T + qrow, T, a.thr16);
        }
        keepw[lane] = kw_own;                                                            // row index = 16 gq + m = lane
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            if (qi >= NT) break;
            const int q = qi * 16 + m;
            const float4 qf = qfr[qi], dof = dofr[qi], of = ofr[qi];
            const float delta = quad_group_sum(f4hsum(f4mul(dof, of)));
            const float mrow = str[qi].x, rl = str[qi].y;
            if (gq == 0) rstat[q] = make_float4(mrow, rl, delta, 0.f);
            const unsigned long long kw = shfl64(kw_own, qi * 16 + m);
            f32x4 dq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kj = 0; kj < 4; ++kj) {
                if (kj <= qi) {
                    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
                    s = mfma_frag(kfr[kj], qf, s);
                    dp = mfma_frag(vfr[kj], dof, dp);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = kj * 16 + 4 * gq + r;
                        const float p = (n > q) ? 0.f : fast_exp(s[r] - mrow) * rl;
                        const float dpk = ((kw >> n) & 1ull) ? dp[r] * a.dscale : 0.f;
                        const float ds = p * (dpk - delta);
                        dq = mfma4(kt[kj][r], ds, dq);
                    }
                }
            }
            if (q < T) st4(a.dq + (rowbase + q) * D + col4, make_float4(dq[0] * a.scale, dq[1] * a.scale, dq[2] * a.scale, dq[3] * a.scale));
        }
    }
    // no workgroup barrier here: rstat / keepw of a wave are written and read by that wave only (LDS operations of one wave
    // complete in order), and without it the waves of a workgroup drift apart -- the late ones' loads overlap the early ones' MFMAs
    // ---------------- phase 2: lanes = keys -> dK, dV (operands already in registers) ----------------
#pragma unroll
    for (int kj = 0; kj < 4; ++kj) {
        if (kj >= NT) break;
        const int key = kj * 16 + m;
        const float4 kf = kfr[kj], vf = vfr[kj];
        f32x4 dk = f32x4{0.f, 0.f, 0.f, 0.f}, dv = dk;
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            if (qi < kj || qi >= NT) continue;
            const float4 qf = qfr[qi], dof = dofr[qi];
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dpt = st;
            st = mfma_frag(qf, kf, st);                  // S^T: lane (key m, gq), reg r <-> query qi*16 + 4 gq + r
            dpt = mfma_frag(dof, vf, dpt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = qi * 16 + 4 * gq + r;
                const float qt = qts[qi][r], dot = dots[qi][r];
                const float4 rs = rstat[min(qq, 63)];
                const bool live = (qq < T) && (key <= qq);
                const bool keep = (keepw[min(qq, 63)] >> key) & 1ull;
                const float p = live ? fast_exp(st[r] - rs.x) * rs.y : 0.f;
                const float pd = keep ? p * a.dscale : 0.f;
                const float dpk = keep ? dpt[r] * a.dscale : 0.f;
                const float ds = p * (dpk - rs.z);
                dv = mfma4(dot, pd, dv);
                dk = mfma4(qt, ds, dk);
            }
        }
        if (key < T) {
            st4(a.dk + (rowbase + key) * D + col4, make_float4(dk[0], dk[1], dk[2], dk[3]));
            st4(a.dv + (rowbase + key) * D + col4, make_float4(dv[0], dv[1], dv[2], dv[3]));
        }
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
    const int hw = (a.H % 4 == 0) ? 4 : a.H;                   // heads per workgroup
    const int parts = a.H / hw, grid = (a.live != nullptr ? 1 : 2) * a.B * parts;
    a.stagger_from = -1;                                       // measured: any stagger only delays the forward kernel (20.6 -> 22.5+ us)
    a.stagger_sleeps = 0;
    attn_fwd_mfma_kernel<<<grid, hw * 64, 0, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

int amid_attn_mfma_bwd_launch(const void* args, void* stream) {
    AttnArgs a = *(const AttnArgs*)args;
    const int hw = (a.H % 4 == 0) ? 4 : a.H;                   // heads per workgroup
    const int parts = a.H / hw, grid = (a.live != nullptr ? 1 : 2) * a.B * parts;
    // workgroups are handed out one per CU first, so with more than 256 of them [256, 512) are the second residents of the CUs:
    // they wait ~8 us (their neighbour's load phase) before requesting their own operands
    const int n_cu = cu_count();
    a.stagger_from = (grid >= 2 * n_cu) ? n_cu : -1;
    // measured at B 256, T 50 (4 key tiles): 0 sleeps: 52.2 us, 1: 50.4, 2: 48.2, 3: 53.5, 4: 55.6; at T 20 (2 tiles) the neighbour's load
    // phase is too short to be worth waiting for: 0: 22.2 us, 1: 23.8, 2: 25.7
    const int nt = (a.T + 15) >> 4;
    a.stagger_sleeps = nt >= 4 ? 2 : nt == 3 ? 1 : 0;
    const size_t lds = (size_t)hw * 64 * (16 + 8);
    attn_bwd_mfma_kernel<<<grid, hw * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
