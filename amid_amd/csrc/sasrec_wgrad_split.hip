// The SASRec weight gradients on bf16 pieces (mma mode 2 / 3: every fp32 operand as hi + mid + lo, nine / six piece pairs, fp32 accuracy --
// csrc/wgrad_split.h).  Reference: the weight gradients of Log2feats.forward's twelve projections (model_seq.py:371-383) under
// loss.backward(), train_sr.py:214.  Split out of sasrec_bwd.hip in round 5: this unit is compiled WITHOUT the SLP vectorizer (Makefile
// FLAGS_sasrec_wgrad_split; packed fp32 math in the staging arithmetic miscompared under co-residency, DESIGN.md section 5.0 "The wgrad
// finding"), and that flag cost the row-tile kernels of sasrec_bwd.hip 128 bytes per lane of scratch.
#include "common.h"
#include "rng.h"
#include "tile_gemm.h"
#include "wgrad_split.h"
#include "sort_phases.h"
#include "wgrad_args.h"

namespace amid {

// the same on the bf16 matrix cores at fp32 accuracy (mma mode 2 / 3): csrc/wgrad_split.h
// RIDER: the launch has one more z-slice whose first rd.plan.nblk workgroups run the LAST phase of the step's index sort (sort_phases.h:
// run heads; it rode in the embedding-backward launch while the live-sequence step had one): waves 0 .. 3 of a 512-thread workgroup --
// the others leave at once (a barrier counts the waves that have not ended)
template <int D, int NTERM, bool HINT, bool RIDER = false>
__global__ __launch_bounds__(GEMM_THREADS, 4) void sas_wgrad_split_kernel(const WgradArgs a, const SortRider rd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(D == 128, "eight waves = eight 16-row tiles of dW");
    if constexpr (RIDER) {
        if (blockIdx.z == 2) {
            const int rb = blockIdx.y * gridDim.x + blockIdx.x;
            if (rb < rd.plan.nblk && threadIdx.x < SORT_THREADS) sort_phase_ct<5>(rd.plan, rb, smem);
            return;
        }
    }
    const int split = blockIdx.x, wsel = blockIdx.y, g = blockIdx.z;
    const int layer = wsel / 6, wi = wsel - layer * 6;
    const WgsRows rw{a.M, a.splits, a.rows_per_split, a.row_domain, a.B, a.T};
    WgsLn ln{nullptr, 0, nullptr, nullptr};
    if (a.ln_stat[layer] != nullptr && (wi == 0 || wi == 4))       // (block-uniform) q: LN1 over x; conv1: LN2 over r
        ln = wi == 0 ? WgsLn{a.ln_stat[layer], 4, a.ln1_w[layer][g], a.ln1_b[layer][g]} : WgsLn{a.ln_stat[layer] + 2, 4, a.ln2_w[layer][g], a.ln2_b[layer][g]};
    f32x4 acc[8];
    wgrad_split_tile<NTERM, HINT>(smem, a.dy[wsel], D, a.xin[wsel], D, g, split, rw, acc,
                                  a.b_part[layer] + (((long long)g * 6 + wi) * a.splits + split) * D, ln);
    const int w = wave_id(), lane = lane_id(), i = lane & 15, gq = lane >> 4;
    float* wp = a.w_part[layer] + (((long long)g * 6 + wi) * a.splits + split) * D * D;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[(long long)(w * 16 + gq * 4 + r) * D + t * 16 + i] = acc[t][r];
}

int launch_sas_wgrad_split(const WgradArgs& a, const SortRider* rdp, int n_layers, int mode, size_t live_bytes, void* stream) {
    SortRider rd;
    rd.phase = 0;
    const dim3 grid(a.splits, 6 * n_layers, 2);
    if (rdp != nullptr) {
        rd = *rdp;
        if (mode == 4) {          // ONE piece per operand: bf16 products (compute = "bf16" on the folded step)
            static unsigned long long done_r1 = 0;
            if (int e = lds_attr_once((const void*)sas_wgrad_split_kernel<128, 1, true, true>, WGS_LDS_FIXED + WG_LIVE_MAX * sizeof(int), done_r1)) return e;
            sas_wgrad_split_kernel<128, 1, true, true><<<dim3(a.splits, 6 * n_layers, 3), GEMM_THREADS, WGS_LDS_FIXED + live_bytes, (hipStream_t)stream>>>(a, rd);
            AMID_LAUNCH_CHECK();
            return AMID_OK;
        }
        static unsigned long long done_r = 0;
        if (int e = lds_attr_once((const void*)sas_wgrad_split_kernel<128, 6, true, true>, WGS_LDS_FIXED + WG_LIVE_MAX * sizeof(int), done_r)) return e;
        sas_wgrad_split_kernel<128, 6, true, true><<<dim3(a.splits, 6 * n_layers, 3), GEMM_THREADS, WGS_LDS_FIXED + live_bytes, (hipStream_t)stream>>>(a, rd);
        AMID_LAUNCH_CHECK();
        return AMID_OK;
    }
#ifdef AMID_WGS_LDS_GUARD        // diagnostic builds (profiles/tools/probe/wgrad_opsel_repro.sh): unused LDS behind every workgroup's allocation
    const size_t fixed = WGS_LDS_FIXED + AMID_WGS_LDS_GUARD;
#else
    const size_t fixed = WGS_LDS_FIXED;
#endif
    static unsigned long long done[4] = {0, 0, 0, 0};
#define AMID_WGS_LAUNCH(NT, H, SLOT)                                                                                                  \
    do {                                                                                                                          \
        if (int e = lds_attr_once((const void*)sas_wgrad_split_kernel<128, NT, H>, fixed + WG_LIVE_MAX * sizeof(int), done[SLOT])) return e; \
        sas_wgrad_split_kernel<128, NT, H><<<grid, GEMM_THREADS, fixed + live_bytes, (hipStream_t)stream>>>(a, rd);                   \
    } while (0)
    const bool h = a.row_domain != nullptr;
    if (mode == 2) { if (h) AMID_WGS_LAUNCH(9, true, 0); else AMID_WGS_LAUNCH(9, false, 1); }
    else if (mode == 3) { if (h) AMID_WGS_LAUNCH(6, true, 2); else AMID_WGS_LAUNCH(6, false, 3); }
    else return AMID_ERR_ARG;
#undef AMID_WGS_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

}  // namespace amid
