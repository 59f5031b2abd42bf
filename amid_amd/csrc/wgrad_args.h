// The argument block of the SASRec weight-gradient launches (sasrec_bwd.hip: fp32 / bf16-rounded builds and the host side; sasrec_wgrad_split.hip:
// the builds on bf16 pieces, a translation unit of its own because it is compiled without the SLP vectorizer).
#pragma once
#include "common.h"
#include "sort_phases.h"

namespace amid {

constexpr int WG_MAX = 12;
struct WgradArgs {
    const float* dy[WG_MAX];   // per layer: dq, dk, dv, dr, dpre1, dpre2   [2M, D]
    const float* xin[WG_MAX];  // per layer: qn, x,  x,  o,  y,     h       [2M, D]
    float* w_part[2];          // per layer: [2][6][splits][D*D]
    float* b_part[2];          // per layer: [2][6][splits][D]
    int M, splits, rows_per_split;
    // optional hint (amid_sas_wgrad_rows_f32): only the sequences b of domain g with (row_domain[b] != 0) == g have non-zero dY rows
    // (the loss masks the other domain of every sample, train_sr.py:205-211); the M = B * T rows of a domain are then walked as
    // n_live * T "virtual" rows -- the live sequences back to back -- and the dead half is never read
    const long long* row_domain; int B, T;
    // optional (amid_sas_wgrad_rows_sort_ln_f32; the six-pair build only): per layer [2M][4] row statistics of a forward that did not store
    // qn and y (seq_fwd.h SeqLayer::ln_stat) -- xin of weights 0 (q) and 4 (conv1) then points at x / r and the operand is rebuilt while staged
    const float* ln_stat[2];
    const float* ln1_w[2][2]; const float* ln1_b[2][2]; const float* ln2_w[2][2]; const float* ln2_b[2][2];      // [layer][domain]
};

constexpr int WG_LIVE_MAX = 1024;      // live sequences a split's window may hold (LDS, one int each)

// the launch of sas_wgrad_split_kernel (mode 2: nine piece pairs, 3: six; rd != nullptr: the sort's last phase rides in a third z-slice --
// mode 3 with the live-row hint only); live_bytes: the live window's share of the dynamic LDS
int launch_sas_wgrad_split(const WgradArgs& a, const SortRider* rd, int n_layers, int mode, size_t live_bytes, void* stream);

}  // namespace amid
