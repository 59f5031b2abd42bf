// Arguments of the strip backward chains and of the one-launch backward of a sequence (sasrec_strip.hip: seq_bwd_kernel, a wave per
// 16-row strip; sasrec_seqn_bwd.hip: seqn_bwd_kernel, the N-split build -- two waves share a strip, each owning half the columns).
#pragma once
#include "common.h"
#include "strip_gemm.h"
#include "attention_mfma.h"

namespace amid {

struct StripFfnBwdArgs {
    const float* dxo;                   // [2M, D] gradient of the layer output
    const unsigned char* tmq;
    const float* h; const float* r;     // saved relu output, saved LN2 input
    const float* ln_w[2];
    const float* w1T[2]; const float* w2T[2]; const float* woT[2];
    float* dpre2; float* dpre1; float* dr; float* d_o;
    float* ln_part;                     // [2 tpg][2][D]
    float ln_eps;
    const StepState* st; int train; unsigned spec; float scale; int layer;
};

struct StripQkvBwdArgs {
    const float* dq; const float* dk; const float* dv; const float* dr; const float* x;
    const float* ln_w[2];
    const float* wqT[2]; const float* wkT[2]; const float* wvT[2];
    float* dx;
    float* ln_part;                     // [2 tpg][2][D]
    float ln_eps;
    // layer 0's launch of the live-sequence train step (strip_qkv_bwd_kernel without the fused feed-forward): d x is the gradient of the
    // encoder INPUT -- with emb_tmq set the embedding layer's own backward runs on the strip before it is stored (the input dropout's keep
    // bits redrawn, the "== 0" mask, model_seq.py:361-366): what amid_embed_bwd_f32 did in a pass of its own over the stored rows
    const unsigned char* emb_tmq; const StepState* emb_st; int emb_train; unsigned emb_spec; float emb_scale;
};

struct SeqBwdLayer {
    StripFfnBwdArgs f;                  // f.dxo: the top layer's only; f.ln_part / a.ln_part: [2 B][2][D], slot g * B + (index in the domain's live list)
    StripQkvBwdArgs a;                  // a.dx: layer 0's only
    AttnArgs at;
};
struct SeqBwdArgs { SeqBwdLayer L[2]; int n_layers; };

// the N-split build of the one-launch backward (sasrec_seqn_bwd.hip); AMID_ERR_UNSUPPORTED when it does not cover the arguments
int launch_seqn_bwd(const SeqBwdArgs& a, const StripGeom& sg, int D, int mma_bf16, void* stream);

}  // namespace amid
