// Shared by the two builds of the one-launch SASRec encoder forward: sasrec_seq.hip (a wave owns a 16-row strip and all D columns) and
// sasrec_seqn.hip (NS waves share a strip, each owning D / NS columns).  Reference: Log2feats.forward, model_seq.py:371-383.
#pragma once
#include "common.h"
#include "rng.h"
#include "strip_gemm.h"

namespace amid {

struct SeqLayer {
    const float* ln1_w[2]; const float* ln1_b[2]; const float* w_in[2]; const float* b_in[2];
    const float* w_o[2]; const float* b_o[2]; const float* ln2_w[2]; const float* ln2_b[2];
    const float* w1[2]; const float* b1[2]; const float* w2[2]; const float* b2[2];
    float* x;                           // this layer's INPUT rows (written for layers >= 1: the previous layer's output)
    float* qn; float* q; float* k; float* v; float* o; float* stats; float* r; float* y; float* h;
    // optional [2M][4] (seqn_fwd_px_kernel only): with it the layer does NOT store qn and y (two of its nine saved tensors: 13 MB at the
    // headline shape) but the row statistics they are rebuilt from -- (mean, rstd) of LayerNorm 1 over x and of LayerNorm 2 over r; the
    // backward strips rebuild them from x / r anyway, and the weight gradients apply the affine map while staging their operand
    float* ln_stat;
};

struct SeqFwdArgs {
    SeqLayer L[2];
    int n_layers;
    const float* x0;                    // layer 0's input (the gathered rows)
    float* xout;                        // the last layer's output
    const unsigned char* tmq;
    float ln_eps, att_scale, dscale, ffn_scale;
    const StepState* st; int train; unsigned spec;
    // bf16 matrix products (sasrec_seqn.hip only): the projection weights as bf16 fragment images [layer][domain][q, k, v, o, c1, c2]
    // [D][D] written by amid_sas_weights_bf16 for THIS step's weights; nullptr = exact fp32 products
    const unsigned short* w16;
    int w16_planes;                      // 1: bf16 products (operands rounded); 3: fp32 products on three bf16 pieces per operand
    int one_piece;                       // (seqn_fwd_px kernels, three-plane images) bf16 products: only the hi planes and the operands' hi pieces
    // optional (seqn_fwd_px kernels, round 6): the gather K1 as the workgroup's PROLOGUE -- embItemLayerEnhance.forward + Log2feats' position
    // add, embedding dropout and == 0 mask (model_seq.py:27-29, :361-366).  g_table != nullptr: layer 0's input rows are not read from x0 but
    // built from table[g_idx[row]] + g_pos[domain][t] (K1's arithmetic, operation for operation: the same bits) and -- in a forward that a
    // backward follows -- stored to x0 / tmq for it; wave 0 also gathers the sample's g_ni item rows into g_items (the head and the
    // scorer sums read them there), and the launch's last workgroup re-joins StepState::step_done (K1 did).
    const float* g_table; const int* g_idx; const float* g_pos[2]; float* g_items; int g_ni; float g_scale; StepState* g_done;
};

struct SeqGeom {
    int B, T, M;
    unsigned act_bytes, tm_bytes, stats_bytes;
    // the sizes the SAVED tensors' buffer descriptors are built with (seqn_fwd_px_kernel): act_bytes / stats_bytes in a forward that a
    // backward follows; 0 in an inference forward (amid_sas_seq_fwd_split_infer_f32) -- every store of a tensor only a backward would read
    // falls outside its descriptor and is dropped by the memory pipeline: no HBM write, no buffer
    unsigned save_bytes, save_stats_bytes;
    const int* live;                    // as StripGeom::live
};

// workgroup barrier between LDS phases WITHOUT draining the vector-memory queue (__syncthreads() waits for every store in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// 32-bit value of lane group `src` (lanes m + 16 src) to all four groups of the same m: two half-exchanges
__device__ __forceinline__ unsigned bcast_group(unsigned v, int SRC) {
    float a = __builtin_bit_cast(float, v), b = a;
    swap16(a, b);                                          // a: rows (0, 0, 2, 2), b: rows (1, 1, 3, 3)
    float x = (SRC & 1) ? b : a, y = x;
    swap32(x, y);                                          // x: (lo half, lo half), y: (hi half, hi half)
    return __builtin_bit_cast(unsigned, (SRC >> 1) ? y : x);
}


// sasrec_seqn.hip: the N-split kernels.  variant: 0 = auto.  Returns AMID_ERR_UNSUPPORTED when no N-split build covers the shape.
// head != nullptr: the train step's head on the tail of every workgroup (csrc/head_parts.h head_own_rows_body; T 33 ... 64, live sequences, pieces)
struct HeadArgs;
int launch_seqn_fwd(const SeqFwdArgs& a, const SeqGeom& sg, int D, int variant, void* stream, const HeadArgs* head = nullptr);

}  // namespace amid
