// Scalar arithmetic of the BERT4Rec block shared by its row-tile kernels (bert.hip) and its strip kernels (bert_strip.hip).
#pragma once
#include "common.h"

namespace amid {

constexpr float BERT_EPS = 1e-6f;      // the reference LayerNorm adds eps to the (unbiased) std, model_seq.py:124-127

// tanh GELU (model_seq.py:204) through the identity 0.5 (1 + tanh u) = sigmoid(2u): one v_exp_f32 and one v_rcp_f32 per element,
// no cancellation anywhere (libm's tanhf is ~40 instructions over two divergent branches; the feed-forward kernels evaluate
// 13 M of these per launch).  gelu'(x) = s + 2 x s (1 - s) u'(x) with s = sigmoid(2u), since 1 - tanh^2 u = 4 s (1 - s).
__device__ __forceinline__ float gelu_sig(float x) {
    const float u2 = 1.5957691216057308f * (x + 0.044715f * x * x * x);         // 2 u
    return __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_sig(x); }
__device__ __forceinline__ float gelu_df(float x) {
    const float sg = gelu_sig(x);
    return sg + 2.0f * x * sg * (1.0f - sg) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

}  // namespace amid
