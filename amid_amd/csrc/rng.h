// Counter-based dropout RNG shared by every kernel (and restated in
// oracle/amid_oracle.py: philox_keep_flat).  Philox4x32-10, key = 64-bit seed,
// counter = (call_lo, call_hi, site, step).  One call serves 8 consecutive
// elements: element e uses 16-bit half (e & 1) of word ((e >> 1) & 3) of call
// (e >> 3); keep <=> half >= thr16, thr16 = round(p * 65536).
#pragma once
#include "common.h"

namespace amid {

struct StepState {         // lives in device memory so that hipGraph replays see fresh values
    unsigned long long seed;   // dropout seed
    long long step;            // 1-based global step t: dropout counter and Adam bias correction; bumped by amid_step_begin
    double lr, beta1, beta2, eps;   // torch.optim.Adam hyper-parameters (train_sr.py:480: lr only, rest defaults)
};
using RngState = StepState;

enum Site { SITE_EMB = 0, SITE_ATTN = 1, SITE_FFN1 = 2, SITE_FFN2 = 3, SITE_SUB_IN = 4, SITE_SUB_OUT = 5, SITE_BLOCK = 6 };
__host__ __device__ __forceinline__ unsigned site_id(int domain, int layer, int kind) { return (unsigned)((domain * 2 + layer) * 8 + kind); }

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// random words for elements [8*call, 8*call+8)
__device__ __forceinline__ uint4 rng_call(unsigned long long seed, unsigned long long call, unsigned site, unsigned step) {
    return philox4x32_10(make_uint4((unsigned)call, (unsigned)(call >> 32), site, step), (unsigned)seed, (unsigned)(seed >> 32));
}
__device__ __forceinline__ unsigned rng_half(uint4 r, int e_in_call) {   // e_in_call in [0,8)
    unsigned w = (e_in_call >> 1) == 0 ? r.x : (e_in_call >> 1) == 1 ? r.y : (e_in_call >> 1) == 2 ? r.z : r.w;
    return (e_in_call & 1) ? (w >> 16) : (w & 0xFFFFu);
}
__host__ __device__ __forceinline__ unsigned keep_thr16(float p) {
    float t = p * 65536.0f + 0.5f;
    unsigned u = (unsigned)t;
    return u > 0xFFFFu ? 0xFFFFu : u;
}

// keep flags (as 0/scale multipliers) for 4 consecutive elements starting at e0 (e0 % 4 == 0)
__device__ __forceinline__ float4 dropout_mult4(unsigned long long seed, unsigned site, unsigned step, unsigned long long e0,
                                                unsigned thr16, float scale) {
    uint4 r = rng_call(seed, e0 >> 3, site, step);
    unsigned w0 = (e0 & 4) ? r.z : r.x, w1 = (e0 & 4) ? r.w : r.y;
    float4 m;
    m.x = ((w0 & 0xFFFFu) >= thr16) ? scale : 0.f;
    m.y = ((w0 >> 16) >= thr16) ? scale : 0.f;
    m.z = ((w1 & 0xFFFFu) >= thr16) ? scale : 0.f;
    m.w = ((w1 >> 16) >= thr16) ? scale : 0.f;
    return m;
}

}  // namespace amid
