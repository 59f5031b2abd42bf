// Counter-based dropout RNG shared by every kernel (and restated in
// oracle/amid_oracle.py: philox_keep_flat).  Philox4x32-10, key = 64-bit seed,
// counter = (call_lo, call_hi, site, step).  How the 128 bits of a call are cut
// into keep/drop decisions: "decision format" below.
#pragma once
#include "common.h"

namespace amid {

struct StepState {         // lives in device memory so that hipGraph replays see fresh values
    unsigned long long seed;   // dropout seed
    long long step;            // 1-based global step t: dropout counter and Adam bias correction; bumped by amid_step_begin
    double lr, beta1, beta2, eps;   // torch.optim.Adam hyper-parameters (train_sr.py:480: lr only, rest defaults)
    unsigned ticket, pad_;     // "last block out" counter of the pool-input pack kernel (embed.hip); wraps to 0 by itself
    long long step_done;       // steps completed: equals `step` whenever a step's first launch starts.  The one-launch step head (adam.hip
                               // step_head_kernel) reads t - 1 HERE while one of its threads writes `step` = t (no block of that launch reads
                               // `step`, so no ticket is needed); the gather K1 behind it copies `step` back here (nobody reads it there)
};
using RngState = StepState;

enum Site { SITE_EMB = 0, SITE_ATTN = 1, SITE_FFN1 = 2, SITE_FFN2 = 3, SITE_SUB_IN = 4, SITE_SUB_OUT = 5, SITE_BLOCK = 6 };
__host__ __device__ __forceinline__ unsigned site_id(int domain, int layer, int kind) { return (unsigned)((domain * 2 + layer) * 8 + kind); }

// (each round's two 32 x 32 -> 64-bit products written as 64-bit multiplies: hipcc then selects ONE v_mad_u64_u32 each -- high and low
// word together -- instead of the v_mul_hi_u32 + v_mul_lo_u32 pair it emits for __umulhi() and a separate product; integer multiplies
// issue at a quarter of the plain rate and are most of a call's cost.  Written as inline asm with the carry-out in vcc the same
// instruction faulted inside the strip kernels' matrix loops -- left to the compiler.)
__device__ __forceinline__ unsigned long long mul_wide(unsigned a, unsigned b) { return (unsigned long long)a * b; }
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = mul_wide(0xD2511F53u, c.x), p1 = mul_wide(0xCD9E8D57u, c.z);
        c = make_uint4((unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// 128 random bits of one call
__device__ __forceinline__ uint4 rng_call(unsigned long long seed, unsigned long long call, unsigned site, unsigned step) {
    return philox4x32_10(make_uint4((unsigned)call, (unsigned)(call >> 32), site, step), (unsigned)seed, (unsigned)(seed >> 32));
}

// ---- decision format ("drop spec") ---------------------------------------------------------------------------------
// A keep/drop decision consumes b bits: the smallest b in {1, 2, 4, 8, 16} for which p * 2^b is an integer (p = 0.5 -> 1
// bit, so ONE Philox call decides 128 elements: a whole D = 128 activation row, or a whole attention row of <= 128 keys);
// otherwise b = 16 with the threshold rounded (p = 0.1 -> 6554 / 65536).  keep <=> field >= thr, thr = p * 2^b.
// Element e uses call e / (128 / b), field e % (128 / b); fields are packed LSB-first in the words x, y, z, w.
// spec = (b << 16) | thr travels as one int.  Restated in oracle/amid_oracle.py (drop_bits / philox_keep_flat).
__host__ __device__ __forceinline__ unsigned drop_spec(float p) {
    for (int b = 1; b <= 8; b <<= 1) {
        const float t = p * (float)(1 << b);
        if (t == (float)(int)t) return ((unsigned)b << 16) | (unsigned)(int)t;
    }
    float t = p * 65536.0f + 0.5f;
    unsigned u = (unsigned)t;
    if (u > 0xFFFFu) u = 0xFFFFu;
    return (16u << 16) | u;
}
__host__ __device__ __forceinline__ unsigned keep_thr16(float p) { return drop_spec(p); }      // historical name at the call sites
__host__ __device__ __forceinline__ int spec_bits(unsigned spec) { return (int)(spec >> 16); }
__host__ __device__ __forceinline__ unsigned spec_thr(unsigned spec) { return spec & 0xFFFFu; }
__host__ __device__ __forceinline__ int spec_per_call(unsigned spec) { return 128 / spec_bits(spec); }

__device__ __forceinline__ unsigned rng_word(uint4 r, int w) { return w == 0 ? r.x : w == 1 ? r.y : w == 2 ? r.z : r.w; }
// field f (0 .. 128/b - 1) of a call
__device__ __forceinline__ unsigned rng_field(uint4 r, int f, int b) {
    const int off = f * b;
    return (rng_word(r, off >> 5) >> (off & 31)) & ((1u << b) - 1u);
}

// keep multipliers (scale or 0) for 4 consecutive elements starting at e0 (e0 % 4 == 0)
__device__ __forceinline__ float4 dropout_mult4(unsigned long long seed, unsigned site, unsigned step, unsigned long long e0,
                                                unsigned spec, float scale) {
    const int b = spec_bits(spec);
    const unsigned thr = spec_thr(spec);
    const int lg_per = 7 - (__ffs(b) - 1);                       // b is a power of two: shifts, not 64-bit divisions
    const uint4 r = rng_call(seed, e0 >> lg_per, site, step);
    const int off = ((int)e0 & ((1 << lg_per) - 1)) * b;         // 4 fields never straddle a word (4 b <= 32 needs b <= 8; b = 16: two words)
    float4 m;
    if (b == 16) {
        const unsigned w0 = rng_word(r, off >> 5), w1 = rng_word(r, (off >> 5) + 1);
        m.x = ((w0 & 0xFFFFu) >= thr) ? scale : 0.f;
        m.y = ((w0 >> 16) >= thr) ? scale : 0.f;
        m.z = ((w1 & 0xFFFFu) >= thr) ? scale : 0.f;
        m.w = ((w1 >> 16) >= thr) ? scale : 0.f;
    } else {
        const unsigned w = rng_word(r, off >> 5) >> (off & 31);
        const unsigned mask = (1u << b) - 1u;
        m.x = ((w & mask) >= thr) ? scale : 0.f;
        m.y = (((w >> b) & mask) >= thr) ? scale : 0.f;
        m.z = (((w >> (2 * b)) & mask) >= thr) ? scale : 0.f;
        m.w = (((w >> (3 * b)) & mask) >= thr) ? scale : 0.f;
    }
    return m;
}

}  // namespace amid
