// The arithmetic of the dense-equivalent lazy Adam (adam.hip) that more than one kernel needs: torch's single-tensor Adam step op by
// op, and the exact replay of a row's pending ZERO-gradient steps -- by the optimizer launch for the rows it updates, by the flush, by
// the catch-up launch, and, in registers only, by the gather K1 (embed.hip) for the value it hands the forward.
// Reference: torch.optim.Adam(model.parameters(), lr) over every parameter, the whole table included (train_sr.py:480, :213-215).
#pragma once
#include "common.h"
#include "rng.h"

namespace amid {

struct AdamCoef {              // per-step scalars of a REAL step, as torch computes them (double -> float at the op)
    float w1;                  // 1 - beta1                     (lerp weight)
    float beta2, w2;           // beta2, 1 - beta2
    float neg_step_size;       // -(lr / (1 - beta1^t))
    float bc2_sqrt;            // sqrt(1 - beta2^t)
    float eps;
};

// beta^s for an integer step count by repeated squaring: <= 2 log2(s) double multiplies (relative error < 1e-14, invisible after
// the coefficients are rounded to float) instead of libm's pow(double, double), several hundred fp64 instructions per call
__device__ __forceinline__ double pow_step(double b, long long s) {
    double r = 1.0;
    while (s > 0) {
        if (s & 1) r *= b;
        b *= b;
        s >>= 1;
    }
    return r;
}

__device__ __forceinline__ AdamCoef adam_coef(const StepState& st, double b1pow, double b2pow) {
    AdamCoef c;
    c.w1 = (float)(1.0 - st.beta1);
    c.beta2 = (float)st.beta2;
    c.w2 = (float)(1.0 - st.beta2);
    c.neg_step_size = (float)(-(st.lr / (1.0 - b1pow)));
    c.bc2_sqrt = (float)sqrt(1.0 - b2pow);
    c.eps = (float)st.eps;
    return c;
}
__device__ __forceinline__ AdamCoef adam_coef_now(const StepState& st) { return adam_coef(st, pow_step(st.beta1, st.step), pow_step(st.beta2, st.step)); }

// (contraction off for the step's own expressions: torch's addcmul_ / addcdiv_ are unfused, and every kernel that inlines this step must
// produce the same bits)
#pragma clang fp contract(off)
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, const AdamCoef& c) {
    m = __fmaf_rn(c.w1, __fsub_rn(g, m), m);                                   // exp_avg.lerp_(grad, 1 - beta1)
    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2): three products and a sum, each rounded -- written with the language's own
    // operators so that this file's `contract(off)` governs them (the __fmul_rn / __fadd_rn wrappers are plain `*` / `+` compiled where THEY
    // are defined, under the default contraction: inlined into two different kernels they came out as an fma in one and not in the other)
    v = (v * c.beta2) + ((c.w2 * g) * g);
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), c.bc2_sqrt), c.eps); // (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
    p = __fadd_rn(p, __fdiv_rn(__fmul_rn(c.neg_step_size, m), denom));          // param.addcdiv_(exp_avg, denom, value=-step_size)
}
__device__ __forceinline__ void adam_quad(float4& p, float4& m, float4& v, float4 g, const AdamCoef& c) {
    adam_elem(p.x, m.x, v.x, g.x, c);
    adam_elem(p.y, m.y, v.y, g.y, c);
    adam_elem(p.z, m.z, v.z, g.z, c);
    adam_elem(p.w, m.w, v.w, g.w, c);
}

#pragma clang fp contract(fast)

// ---- the zero-gradient step of the lazy replay ------------------------------------------------------------------------------------------
// Same recurrences for m and v (exact: one fma / one multiply); the parameter's increment through the hardware sqrt / reciprocal
// (<= 1 ulp each) instead of the IEEE divide sequences.  Its two per-step coefficients depend on the STEP NUMBER ALONE -- never on who
// replays the step, when, or from where: beta^s is the product pow_step(beta, s & ~63) * beta * ... * beta ((s & 63) factors, in this
// order), rounded bias corrections, then the hardware rcp / rsq -- so a row flushed at step t and caught up later lands on the same
// bits as one caught up in a single go (a checkpoint taken mid-run resumes bit-identically), and so do the gather's in-register replay
// and the optimizer's replay of the same row.  (Round 3 had a table of double-precision coefficients for the last 256 steps and the
// rcp / rsq form beyond it: the bits of a step depended on how old it was when it was replayed.)
struct IdleCoef { float nss, ibs; };          // -(lr rcp(1 - beta1^s)), rsq(1 - beta2^s)
struct IdleConst { float w1, beta2, eps; };
__device__ __forceinline__ IdleConst idle_const(const StepState& st) { return IdleConst{(float)(1.0 - st.beta1), (float)st.beta2, (float)st.eps}; }
constexpr int ANCHOR = 64;
struct StepPows {              // beta1^s, beta2^s of the running step s
    double b1, b2;
    __device__ __forceinline__ void at(const StepState& st, long long s) {
        const long long s0 = s & ~(long long)(ANCHOR - 1);
        b1 = pow_step(st.beta1, s0); b2 = pow_step(st.beta2, s0);
        for (long long k = s0; k < s; ++k) { b1 *= st.beta1; b2 *= st.beta2; }
    }
    // s -> s + 1
    __device__ __forceinline__ void next(const StepState& st, long long s_new) {
        if ((s_new & (ANCHOR - 1)) == 0) { b1 = pow_step(st.beta1, s_new); b2 = pow_step(st.beta2, s_new); }
        else { b1 *= st.beta1; b2 *= st.beta2; }
    }
    __device__ __forceinline__ IdleCoef coef(const StepState& st) const {
        IdleCoef c;
        c.nss = -(float)st.lr * __builtin_amdgcn_rcpf((float)(1.0 - b1));
        c.ibs = __builtin_amdgcn_rsqf((float)(1.0 - b2));
        return c;
    }
};

// coefficients of the last COEF_TAB steps (t - COEF_TAB + 1 .. t), computed once per block: a row idle for g steps replays g of them
constexpr int COEF_TAB = 256;
__device__ __forceinline__ void fill_coef_table(IdleCoef* tab, const StepState& st) {
    for (int i = threadIdx.x; i < COEF_TAB; i += blockDim.x) {
        const long long s = st.step - (COEF_TAB - 1) + i;
        if (s >= 1) { StepPows pw; pw.at(st, s); tab[i] = pw.coef(st); }
    }
    __syncthreads();
}

// K consecutive zero-gradient steps with coefficients c[0..K-1] on N elements; returns whether the LAST of them moved a parameter.  The
// moments first (two short chains), then the K denominators of an element -- independent square roots / reciprocals the hardware
// overlaps --, then the parameter's K additions in step order.
template <int N, int K>
__device__ __forceinline__ bool idle_steps(float (&p)[N], float (&m)[N], float (&v)[N], const IdleConst& k0, const IdleCoef (&c)[K]) {
    bool moved = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float mk[K], rk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            m[i] = __fmaf_rn(k0.w1, -m[i], m[i]);
            v[i] = __fmul_rn(v[i], k0.beta2);
            mk[k] = m[i];
            rk[k] = __builtin_amdgcn_rcpf(__fmaf_rn(__builtin_amdgcn_sqrtf(v[i]), c[k].ibs, k0.eps));
        }
        float before = p[i];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            before = p[i];
            p[i] = __fmaf_rn(__fmul_rn(c[k].nss, mk[k]), rk[k], p[i]);
        }
        moved |= p[i] != before;
    }
    return moved;
}

// Replay the zero-gradient steps s = from .. to (inclusive) on N elements of a row.  Steps older than the table take their coefficients
// from running powers (StepPows: re-anchored every 64 steps); the table holds the same values for the newest COEF_TAB steps.
// The parameter's part of a step stops mattering long before a long gap ends: the increment shrinks by ~beta1 per step (|m| does, the
// denominator and the bias corrections barely move), so once `p + increment` returns p it does so for every later step too: a lane
// whose parameters did not change in a chunk's last step skips the parameter arithmetic for the rest of the gap and only carries m and
// v on -- the SAME bits as the full replay at a fraction of its cost ("Lazy Adam over long gaps").
template <int N>
__device__ __forceinline__ void replay_elems(float (&p)[N], float (&m)[N], float (&v)[N], long long from, long long to, const StepState& st,
                                             const IdleCoef* tab) {
    constexpr int K = 4;                               // steps per chunk (the "did it move" test looks at a chunk's last step)
    const IdleConst k0 = idle_const(st);
    long long s = from;
    const long long tab_first = st.step - (COEF_TAB - 1);
    bool live = true;                                  // this lane's parameters still move
    if (s < tab_first) {
        StepPows pw;
        pw.at(st, s);
        const long long stop = (to + 1 < tab_first) ? to + 1 : tab_first;
        for (; s + K <= stop && live; s += K) {
            IdleCoef c[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { c[k] = pw.coef(st); pw.next(st, s + k + 1); }
            live = idle_steps<N, K>(p, m, v, k0, c);
        }
        for (; s < stop && live; ++s) {
            const IdleCoef c[1] = {pw.coef(st)};
            live = idle_steps<N, 1>(p, m, v, k0, c);
            pw.next(st, s + 1);
        }
    }
    if (s >= tab_first) {
        for (; s + K - 1 <= to && live; s += K) {
            IdleCoef c[K];
#pragma unroll
            for (int k = 0; k < K; ++k) c[k] = tab[s + k - tab_first];
            live = idle_steps<N, K>(p, m, v, k0, c);
        }
        for (; s <= to && live; ++s) {
            const IdleCoef c[1] = {tab[s - tab_first]};
            live = idle_steps<N, 1>(p, m, v, k0, c);
        }
    }
    for (; s <= to; ++s) {
#pragma unroll
        for (int i = 0; i < N; ++i) { m[i] = __fmaf_rn(k0.w1, -m[i], m[i]); v[i] = __fmul_rn(v[i], k0.beta2); }
    }
}
__device__ __forceinline__ void replay_quad(float4& p, float4& m, float4& v, long long from, long long to, const StepState& st,
                                            const IdleCoef* tab) {
    float pp[4] = {p.x, p.y, p.z, p.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
    replay_elems<4>(pp, mm, vv, from, to, st, tab);
    p = make_float4(pp[0], pp[1], pp[2], pp[3]); m = make_float4(mm[0], mm[1], mm[2], mm[3]); v = make_float4(vv[0], vv[1], vv[2], vv[3]);
}

}  // namespace amid
