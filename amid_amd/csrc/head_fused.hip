// Fused head of the SASRec step, one workgroup per batch row b (both domains):
//   head_fwd : u_g[b] = mean_t LN_last(x[g,b,t,:])  (model_seq.py:385, :432-434)  ->  scorer (predictModule.forward,
//              model_seq.py:40-54)  ->  masked BCE partial + dLoss/dp (train_sr.py:203-212)
//   head_bwd : scorer backward  ->  d u_g[b]  ->  LN_last' + mean' for the T rows of (g, b)
// One launch each instead of lnmean + scorer (+ sum) launches; every short kernel of the step costs ~5 us of
// timeline whatever it does.  The first scorer weight W1 [hid, 2D] is staged TRANSPOSED in LDS (row stride hid+1:
// conflict-free both for "lane = hidden unit" and for "lane = feature" accesses), so no dot product waits on a
// global load.  Extra workgroups (blockIdx >= B) of head_bwd refresh the transposed copies of the encoder's
// projection weights for the backward GEMMs.
#include "head_parts.h"

namespace amid {

__global__ __launch_bounds__(512) void head_fwd_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    head_fwd_body(a, sm, blockIdx.x);
}

__global__ __launch_bounds__(512) void head_bwd_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    head_bwd_body<false>(a, sm);
}

// the training step's head in ONE launch: forward (loss partials, dLoss/dp) then, in the same workgroup with its LDS state
// intact, backward; the extra workgroups (blockIdx >= B) only transpose the projection weights
__global__ __launch_bounds__(512) void head_fwd_bwd_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    if ((int)blockIdx.x < a.B && a.own_only && a.T <= 16 * (HEAD_CHUNK / 2)) {
        // the live-sequence train step: the sample's own sequence, loaded once and kept in registers from the forward LayerNorm to its backward
        const int b = blockIdx.x;
        HEAD_STAMP(0);
        OwnRows R;
        own_rows_load(a, b, R);
        head_own_rows_body(a, sm, b, a.domain[b] != 0 ? 1 : 0, R);
        return;
    }
    if ((int)blockIdx.x < a.B) {
        head_fwd_body(a, sm, blockIdx.x);
        __threadfence_block();                            // dLoss/dp written above is read by other threads of this workgroup below
        __syncthreads();
    }
    head_bwd_body<true>(a, sm);
}

// scorer forward alone on given user vectors u [2, B, D] (evaluation of the isItC / isDR models, up to 1 000 candidates per row):
// W1^T staged in LDS, items in chunks of 64 -- the same code path as the fused head's forward half
__global__ __launch_bounds__(512) void scorer_fwd_only_kernel(const HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const HeadLds s(sm, a.D, a.hid);
    const int b = blockIdx.x, D = a.D;
    stage_w1t(s.w1t, a.w1, 2 * D, a.hid, whole_block());
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) s.u_s[e] = a.u[((long long)(e / D) * a.B + b) * D + (e % D)];
    __syncthreads();
    scorer_fwd_part(a, s, b, whole_block());
}

// ---------------------------------------------------------------------------------------------
// The head of the isItC / isDR train step after the user vectors are mixed (csrc/intercomp.hip): up to three scorers
// (predictModule, predict_ips, predict_gfunc; model_seq.py:436-440) forward on the same (u, items) of row b, the row's loss
// terms -- masked BCE (one head) or the doubly-robust objective of the current mode (three heads; train_sr_dr.py:216-221,
// :392-394, same arithmetic as dr_loss_kernel in head.hip) --, then every scorer's backward: d items and d u summed over the
// heads, weight-gradient partials per head.  One launch instead of 3 + 1 + 3 (each 20-40 us of mostly latency).
// Every head keeps its own LDS carve (W1^T, hidden pre-activations of the user vectors and of the items) from forward to backward,
// and the three heads run SIDE BY SIDE on three 256-thread groups of the workgroup (run one after the other they cost 89 us).
// ---------------------------------------------------------------------------------------------
struct MultiHeadArgs {
    HeadArgs h[3];             // per head: w1..b2, p1/p2 (outputs), dp1/dp2 (their gradients), sc_part; shared x-less fields
    int n_heads;               // 1: masked BCE on head 0 (h[0].labels set) ; 3: doubly-robust objective
    const float* labels; const long long* domain; const long long* ob;
    int mode; float w;
    float* dr_loss_part;       // [B][3]
    float* du;                 // [2, B, D]
};

__global__ __launch_bounds__(768) void scorer_multi_fwd_bwd_kernel(const MultiHeadArgs m) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const HeadArgs& a0 = m.h[0];
    if ((int)blockIdx.x >= a0.B) { transpose_extra(a0, sm); return; }
    const int b = blockIdx.x, D = a0.D, NI = a0.NI, nh = m.n_heads;
    // three heads: the workgroup's 768 threads are three groups of 256, one head each, side by side (same control flow, so the
    // barriers inside the scorer parts line up); one head: the whole workgroup
    const int gsz = nh == 1 ? (int)blockDim.x : 256;
    const int h = __builtin_amdgcn_readfirstlane((int)threadIdx.x / gsz);
    const Tg tg{(int)threadIdx.x - h * gsz, gsz};
    const int CH = nh == 1 ? 64 : 16;
    const size_t carve = head_carve_floats(D, a0.hid, CH);
    HeadLds s(sm + h * carve, D, a0.hid, CH);
    float* dit_all = sm + nh * carve;                         // three heads: [3][NI <= 16][D] shares of d items
    float* scratch = dit_all + (nh == 1 ? 0 : nh * CH * D);   // [8 D]
    s.scr = scratch;
    const HeadArgs& a = m.h[h];
    stage_w1t(s.w1t, a.w1, 2 * D, a0.hid, tg);
    for (int e = tg.tid; e < 2 * D; e += tg.n) s.u_s[e] = a0.u[((long long)(e / D) * a0.B + b) * D + (e % D)];
    __syncthreads();
    scorer_fwd_part(a, s, b, tg);
    if (nh == 3) {
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x < 64) {                               // wave 0: the row's doubly-robust terms (dr_loss_kernel, head.hip)
            const int lane = threadIdx.x;
            const int d = m.domain[b] != 0;
            const float ob = (m.ob != nullptr) ? (float)m.ob[b] : 0.f;
            const float inv = 1.0f / (float)((long long)a0.B * NI);
            float lc = 0.f, le = 0.f, lr = 0.f;
            for (int n = lane; n < NI; n += 64) {
                const long long o = (long long)b * NI + n;
                const float p = (d ? m.h[0].p2 : m.h[0].p1)[o], ips = (d ? m.h[1].p2 : m.h[1].p1)[o], g = (d ? m.h[2].p2 : m.h[2].p1)[o];
                const float y = m.labels[o];
                const float bce = -(y * fmaxf(logf(p), -100.f) + (1.f - y) * fmaxf(logf(1.f - p), -100.f));
                const float dbce = (p - y) / fmaxf(p * (1.f - p), 1e-12f);
                float dp, dg, dips;
                lc += bce;
                le += (bce - g) * (bce - g) / ips;
                const float q = bce * bce - g * g;
                lr += g * g + ob * q * q / ips;
                if (m.mode == 0) {
                    dp = (1.f + 2.f * m.w * (bce - g) / ips) * dbce;
                    dg = -2.f * m.w * (bce - g) / ips;
                    dips = -m.w * (bce - g) * (bce - g) / (ips * ips);
                } else {
                    dp = ob * 4.f * q * bce / ips * dbce;
                    dg = 2.f * g - ob * 4.f * q * g / ips;
                    dips = -ob * q * q / (ips * ips);
                }
                (d ? m.h[0].dp2 : m.h[0].dp1)[o] = dp * inv; (d ? m.h[0].dp1 : m.h[0].dp2)[o] = 0.f;
                (d ? m.h[1].dp2 : m.h[1].dp1)[o] = dips * inv; (d ? m.h[1].dp1 : m.h[1].dp2)[o] = 0.f;
                (d ? m.h[2].dp2 : m.h[2].dp1)[o] = dg * inv; (d ? m.h[2].dp1 : m.h[2].dp2)[o] = 0.f;
            }
            lc = group_sum<64>(lc); le = group_sum<64>(le); lr = group_sum<64>(lr);
            if (lane == 0) { m.dr_loss_part[b * 3 + 0] = lc * inv; m.dr_loss_part[b * 3 + 1] = le * inv; m.dr_loss_part[b * 3 + 2] = lr * inv; }
        }
    }
    __threadfence_block();                                    // dLoss/dp written above is read by other threads of this workgroup below
    __syncthreads();
    scorer_bwd_part<true, false>(a, s, b, tg, nh == 1 ? nullptr : dit_all + h * CH * D);      // ends with a barrier
    // d u (left by every head at the start of its ci region) and, with three heads, d items: fixed-order sums over the heads
    const size_t ci_off = (size_t)(s.ci - (sm + h * carve));
    for (int e = threadIdx.x; e < 2 * D; e += blockDim.x) {
        float t = sm[ci_off + e];
        for (int k = 1; k < nh; ++k) t += sm[k * carve + ci_off + e];
        m.du[((long long)(e / D) * a0.B + b) * D + (e % D)] = t;
    }
    if (nh > 1)
        for (int i = threadIdx.x; i < NI * D; i += blockDim.x) {
            float t = dit_all[i];
            for (int k = 1; k < nh; ++k) t += dit_all[k * CH * D + i];
            a0.ditems[(long long)b * NI * D + i] = t;
        }
}

}  // namespace amid

using namespace amid;

#ifdef AMID_HEAD_STAMPS
extern "C" int amid_head_stamps_read(unsigned long long* host) {       // diagnostic library only
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(amid::amid_head_stamp_buf), sizeof(unsigned long long) * 32);
}
#endif

static int head_fill(HeadArgs& a, const float* x, const float* const* lnw, const float* const* lnb, const float* items, const float* w1,
                     const float* b1, const float* w2, const float* b2, int B, int T, int NI, int D, int hid, float eps) {
    if (!(x && items && w1 && b1 && w2 && b2) || B <= 0 || T <= 0 || NI <= 0 || D <= 0 || (D % 32) != 0 || D > 128 || hid <= 0 || hid > 64 || (hid % 4) != 0)
        return AMID_ERR_ARG;
    a.x = x; a.items = items; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2;
    for (int g = 0; g < 2; ++g) { a.lnw[g] = lnw ? lnw[g] : nullptr; a.lnb[g] = lnb ? lnb[g] : nullptr; }
    a.B = B; a.T = T; a.NI = NI; a.D = D; a.hid = hid; a.eps = eps; a.n_tr = 0;
    return AMID_OK;
}

static int head_lds_attr(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
    }
    return AMID_OK;
}

// ln_w / ln_b: host arrays of 2 device pointers (both null arrays: no LayerNorm, plain mean over T)
extern "C" int amid_head_fwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                 const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id, int B,
                                 int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                 float* loss_part, void* stream) {
    HeadArgs a = {};
    if (int e = head_fill(a, x, ln_w, ln_b, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    AMID_CHECK_ARG(u && p1 && p2 && (!labels || (domain_id && dp1 && dp2 && loss_part)));
    a.labels = labels; a.domain = domain_id; a.u = u; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)head_fwd_kernel, lds)) return e;
    head_fwd_kernel<<<B, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// tr_src / tr_dst: optional host arrays of n_tr (<= 32) device pointers of square D x D matrices transposed by extra workgroups
extern "C" int amid_head_bwd_f32(const float* x, const float* const* ln_w, const float* u, const float* items, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* p1, const float* p2, const float* dp1, const float* dp2, int B,
                                 int T, int NI, int D, int hid, float eps, float* dx, float* ditems, float* ln_part, float* sc_part,
                                 const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    HeadArgs a = {};
    if (int e = head_fill(a, x, ln_w, nullptr, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    AMID_CHECK_ARG(u && p1 && p2 && dp1 && dp2 && dx && ditems && sc_part && (!ln_w || ln_part) && n_tr >= 0 && n_tr <= 32);
    a.u = const_cast<float*>(u); a.p1 = const_cast<float*>(p1); a.p2 = const_cast<float*>(p2);
    a.dp1 = const_cast<float*>(dp1); a.dp2 = const_cast<float*>(dp2);
    a.dx = dx; a.ditems = ditems; a.ln_part = ln_part; a.sc_part = sc_part; a.n_tr = n_tr;
    for (int i = 0; i < n_tr; ++i) { AMID_CHECK_ARG(tr_src && tr_dst && tr_src[i] && tr_dst[i]); a.tr_src[i] = tr_src[i]; a.tr_dst[i] = tr_dst[i]; }
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)head_bwd_kernel, lds)) return e;
    const int extra = n_tr > 0 ? min(n_tr * (D / 32) * (D / 32), 96) : 0;
    head_bwd_kernel<<<B + extra, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// amid_head_fwd_f32 (with labels) immediately followed by amid_head_bwd_f32, ONE launch -- the training step's head
static int head_fwd_bwd(int own_only, const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                     int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                     float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part,
                                     const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream, float* hidg = nullptr) {
    HeadArgs a = {};
    if (int e = head_fill(a, x, ln_w, ln_b, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    a.own_only = own_only;
    a.hidg = hidg;
    AMID_CHECK_ARG(labels && domain_id && u && p1 && p2 && dp1 && dp2 && loss_part && dx && ditems && (sc_part || hidg) && (!ln_w || ln_part) &&
                   n_tr >= 0 && n_tr <= 32);
    AMID_CHECK_ARG(hidg == nullptr || NI <= 64);
    a.labels = labels; a.domain = domain_id; a.u = u; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    a.dx = dx; a.ditems = ditems; a.ln_part = ln_part; a.sc_part = sc_part; a.n_tr = n_tr;
    for (int i = 0; i < n_tr; ++i) { AMID_CHECK_ARG(tr_src && tr_dst && tr_src[i] && tr_dst[i]); a.tr_src[i] = tr_src[i]; a.tr_dst[i] = tr_dst[i]; }
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)head_fwd_bwd_kernel, lds)) return e;
    const int extra = n_tr > 0 ? min(n_tr * (D / 32) * (D / 32), 96) : 0;
    head_fwd_bwd_kernel<<<B + extra, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

namespace amid {
int head_own_vec_args(HeadArgs& a, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1, const float* b1,
                      const float* w2, const float* b2, const float* labels, const long long* domain_id, int B, int T, int NI, int D, int hid,
                      float eps, float* u, float* p1, float* p2, float* dp1, float* dp2, float* loss_part, float* dx, float* ditems,
                      float* ln_part, float* hidg) {
    a = HeadArgs{};
    if (int e = head_fill(a, items /* (x: the rows come from the forward's workgroup) */, ln_w, ln_b, items, w1, b1, w2, b2, B, T, NI, D, hid, eps)) return e;
    a.x = nullptr;
    AMID_CHECK_ARG(labels && domain_id && u && p1 && p2 && dp1 && dp2 && loss_part && dx && ditems && hidg && (!ln_w || ln_part) && NI <= 64);
    a.own_only = 1; a.hidg = hidg;
    a.labels = labels; a.domain = domain_id; a.u = u; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    a.dx = dx; a.ditems = ditems; a.ln_part = ln_part; a.sc_part = nullptr; a.n_tr = 0;
    return AMID_OK;
}
}  // namespace amid

extern "C" int amid_head_fwd_bwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                     int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                     float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part,
                                     const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    return head_fwd_bwd(0, x, ln_w, ln_b, items, w1, b1, w2, b2, labels, domain_id, B, T, NI, D, hid, eps, u, p1, p2, dp1, dp2, loss_part, dx,
                        ditems, ln_part, sc_part, tr_src, tr_dst, n_tr, stream);
}

// the same when only the sequence (domain_id[b], b) of every sample was encoded (the train step over the live sequences): the other
// domain's user vector reads as zero (its logits are NOT the model's: the masked loss never reads them, train_sr.py:205-211), it
// receives no gradient, and of x / dx only the own sequences' rows are read / written
extern "C" int amid_head_fwd_bwd_own_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                         const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                         int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1,
                                         float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part,
                                         const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    return head_fwd_bwd(1, x, ln_w, ln_b, items, w1, b1, w2, b2, labels, domain_id, B, T, NI, D, hid, eps, u, p1, p2, dp1, dp2, loss_part, dx,
                        ditems, ln_part, sc_part, tr_src, tr_dst, n_tr, stream);
}

// amid_head_fwd_bwd_own_f32 that hands the scorer's weight gradients on as per-sample HIDDEN gradients (hidg [B][amid_scorer_vec_floats(NI, hid)]:
// da [2][hid] | dc [NI][hid] | dW2's [hid] | db2's) instead of per-sample partials of every weight (sc_part: 32 KB a sample, 8.5 MB a step
// written here and read back by the gradient tail): amid_grad_tail_live_f32 forms dW1 = sum_b da_b (x) u_b + dc_b (x) items_b itself.  NI <= 64.
extern "C" long long amid_scorer_vec_floats(int NI, int hid) { return ((long long)(3 + NI) * hid + 1 + 3) & ~3LL; }
extern "C" int amid_head_fwd_bwd_own_vec_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                             const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                                             int B, int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1,
                                             float* dp2, float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg,
                                             const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    AMID_CHECK_ARG(hidg != nullptr);
    return head_fwd_bwd(1, x, ln_w, ln_b, items, w1, b1, w2, b2, labels, domain_id, B, T, NI, D, hid, eps, u, p1, p2, dp1, dp2, loss_part, dx,
                        ditems, ln_part, nullptr, tr_src, tr_dst, n_tr, stream, hidg);
}

// Up to three scorers forward + loss + backward in ONE launch on given user vectors u [2, B, D] (the isItC / isDR train step, see
// scorer_multi_fwd_bwd_kernel).  Host arrays of n_heads device pointers: w1, b1, w2, b2, p1, p2 (outputs), dp1, dp2 (gradients of
// the outputs, written here), sc_part (weight-gradient partials).  n_heads == 1: masked BCE of train_sr.py:203-212 (loss_part [B]);
// n_heads == 3 (predictModule, predict_ips, predict_gfunc): the doubly-robust objective `mode` of amid_dr_loss_f32
// (dr_loss_part [B][3]; ob_label needed for mode 1).  du [2, B, D] and ditems [B, NI, D] are sums over the heads.
extern "C" int amid_scorer_multi_fwd_bwd_f32(const float* u, const float* items, const float* const* w1, const float* const* b1,
                                             const float* const* w2, const float* const* b2, int n_heads, const float* labels,
                                             const long long* domain_id, const long long* ob_label, int mode, float dr_e_w, int B, int NI,
                                             int D, int hid, float* const* p1, float* const* p2, float* const* dp1, float* const* dp2,
                                             float* loss_part, float* dr_loss_part, float* du, float* ditems, float* const* sc_part,
                                             const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream) {
    AMID_CHECK_ARG(u && items && w1 && b1 && w2 && b2 && labels && domain_id && p1 && p2 && dp1 && dp2 && du && ditems && sc_part);
    AMID_CHECK_ARG((n_heads == 1 && loss_part) || (n_heads == 3 && dr_loss_part && (mode == 0 || (mode == 1 && ob_label))));
    AMID_CHECK_ARG(n_tr >= 0 && n_tr <= 32);
    MultiHeadArgs m = {};
    for (int h = 0; h < n_heads; ++h) {
        HeadArgs& a = m.h[h];
        AMID_CHECK_ARG(p1[h] && p2[h] && dp1[h] && dp2[h] && sc_part[h]);
        if (int e = head_fill(a, u, nullptr, nullptr, items, w1[h], b1[h], w2[h], b2[h], B, 1, NI, D, hid, 0.f)) return e;
        a.u = const_cast<float*>(u); a.p1 = p1[h]; a.p2 = p2[h]; a.dp1 = dp1[h]; a.dp2 = dp2[h]; a.sc_part = sc_part[h]; a.ditems = ditems;
        if (n_heads == 1) { a.labels = labels; a.domain = domain_id; a.loss_part = loss_part; }
    }
    m.h[0].n_tr = n_tr;
    for (int i = 0; i < n_tr; ++i) { AMID_CHECK_ARG(tr_src && tr_dst && tr_src[i] && tr_dst[i]); m.h[0].tr_src[i] = tr_src[i]; m.h[0].tr_dst[i] = tr_dst[i]; }
    m.n_heads = n_heads; m.labels = labels; m.domain = domain_id; m.ob = ob_label; m.mode = mode; m.w = dr_e_w;
    m.dr_loss_part = dr_loss_part; m.du = du;
    if (n_heads == 3 && NI > 16) return AMID_ERR_UNSUPPORTED;          // the three heads' d item shares are summed through LDS
    const size_t lds = ((size_t)n_heads * head_carve_floats(D, hid, n_heads == 1 ? 64 : 16) + (n_heads == 1 ? 0 : (size_t)n_heads * 16 * D) +
                        8 * (size_t)D) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)scorer_multi_fwd_bwd_kernel, lds)) return e;
    const int extra = n_tr > 0 ? min(n_tr * (D / 32) * (D / 32), 96) : 0;
    scorer_multi_fwd_bwd_kernel<<<B + extra, n_heads == 1 ? 512 : 768, lds, (hipStream_t)stream>>>(m);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// replaces: predictModule.forward model_seq.py:40-54 on given user vectors; with labels != NULL also the masked BCE mean of
// train_sr.py:203-212 (per-row loss partials + dLoss/dp)
extern "C" int amid_scorer_fwd_f32(const float* u, const float* items, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* labels, const long long* domain_id, int B, int NI, int D, int hid, float* p1, float* p2,
                                   float* dp1, float* dp2, float* loss_part, void* stream) {
    AMID_CHECK_ARG(u && p1 && p2 && (!labels || (domain_id && dp1 && dp2 && loss_part)));
    HeadArgs a = {};
    if (int e = head_fill(a, u, nullptr, nullptr, items, w1, b1, w2, b2, B, 1, NI, D, hid, 0.f)) return e;
    a.u = const_cast<float*>(u); a.labels = labels; a.domain = domain_id; a.p1 = p1; a.p2 = p2; a.dp1 = dp1; a.dp2 = dp2; a.loss_part = loss_part;
    const size_t lds = head_lds_floats(D, hid) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    if (int e = head_lds_attr((const void*)scorer_fwd_only_kernel, lds)) return e;
    scorer_fwd_only_kernel<<<B, 512, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
