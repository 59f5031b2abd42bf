// The evaluation loop's head (test(), /root/reference/train_sr.py:31-128) for the plain SASRec model, ONE launch, a workgroup per sample:
//   u      = mean_t LN_last(x[own, b, t, :])                                   (model_seq.py:385, :432-434; own = domain_id[b])
//   p[n]   = sigmoid(W2 relu(W1 [u ; E[id_n]] + b1) + b2), n < NI = 1 + neg_nums  (predictModule.forward, model_seq.py:40-54)
//   loss_b = sum_n BCE(p[n], y[n]) / (B NI)                                    (train_sr.py:63-64: the other domain's terms are masked)
//   rank_b = #{ n >= 1 : p[n] > p[0] - fix_value }                             (choose_predict + get_sample_scores, utils.py:21-40, :296-297;
//                                                                               train_sr.py:114-115 subtracts fix_value from the positive first)
// test() reads of every sample only the logits of its OWN domain's head (utils.py:21-40) and the loss masks the other domain's, so only that
// sequence is encoded (amid_sas_seq_fwd_split_infer_f32 over the live list) and only those NI logits are formed.  The candidates' rows are
// gathered from the table HERE -- the [B, NI, D] copy K1 used to write and the head to re-read (131 MB each way at 999 negatives) does not
// exist -- and logits, loss terms and the rank never leave the workgroup.
//
// Bit-compatible with the launches it replaces (amid_head_fwd_f32 + amid_positive_rank_f32 on a forward over both domains): every output is
// computed by the same operations in the same order -- lnmean_rows' eight row groups, user_half's / item_half's eight partial chains per hidden
// unit (e = part, part + 8, ... for the user half; the quads 4 part + 32 k for the item half) joined by group_sum<8>, the logit's balanced
// tree over the hidden units (group_sum<32>) -- so p, and with it every rank, is the same bits; tests/test_gpu_eval.py holds it to that.
//
// Shape of the fast path (D 128, hid 32, 512 threads): thread (jb = (tid & 63) >> 3, part = tid & 7) of every wave keeps the 4 x 16 weights
// W1[4 jb + jj][D + 4 part + 32 k + c] in registers for the whole launch; a wave takes eight candidates at a time: their rows go global ->
// registers -> a wave-private LDS slot (4 KB, double-buffered: the next eight are in flight while these are scored), every lane then reads its
// 4 x 16 bytes of a row (the eight lanes of a part group of ONE hidden block read 128 contiguous bytes, the other blocks the same: broadcast,
// conflict-free) and runs 64 independent-by-four fma chains.  The matrix cores are not used: their accumulation order is not the reference
// path's, and the launch is bound by the gather (NI rows of 512 B per sample) and the chains (NI hid D fma per sample) about equally.
#include "head_parts.h"

namespace amid {

#pragma clang fp contract(off)

struct EvalHeadArgs {
    const float* x;                        // [2, B, T, D] the last encoder layer's output (only the own sequences' rows are read)
    const float* lnw[2]; const float* lnb[2];
    const float* table; const int* ids;    // item table [n_rows, D]; candidate ids [B, NI] (column 0 = the positive), range-checked by the packing launch
    const float* w1; const float* b1; const float* w2; const float* b2;
    const float* labels;                   // [B, NI] or null (no loss)
    const long long* domain;               // [B]
    float* u;                              // [B, D] the own-domain user vectors (optional)
    float* p;                              // [B, NI] the own-domain scores (optional)
    int* rank; int* rank_raw;              // [B] with fix_value / with 0 (optional)
    float* loss_part;                      // [B] (with labels)
    int B, T, NI, D, hid;
    float eps, fix_value;
};

constexpr int EVAL_THREADS = 512;
typedef float ev_v2 __attribute__((ext_vector_type(2)));

// u_s[D] (LDS) = mean over T of LN_last of the own sequence's rows, by threads 0..255 in lnmean_rows' order (8 row groups of 32 lanes, rows
// rg + 8 i, chunks of 64 rows); red [8][D] LDS
__device__ __forceinline__ void eval_lnmean(const EvalHeadArgs& a, int b, int own, float* __restrict__ red, float* __restrict__ u_s) {
    const int D = a.D, T = a.T, q = D >> 2;
    const int sub = threadIdx.x & 31, rg = (threadIdx.x >> 5) & 7;
    const bool use_ln = a.lnw[0] != nullptr;
    if (threadIdx.x < 256) {
        const float* xb = a.x + ((long long)own * a.B + b) * T * D;
        const int c = sub;
        const bool on = c < q;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (use_ln && on) { ww = ld4(a.lnw[own] + 4 * c); b4 = ld4(a.lnb[own] + 4 * c); }
        for (int t0 = 0; t0 < T; t0 += 8 * HEAD_CHUNK) {
            float4 v[HEAD_CHUNK];
#pragma unroll
            for (int i = 0; i < HEAD_CHUNK; ++i) {
                const int t = t0 + rg + 8 * i;
                v[i] = (t < T && on) ? ld4(xb + (long long)t * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < HEAD_CHUNK; ++i) {
                const int t = t0 + rg + 8 * i;
                if (t < T) {                              // uniform over the 32 lanes of the row group
                    float4 y = v[i];
                    if (use_ln) {
                        const float mean = group_sum<32>(h4hsum(y)) / D;
                        float4 d4 = make_float4(y.x - mean, y.y - mean, y.z - mean, y.w - mean);
                        if (!on) d4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float rstd = 1.0f / sqrtf(group_sum<32>(h4hsum(h4mul(d4, d4))) / D + a.eps);
                        y = make_float4(d4.x * rstd * ww.x + b4.x, d4.y * rstd * ww.y + b4.y, d4.z * rstd * ww.z + b4.z, d4.w * rstd * ww.w + b4.w);
                    }
                    acc = h4add(acc, y);
                }
            }
        }
        if (on) st4(red + rg * D + 4 * c, acc);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < D; e += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k * D + e];
        s /= T;
        u_s[e] = s;
        if (a.u != nullptr) a.u[(long long)b * D + e] = s;
    }
    __syncthreads();
}

// au[j] = b1[j] + sum_e W1[j][e] u[e], j < hid: user_half's chains (eight lanes per hidden unit, lane `part` walks e = part, part + 8, ...)
__device__ __forceinline__ void eval_user_half(const EvalHeadArgs& a, const float* __restrict__ u_s, float* __restrict__ au) {
    const int D = a.D, hid = a.hid;
    const int part = threadIdx.x & 7;
    for (int o0 = 0; o0 < hid; o0 += EVAL_THREADS >> 3) {          // (uniform trip count: the shuffles need every lane)
        const int j = o0 + (threadIdx.x >> 3);
        const bool on = j < hid;
        float acc = 0.f;
        if (on) {
            const float* wr = a.w1 + (long long)j * 2 * D;
#pragma unroll 8
            for (int e = part; e < D; e += 8) acc = fmaf(wr[e], u_s[e], acc);
        }
        acc = group_sum<8>(acc);
        if (on && part == 0) au[j] = acc + a.b1[j];
    }
}

// what a thread does once all NI scores of the sample sit in p_s (LDS): loss terms, ranks, the optional copy of the scores
__device__ __forceinline__ void eval_finish(const EvalHeadArgs& a, int b, const float* __restrict__ p_s, float* __restrict__ scr) {
    const int NI = a.NI;
    const float pos = p_s[0] - a.fix_value, pos_raw = p_s[0];
    int c = 0, c_raw = 0;
    float lsum = 0.f;
    const float inv = 1.0f / ((float)a.B * (float)NI);
    for (int n = threadIdx.x; n < NI; n += EVAL_THREADS) {
        const float p = p_s[n];
        if (n >= 1) { c += p > pos ? 1 : 0; c_raw += p > pos_raw ? 1 : 0; }
        if (a.p != nullptr) a.p[(long long)b * NI + n] = p;
        if (a.labels != nullptr) {
            const float y = a.labels[(long long)b * NI + n];
            const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.0f - p), -100.f);      // torch's clamp of binary_cross_entropy
            lsum += -(y * lp + (1.f - y) * l1p) * inv;
        }
    }
    // integer counts: any order; the loss: lanes in a balanced tree, waves in order
    for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); c_raw += __shfl_xor(c_raw, o, 64); }
    lsum = group_sum<64>(lsum);
    int* ci = (int*)(scr + 8);
    if (lane_id() == 0) { scr[wave_id()] = lsum; ci[wave_id()] = c; ci[8 + wave_id()] = c_raw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int r = 0, rr = 0;
        float l = 0.f;
        for (int w = 0; w < EVAL_THREADS / 64; ++w) { r += ci[w]; rr += ci[8 + w]; l += scr[w]; }
        if (a.rank != nullptr) a.rank[b] = r;
        if (a.rank_raw != nullptr) a.rank_raw[b] = rr;
        if (a.labels != nullptr && a.loss_part != nullptr) a.loss_part[b] = l;
    }
}

// ---- D 128, hid 32: weights in registers, candidates through wave-private LDS slots -----------------------------------------------------------
constexpr int EV_D = 128, EV_HID = 32, EV_BATCH = 8;           // candidates per wave and round
// LDS (floats): red [8][D] | u_s [D] | au [hid] | scr [32] | p_s [NI_pad] | slots [8 waves][2][EV_BATCH][D]
__host__ __device__ inline size_t eval_fast_lds_floats(int NI) {
    return (size_t)8 * EV_D + EV_D + EV_HID + 32 + ((NI + 3) & ~3) + (size_t)(EVAL_THREADS / 64) * 2 * EV_BATCH * EV_D;
}

__global__ __launch_bounds__(EVAL_THREADS) void eval_head_fast_kernel(const EvalHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int D = EV_D, hid = EV_HID;
    const int b = blockIdx.x, NI = a.NI;
    const int own = a.domain[b] != 0 ? 1 : 0;
    float* red = sm;
    float* u_s = red + 8 * D;
    float* au = u_s + D;
    float* scr = au + hid;
    float* p_s = scr + 32;
    float* slots = p_s + ((NI + 3) & ~3);
    const int w = wave_id(), lane = lane_id();
    const int part = lane & 7, jb = lane >> 3;
    // this lane's 4 x 16 weights of the item half: W1[4 jb + jj][D + 4 part + 32 k + c] (issued first: they fly during the LayerNorm)
    // (held as PAIRS of hidden units -- wp[pr][k][c] = (W1[4 jb + 2 pr][..], W1[4 jb + 2 pr + 1][..]) -- so that two of the four chains advance in one
    // v_pk_fma_f32: the vector pipe's packed rate; every chain keeps its own order of additions)
    ev_v2 wp[2][4][4];
    {
        float4 wt[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int k = 0; k < 4; ++k) wt[jj][k] = ld4(a.w1 + (long long)(4 * jb + jj) * 2 * D + D + 4 * part + 32 * k);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wp[pr][k][0] = ev_v2{wt[2 * pr][k].x, wt[2 * pr + 1][k].x}; wp[pr][k][1] = ev_v2{wt[2 * pr][k].y, wt[2 * pr + 1][k].y};
                wp[pr][k][2] = ev_v2{wt[2 * pr][k].z, wt[2 * pr + 1][k].z}; wp[pr][k][3] = ev_v2{wt[2 * pr][k].w, wt[2 * pr + 1][k].w};
            }
    }
    // the candidates of this wave: n = w + 8 i (round r takes i = 8 r .. 8 r + 7); lane l of a round loads float4 (l & 31) of candidates 2 q + (l >> 5)
    const int* ids = a.ids + (long long)b * NI;
    const int per_wave = (NI - w + 7) / 8;                      // candidates of this wave
    const int rounds = (per_wave + EV_BATCH - 1) / EV_BATCH;
    float* slot0 = slots + (size_t)w * 2 * EV_BATCH * D;
    float4 rv[4];
    auto fetch = [&](int r) {                                   // round r's rows -> registers
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int i = r * EV_BATCH + 2 * qd + (lane >> 5);
            const int n = w + 8 * min(i, per_wave - 1);
            const long long id = ids[n];
            rv[qd] = ld4(a.table + id * D + 4 * (lane & 31));
        }
    };
    auto stage = [&](int r) {                                   // registers -> this round's slot
        float* sl = slot0 + (size_t)(r & 1) * EV_BATCH * D;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) st4(sl + (2 * qd + (lane >> 5)) * D + 4 * (lane & 31), rv[qd]);
    };
    if (rounds > 0) fetch(0);
    eval_lnmean(a, b, own, red, u_s);
    eval_user_half(a, u_s, au);
    __syncthreads();
    const float w2j = a.w2[4 * jb + (part & 3)], auj = au[4 * jb + (part & 3)], b2 = a.b2[0];
    const int pj = part & 3;
    for (int r = 0; r < rounds; ++r) {
        stage(r);
        if (r + 1 < rounds) fetch(r + 1);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot is wave-private: the wave's own stores have landed
        const float* sl = slot0 + (size_t)(r & 1) * EV_BATCH * D;
        float zs = 0.f;                                         // lane 48 + i: the logit of the round's i-th candidate
#pragma unroll
        for (int i = 0; i < EV_BATCH; ++i) {                    // (a round's tail past per_wave scores the last row again: never stored)
            const float* ir = sl + i * D;
            ev_v2 a01 = ev_v2{0.f, 0.f}, a23 = ev_v2{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 it = ld4(ir + 4 * part + 32 * k);
                a01 = __builtin_elementwise_fma(wp[0][k][0], ev_v2{it.x, it.x}, a01); a23 = __builtin_elementwise_fma(wp[1][k][0], ev_v2{it.x, it.x}, a23);
                a01 = __builtin_elementwise_fma(wp[0][k][1], ev_v2{it.y, it.y}, a01); a23 = __builtin_elementwise_fma(wp[1][k][1], ev_v2{it.y, it.y}, a23);
                a01 = __builtin_elementwise_fma(wp[0][k][2], ev_v2{it.z, it.z}, a01); a23 = __builtin_elementwise_fma(wp[1][k][2], ev_v2{it.z, it.z}, a23);
                a01 = __builtin_elementwise_fma(wp[0][k][3], ev_v2{it.w, it.w}, a01); a23 = __builtin_elementwise_fma(wp[1][k][3], ev_v2{it.w, it.w}, a23);
            }
            float acc[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = group_sum<8>(acc[jj]);
            // lanes part < 4 of hidden block jb take hidden unit j = 4 jb + part (lanes part >= 4 mirror them); the logit's tree over
            // j = 0 .. 31 in group_sum<32>'s order: pairs, quads (inside the lane quad), the two quads of an eight (the neighbouring hidden
            // block: the lane 8 further in the 16-lane row), then rows 0 + 1 and 2 + 3 (row_bcast15), then the halves (row_bcast31): the
            // total lands in row 3
            const float ci = pj == 0 ? acc[0] : pj == 1 ? acc[1] : pj == 2 ? acc[2] : acc[3];
            float zp = fmaf(w2j, fmaxf(auj + ci, 0.f), 0.f);
            zp = zp + dpp_move<0xB1>(zp);                       // j ^ 1
            zp = zp + dpp_move<0x4E>(zp);                       // j ^ 2
            zp = zp + dpp_move<0x128>(zp);                      // row_ror 8: the other hidden block of the row
            zp = zp + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, zp), 0x142, 0xA, 0xF, false));   // rows 1, 3 += rows 0, 2
            zp = zp + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, zp), 0x143, 0xC, 0xF, false));   // row 3 += row 1
            zs = lane == 48 + i ? zp : zs;
        }
        // the round's eight logits side by side: lane 48 + i finishes candidate i
        const int ii = r * EV_BATCH + (lane - 48);
        if (lane >= 48 && lane < 48 + EV_BATCH && ii < per_wave) {
            const float z = zs + b2;
            p_s[w + 8 * ii] = 1.0f / (1.0f + expf(-z));
        }
    }
    __syncthreads();
    eval_finish(a, b, p_s, scr);
}

// ---- any D <= 128 (multiple of 32), hid <= 64 (multiple of 4): the head launch's own phases (W1^T staged in LDS, chunks of 64 candidates),
// the candidates' rows read from the table ----------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t eval_gen_lds_floats(int D, int hid, int NI) {
    return head_carve_floats(D, hid, 64) + 32 * D + 32 + ((NI + 3) & ~3);
}

__global__ __launch_bounds__(EVAL_THREADS) void eval_head_gen_kernel(const EvalHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = a.D, hid = a.hid, NI = a.NI, b = blockIdx.x;
    const int own = a.domain[b] != 0 ? 1 : 0;
    const HeadLds s(sm, D, hid);
    float* scr2 = s.scr + 32 * D;
    float* p_s = scr2 + 32;
    const Tg tg = whole_block();
    stage_w1t(s.w1t, a.w1, 2 * D, hid, tg);
    eval_lnmean(a, b, own, s.scr, s.u_s + own * D);
    // au[own][j]: user_half's chains on the staged W1^T (the launch this replaces runs the same code for both domains)
    {
        const int part = tg.tid & 7;
        for (int o0 = 0; o0 < hid; o0 += tg.n >> 3) {
            const int j = o0 + (tg.tid >> 3);
            const bool on = j < hid;
            float acc = 0.f;
            if (on) {
                const float* ur = s.u_s + own * D;
#pragma unroll 8
                for (int e = part; e < D; e += 8) acc = fmaf(s.w1t[e * (hid + 1) + j], ur[e], acc);
            }
            acc = group_sum<8>(acc);
            if (on && part == 0) s.au[own * hid + j] = acc + a.b1[j];
        }
    }
    const int* ids = a.ids + (long long)b * NI;
    for (int n0 = 0; n0 < NI; n0 += 64) {
        const int nn = min(64, NI - n0);
        __syncthreads();
        {   // item_half's chains, rows from the table
            const int part = tg.tid & 7;
            for (int o0 = 0; o0 < nn * hid; o0 += tg.n >> 3) {
                const int nj = o0 + (tg.tid >> 3);
                const bool on = nj < nn * hid;
                const int n = on ? nj / hid : 0, j = on ? nj - n * hid : 0;
                float acc = 0.f;
                if (on) {
                    const float* ir = a.table + (long long)ids[n0 + n] * D;
#pragma unroll 4
                    for (int e = 4 * part; e < D; e += 32) {
                        const float4 it = ld4(ir + e);
                        const float* wp = s.w1t + (D + e) * (hid + 1) + j;
                        acc = fmaf(wp[0], it.x, acc); acc = fmaf(wp[hid + 1], it.y, acc);
                        acc = fmaf(wp[2 * (hid + 1)], it.z, acc); acc = fmaf(wp[3 * (hid + 1)], it.w, acc);
                    }
                }
                acc = group_sum<8>(acc);
                if (on && part == 0) s.ci[n * (hid + 1) + j] = acc;
            }
        }
        __syncthreads();
        for (int nd0 = 0; nd0 < nn; nd0 += tg.n >> 5) {
            const int n = nd0 + (tg.tid >> 5), j0 = tg.tid & 31;
            const bool on = n < nn;
            float zp = 0.f;
            if (on) for (int j = j0; j < hid; j += 32) zp = fmaf(a.w2[j], fmaxf(s.au[own * hid + j] + s.ci[n * (hid + 1) + j], 0.f), zp);
            const float z = group_sum<32>(zp) + a.b2[0];
            if (on && j0 == 0) p_s[n0 + n] = 1.0f / (1.0f + expf(-z));
        }
    }
    __syncthreads();
    eval_finish(a, b, p_s, scr2);
}

#pragma clang fp contract(fast)

}  // namespace amid

using namespace amid;

// x [2, B, T, D]: the last encoder layer's output (rows of the own sequences); ln_w / ln_b: host arrays of 2 device pointers (both null
// arrays: no LayerNorm); ids [B, NI] int32 (column 0 = the positive; validated by amid_pack_indices*); labels optional ([B, NI], with
// loss_part [B]: the sample's share of the batch-mean BCE); u [B, D], p [B, NI], rank, rank_raw [B] optional outputs.
extern "C" int amid_eval_head_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* table, const int* ids,
                                  const float* w1, const float* b1, const float* w2, const float* b2, const float* labels,
                                  const long long* domain_id, int B, int T, int NI, int D, int hid, float eps, float fix_value, float* u, float* p,
                                  int* rank, int* rank_raw, float* loss_part, void* stream) {
    AMID_CHECK_ARG(x && table && ids && w1 && b1 && w2 && b2 && domain_id && B > 0 && T > 0 && NI > 0 && D > 0 && (D % 32) == 0 && D <= 128 &&
                   hid > 0 && hid <= 64 && (hid % 4) == 0);
    AMID_CHECK_ARG((ln_w == nullptr) == (ln_b == nullptr) && (labels == nullptr || loss_part != nullptr) && (rank || rank_raw || p || loss_part));
    EvalHeadArgs a = {};
    a.x = x; a.table = table; a.ids = ids; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.labels = labels; a.domain = domain_id;
    for (int g = 0; g < 2; ++g) {
        AMID_CHECK_ARG(ln_w == nullptr || (ln_w[g] && ln_b[g]));
        a.lnw[g] = ln_w ? ln_w[g] : nullptr; a.lnb[g] = ln_b ? ln_b[g] : nullptr;
    }
    a.u = u; a.p = p; a.rank = rank; a.rank_raw = rank_raw; a.loss_part = loss_part;
    a.B = B; a.T = T; a.NI = NI; a.D = D; a.hid = hid; a.eps = eps; a.fix_value = fix_value;
    const bool fast = D == EV_D && hid == EV_HID;
    const size_t lds = (fast ? eval_fast_lds_floats(NI) : eval_gen_lds_floats(D, hid, NI)) * sizeof(float);
    if (lds > 160 * 1024) return AMID_ERR_UNSUPPORTED;
    static unsigned long long done_fast = 0, done_gen = 0;
    if (fast) {
        if (int rc = lds_attr_once((const void*)eval_head_fast_kernel, 160 * 1024, done_fast)) return rc;
        eval_head_fast_kernel<<<B, EVAL_THREADS, lds, (hipStream_t)stream>>>(a);
    } else {
        if (int rc = lds_attr_once((const void*)eval_head_gen_kernel, 160 * 1024, done_gen)) return rc;
        eval_head_gen_kernel<<<B, EVAL_THREADS, lds, (hipStream_t)stream>>>(a);
    }
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
