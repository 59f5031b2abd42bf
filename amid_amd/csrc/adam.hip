// K4: optimizer.  The reference runs torch.optim.Adam(model.parameters(), lr) DENSE over every
// parameter including the whole item table (train_sr.py:480, :213-215): 7 passes over 458 MB per
// step.  Here the table is updated lazily but dense-EQUIVALENTLY: a row keeps (m, v, last_step);
// a zero-gradient Adam step still moves a row through its momentum, so before a row is read by
// the forward gather of step t its pending zero-gradient steps last+1 .. t-1 are replayed
// (amid_lazy_adam_catchup), and after backward the real step t is applied (amid_lazy_adam_apply).
// Rows never touched have m = v = 0 => every update is exactly 0 => nothing to replay.
// Arithmetic follows torch's single-tensor CPU Adam op by op (lerp as fma, addcmul / addcdiv
// unfused), with bias corrections evaluated in double like the Python side does.
#include "common.h"
#include "rng.h"
#include "sort_phases.h"
#include "adam_replay.h"
#include "live_list.h"
#include "seg_spans.h"
#include "tail_parts.h"
#include "reduce_partials.h"
#include "weights_image.h"

namespace amid {

// mode 0: catch-up (steps last+1 .. t-1, g = 0)   mode 1: apply (catch-up if needed, then step t with g)
template <int MODE>
__global__ __launch_bounds__(256) void lazy_adam_rows_kernel(float* __restrict__ table, float* __restrict__ m_tab, float* __restrict__ v_tab,
                                                             int* __restrict__ last, const int* __restrict__ uniq_ids,
                                                             const int* __restrict__ n_uniq_p, const float* __restrict__ uniq_grad, int D,
                                                             const StepState* __restrict__ stp, float grad_scale) {
    __shared__ IdleCoef tab[COEF_TAB];
    const StepState st = *stp;
    const long long t = st.step;
    const int U = *n_uniq_p;
    const int sub = threadIdx.x & 31;
    const int q = D >> 2;
    const int hw0 = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5), n_hw = gridDim.x * (blockDim.x >> 5);
    if (blockIdx.x * (blockDim.x >> 5) >= U) return;              // block-uniform: nothing to do for this block
    fill_coef_table(tab, st);
    const AdamCoef cnow = adam_coef_now(st);
    for (int u = hw0; u < U; u += n_hw) {
        const long long r = uniq_ids[u];
        const long long l = last[r];
        const bool lag = (l > 0 && l < t - 1);
        if (MODE == 0 && !lag) continue;
        for (int c = sub; c < q; c += 32) {
            const long long off = r * D + 4 * c;
            float4 p = ld4(table + off), m = ld4(m_tab + off), v = ld4(v_tab + off);
            if (lag) replay_quad(p, m, v, l + 1, t - 1, st, tab);
            if (MODE == 1) {
                const float4 g = f4scale(ld4(uniq_grad + (long long)u * D + 4 * c), grad_scale);
                adam_quad(p, m, v, g, cnow);
            }
            st4(table + off, p); st4(m_tab + off, m); st4(v_tab + off, v);
        }
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) last[r] = (MODE == 1) ? (int)t : (int)(t - 1);
    }
}

// Catch-up driven by the raw index list instead of the unique list: one half-wave per gathered POSITION checks its
// row and replays if it lags.  Positions sharing a row are serialised by an atomic claim on `last[row]` (the winner
// moves it to t - 1 and replays from the value it saw; the others find the row current), and the pad row (80-90 %
// of the positions) is touched every step, so it never lags.  This takes the sort/unique off the critical path:
// it is only needed after backward.
__global__ __launch_bounds__(256) void lazy_adam_catchup_pos_kernel(float* __restrict__ table, float* __restrict__ m_tab, float* __restrict__ v_tab,
                                                                    int* __restrict__ last, const int* __restrict__ idx, int n_idx, int D,
                                                                    const StepState* __restrict__ stp, const SortRider rd) {
    // rider: the first workgroups run a phase of the step's index sort (sort_phases.h) beside the catch-up
    const int nrb = rider_blocks(rd);
    __shared__ __attribute__((aligned(16))) int sort_hist[OS_BINS_MAX];      // (16 KB: the 4 096-bin count, or the chain's 1 024-bin scatter)
    static_assert(sizeof(SortScatterLds<1024>) <= sizeof(int) * OS_BINS_MAX, "the chained phases fit the count's histogram");
    if ((int)blockIdx.x < nrb) {
        if (rd.phase == SORT_CHAIN_PHASE) sort_chain_block(rd.plan, blockIdx.x, sort_hist);      // the whole sort (sort_phases.h)
        else sort_phase_ct<1>(rd.plan, blockIdx.x, sort_hist);
        return;
    }
    const int bid = blockIdx.x - nrb, nbk = gridDim.x - nrb;
    __shared__ IdleCoef tab[COEF_TAB];
    __shared__ int any_lag;
    const StepState st = *stp;
    const long long t = st.step;
    const int lane = threadIdx.x & 63;
    // replay row r from its stamp l (claimed by this wave): two elements per lane -- a replay is a chain of dependent steps bound by the
    // quarter-rate sqrt / rcp, the fewer elements a lane carries the shorter the chain
    auto replay_row = [&](long long r, int l) {
        for (int c = lane; c < (D >> 1); c += 64) {
            const long long off = r * D + 2 * c;
            // (non-temporal: a lagging row is read once per step and the tables are far larger than the caches -- the stress's launch 140.5 -> 136.7 us)
            typedef float nt_f2 __attribute__((ext_vector_type(2)));
            const nt_f2 pn = __builtin_nontemporal_load((const nt_f2*)(table + off)), mn = __builtin_nontemporal_load((const nt_f2*)(m_tab + off)),
                        vn = __builtin_nontemporal_load((const nt_f2*)(v_tab + off));
            const float2 p2 = make_float2(pn[0], pn[1]), m2 = make_float2(mn[0], mn[1]), v2 = make_float2(vn[0], vn[1]);
            float p[2] = {p2.x, p2.y}, m[2] = {m2.x, m2.y}, v[2] = {v2.x, v2.y};
            replay_elems<2>(p, m, v, (long long)l + 1, t - 1, st, tab);
            *reinterpret_cast<float2*>(table + off) = make_float2(p[0], p[1]);
            *reinterpret_cast<float2*>(m_tab + off) = make_float2(m[0], m[1]);
            *reinterpret_cast<float2*>(v_tab + off) = make_float2(v[0], v[1]);
        }
    };
    // a WAVE per lagging row (a replay is a chain of dependent steps bound by the quarter-rate sqrt / rcp: the fewer elements a lane carries,
    // the shorter the chain), but the wave's positions are RESOLVED side by side first -- lane j the wave's j-th: index -> stamp is a chain
    // of two dependent loads, paid once per 64 positions instead of once per position (round 5, as step_head_kernel: at the cfg 5 stress
    // every wave walks six to seven positions and the launch was a chain of round trips, 168 us for 650 MB)
    const int n_w = nbk * (blockDim.x >> 6), wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) any_lag = 0;
    __syncthreads();
    bool filled = false;                                 // (block-uniform)
    for (long long base = (long long)bid * (blockDim.x >> 6); base < n_idx; base += 64LL * n_w) {      // (block-uniform trip count)
        const long long i = base + wv + (long long)lane * n_w;
        int id = 0, l = 0;
        if (i < n_idx) { id = idx[i]; l = last[id]; }
        const bool lag = l > 0 && l < t - 1;
        const unsigned long long lagging = __ballot(lag);
        if (lagging != 0ull && lane == 0) any_lag = 1;
        __syncthreads();
        if (any_lag && !filled) { fill_coef_table(tab, st); filled = true; }      // (usually no row of a block lags: no table)
        unsigned long long todo = lagging;
        while (todo != 0ull) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const long long r = __builtin_amdgcn_readlane(id, j);
            const int lj = __builtin_amdgcn_readlane(l, j);
            int won = 0;
            if (lane == 0) won = (atomicCAS(&last[r], lj, (int)(t - 1)) == lj) ? 1 : 0;     // claim the row for this wave
            won = __builtin_amdgcn_readfirstlane(won);
            if (!won) continue;
            replay_row(r, lj);
        }
        if (base + 64LL * n_w < n_idx) __syncthreads();    // (a wave of the next round must not raise any_lag while a slow one still reads it)
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The HEAD of a train step in ONE launch (amid_step_head_f32; the live-sequence step on an input pool): what amid_pack_indices_pool_live,
// the catch-up launch and phase 1 of the step's index sort did in two launches.  Three roles by workgroup index:
//   [0, nrb)          phase 1 of the sort of the step's COMPACT index list (the ids of every sample's own-domain sequence + its items:
//                     half the positions of the full list -- the other domain's sequence of a sample carries no gradient and is not read
//                     by the live forward, train_sr.py:205-211), keys read straight from the batch image
//   [nrb, nrb + npk)  the packing: the image mirrored into the plan's static input words, the full index list (the gather K1 and the head
//                     read it), the compact list (ids + rows of the full layout: what the sort's later phases, the segment reduce and the
//                     row Adam run on), the live-sequence list (last packing workgroup)
//   the rest          the lazy-Adam catch-up of the compact list's rows, a wave per position (lazy_adam_catchup_pos_kernel's arithmetic)
// Nobody waits for anybody: every role reads the batch image, which is input.  The step counter: every workgroup takes t - 1 from
// StepState::step_done and one thread writes `step` = t, which no workgroup of this launch reads (rng.h) -- no ticket, no atomic.
struct PoolBatch {          // a packed batch image (engine_io.py pack_epoch): [i_node B][neg B n_neg][seq_d1 B T][seq_d2 B T][domain B] ...
    const long long* __restrict__ src;
    int B, T, n_neg;
    long long n_rows;
    __device__ __forceinline__ int n_compact() const { return B * T + B * (1 + n_neg); }
    // compact position p: sample b's own-domain sequence at p = b T + t, then the items.  Returns the id (0 when out of range, as the
    // packing launches do -- they raise the flag) and the row of the full [2 B T + B NI] layout its gradient stands in
    __device__ __forceinline__ int at(int p, int& row) const {
        const int M = B * T, NI = 1 + n_neg, items0 = B + B * n_neg;
        int word;
        if (p < M) {
            const int b = p / T;
            const int dom = src[items0 + 2 * M + b] != 0 ? 1 : 0;
            word = items0 + dom * M + p;
            row = dom * M + p;
        } else {
            const int j = p - M, b = j / NI, k = j - b * NI;
            word = k == 0 ? b : B + b * n_neg + (k - 1);
            row = 2 * M + j;
        }
        const long long v = src[word];
        return (v < 0 || v >= n_rows) ? 0 : (int)v;
    }
};
struct PoolKeys { PoolBatch pb; __device__ __forceinline__ int operator()(int k) const { int row; return pb.at(k, row); } };

struct StepHeadArgs {
    const long long* pool; long long stride; int n_pool; long long phase;
    long long* in_pack; int in_words;
    int B, T, n_neg; long long n_rows;
    int* idx_all; int* idx_c; int* row_c; int* live; int* err;
    float* table; float* m_tab; float* v_tab; int* last; int D;
    StepState* st;
    int npk;
};

// W16: the step's weight images (the forward's and the backward strips' operands: 144 planes at the headline shape) written by the launch's LAST
// workgroups -- the gather K1's riders, for a step whose gather is the forward's prologue (amid_step_head_w16_f32)
template <bool W16>
__global__ __launch_bounds__(256) void step_head_kernel(const StepHeadArgs a, const SortRider rd, const W16Rider wr) {
    if constexpr (W16) {
        const int first = (int)gridDim.x - wr.n * wr.per;
        if ((int)blockIdx.x >= first) { w16_rider_block(wr, blockIdx.x - first); return; }
    }
    const long long t_pre = a.st->step_done;
    long long which = (t_pre + a.phase) % a.n_pool;
    if (which < 0) which += a.n_pool;
    PoolBatch pb;
    pb.src = a.pool + which * a.stride; pb.B = a.B; pb.T = a.T; pb.n_neg = a.n_neg; pb.n_rows = a.n_rows;
    const int nrb = rider_blocks(rd);
    if ((int)blockIdx.x < nrb) {
        const SortPlan& sp = rd.plan;
        __shared__ __attribute__((aligned(16))) int sort_hist[OS_BINS_MAX];
        if (sp.g0.bits <= 10)      // (block-uniform: the digit width of the plan -- 1 024 bins below 2^20 rows, 4 096 up to 2^24)
            os_count_block<1024>(blockIdx.x, sp.nblk, PoolKeys{pb}, sp.g0, sp.state, sp.counts0, sp.stot0, sp.stot0_copy, sp.n_zero_a, sp.counts1,
                                 sp.n_counts1, sp.stot1, sp.n_stot1, (int*)sp.hstatus, sort_hist);
        else
            os_count_block<OS_BINS_MAX>(blockIdx.x, sp.nblk, PoolKeys{pb}, sp.g0, sp.state, sp.counts0, sp.stot0, sp.stot0_copy, sp.n_zero_a, sp.counts1,
                                        sp.n_counts1, sp.stot1, sp.n_stot1, (int*)sp.hstatus, sort_hist);
        return;
    }
    const int M = a.B * a.T, NI = 1 + a.n_neg, n_items = a.B * a.n_neg, n_index_words = a.B + n_items + 2 * M;
    const int n_c = pb.n_compact();
    if ((int)blockIdx.x < nrb + a.npk) {
        const int pk = blockIdx.x - nrb;
        if (pk == 0 && threadIdx.x == 0) a.st->step = t_pre + 1;           // (no workgroup of this launch reads `step`)
        if (pk == a.npk - 1) live_list_block(pb.src + n_index_words, a.B, a.live);
        for (int i = pk * 256 + threadIdx.x; i < a.in_words; i += a.npk * 256) {
            long long v = pb.src[i];
            a.in_pack[i] = v;
            if (i < n_index_words) {
                int dst;
                if (i < a.B) dst = 2 * M + i * NI;
                else if (i < a.B + n_items) { const int j = i - a.B; dst = 2 * M + (j / a.n_neg) * NI + 1 + (j % a.n_neg); }
                else dst = i - a.B - n_items;
                if (v < 0 || v >= a.n_rows) { atomicOr(a.err, 1); v = 0; }
                a.idx_all[dst] = (int)v;
            }
        }
        for (int p = pk * 256 + threadIdx.x; p < n_c; p += a.npk * 256) {
            int row;
            a.idx_c[p] = pb.at(p, row);
            a.row_c[p] = row;
        }
        return;
    }
    // ---- catch-up over the compact positions (lazy_adam_catchup_pos_kernel with the ids read from the image) ----
#ifdef AMID_EXP_HEAD_NO_CATCHUP     // (variant libraries only, profiles/tools/build_variant.sh: what the launch costs without this role -- 13.1 of 16.6 us)
    return;
#endif
    const int bid = blockIdx.x - nrb - a.npk, nbk = (int)gridDim.x - nrb - a.npk - (W16 ? wr.n * wr.per : 0);
    __shared__ IdleCoef tab[COEF_TAB];
    __shared__ int any_lag;
    StepState st = *a.st;
    st.step = t_pre + 1;                         // (the copy's own `step` word may or may not have been written yet: not used)
    const long long t = st.step;
    const int lane = threadIdx.x & 63, D = a.D;
    auto replay_row = [&](long long r, int l) {
        for (int c = lane; c < (D >> 1); c += 64) {
            const long long off = r * D + 2 * c;
            const float2 p2 = *reinterpret_cast<const float2*>(a.table + off), m2 = *reinterpret_cast<const float2*>(a.m_tab + off),
                         v2 = *reinterpret_cast<const float2*>(a.v_tab + off);
            float p[2] = {p2.x, p2.y}, m[2] = {m2.x, m2.y}, v[2] = {v2.x, v2.y};
            replay_elems<2>(p, m, v, (long long)l + 1, t - 1, st, tab);
            *reinterpret_cast<float2*>(a.table + off) = make_float2(p[0], p[1]);
            *reinterpret_cast<float2*>(a.m_tab + off) = make_float2(m[0], m[1]);
            *reinterpret_cast<float2*>(a.v_tab + off) = make_float2(v[0], v[1]);
        }
    };
    const int n_w = nbk * (blockDim.x >> 6);
    if (threadIdx.x == 0) any_lag = 0;
    __syncthreads();
    // Every position of the wave is resolved ONCE, lane j the wave's j-th (image -> id -> stamp is a chain of three dependent loads: paid
    // side by side, not once per position); the replay loop below reads (id, stamp) back from lane j.  The grid is one round of resident
    // workgroups (amid_step_head_f32): at the headline shape a wave has two positions, a second ROUND of workgroups would pay the whole
    // chain again behind the first.  (Lists of more than 64 positions per wave: the loop repeats per 64.)
    const int wv = threadIdx.x >> 6;
    bool filled = false;                                 // (block-uniform)
    for (long long base = (long long)bid * (blockDim.x >> 6); base < n_c; base += 64LL * n_w) {      // (block-uniform trip count)
        const long long i = base + wv + (long long)lane * n_w;
        int id = 0, l = 0;
        if (i < n_c) {
            int row;
            id = pb.at((int)i, row);
            l = a.last[id];
        }
        const bool lag = l > 0 && l < t - 1;
        const unsigned long long lagging = __ballot(lag);
        if (lagging != 0ull && lane == 0) any_lag = 1;
        // (the table is needed by the waves that replay; a block whose waves all find nothing skips it)
        __syncthreads();
        if (any_lag && !filled) { fill_coef_table(tab, st); filled = true; }
        unsigned long long todo = lagging;
        while (todo != 0ull) {
            const int j = __builtin_ctzll(todo);
            todo &= todo - 1;
            const long long r = __builtin_amdgcn_readlane(id, j);
            const int lj = __builtin_amdgcn_readlane(l, j);
            int won = 0;
            if (lane == 0) won = (atomicCAS(&a.last[r], lj, (int)(t - 1)) == lj) ? 1 : 0;     // claim the row for this wave
            won = __builtin_amdgcn_readfirstlane(won);
            if (!won) continue;
            replay_row(r, lj);
        }
        if (base + 64LL * n_w < n_c) __syncthreads();    // (a wave of the next round must not raise any_lag while a slow one still reads it)
    }
}

// bring every row with pending zero-gradient steps up to date (before eval / checkpoint / parity dumps)
__global__ __launch_bounds__(256) void lazy_adam_flush_kernel(float* __restrict__ table, float* __restrict__ m_tab, float* __restrict__ v_tab,
                                                              int* __restrict__ last, long long n_rows, int D, const StepState* __restrict__ stp) {
    __shared__ IdleCoef tab[COEF_TAB];
    const StepState st = *stp;
    const long long t = st.step;
    const int sub = threadIdx.x & 31;
    const int q = D >> 2;
    const long long hw0 = (long long)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5), n_hw = (long long)gridDim.x * (blockDim.x >> 5);
    fill_coef_table(tab, st);
    for (long long r = hw0; r < n_rows; r += n_hw) {
        const long long l = last[r];
        if (!(l > 0 && l < t)) continue;
        for (int c = sub; c < q; c += 32) {
            const long long off = r * D + 4 * c;
            float4 p = ld4(table + off), m = ld4(m_tab + off), v = ld4(v_tab + off);
            replay_quad(p, m, v, l + 1, t, st, tab);
            st4(table + off, p); st4(m_tab + off, m); st4(v_tab + off, v);
        }
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) last[r] = (int)t;
    }
}

// dense Adam over the flat (non-table) parameter buffer
__global__ __launch_bounds__(256) void adam_dense_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ g, long long n, const StepState* __restrict__ stp,
                                                         float grad_scale) {
    const StepState st = *stp;
    const AdamCoef c = adam_coef_now(st);
    const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (long long i = i0; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
            adam_quad(pp, mm, vv, f4scale(ld4(g + i), grad_scale), c);
            st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
        } else {
            for (long long k = i; k < n; ++k) adam_elem(p[k], m[k], v[k], g[k] * grad_scale, c);
        }
    }
}

__global__ void step_begin_kernel(StepState* st) { st->step += 1; st->step_done = st->step; }

// One launch for the whole optimizer: blocks [0, dense_blocks) run dense Adam over the flat buffer, the rest apply the
// lazy row Adam to the touched table rows (independent memory, so the two roles need no ordering).
__global__ __launch_bounds__(256) void optimizer_step_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                             const float* __restrict__ g, long long n, int dense_blocks,
                                                             float* __restrict__ table, float* __restrict__ m_tab, float* __restrict__ v_tab,
                                                             int* __restrict__ last, const int* __restrict__ uniq_ids,
                                                             const int* __restrict__ n_uniq_p, const float* __restrict__ uniq_grad, int D,
                                                             const StepState* __restrict__ stp, float grad_scale) {
    __shared__ IdleCoef tab[COEF_TAB];
    const StepState st = *stp;
    if ((int)blockIdx.x < dense_blocks) {
        const AdamCoef c = adam_coef_now(st);
        const long long stride = (long long)dense_blocks * blockDim.x * 4;
        for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
            if (i + 4 <= n) {
                float4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
                adam_quad(pp, mm, vv, f4scale(ld4(g + i), grad_scale), c);
                st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
            } else {
                for (long long k = i; k < n; ++k) adam_elem(p[k], m[k], v[k], g[k] * grad_scale, c);
            }
        }
        return;
    }
    const long long t = st.step;
    const int U = *n_uniq_p;
    const int rb = blockIdx.x - dense_blocks, n_rb = gridDim.x - dense_blocks;
    if (rb * 8 >= U) return;
    fill_coef_table(tab, st);
    const AdamCoef cnow = adam_coef_now(st);
    const int sub = threadIdx.x & 31;
    const int q = D >> 2;
    for (int u = rb * 8 + (threadIdx.x >> 5); u < U; u += n_rb * 8) {
        const long long r = uniq_ids[u];
        const long long l = last[r];
        const bool lag = (l > 0 && l < t - 1);
        for (int c = sub; c < q; c += 32) {
            const long long off = r * D + 4 * c;
            float4 pp = ld4(table + off), mm = ld4(m_tab + off), vv = ld4(v_tab + off);
            if (lag) replay_quad(pp, mm, vv, l + 1, t - 1, st, tab);
            adam_quad(pp, mm, vv, f4scale(ld4(uniq_grad + (long long)u * D + 4 * c), grad_scale), cnow);
            st4(table + off, pp); st4(m_tab + off, mm); st4(v_tab + off, vv);
        }
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) last[r] = (int)t;
    }
}

// optimizer_step_kernel with phase B of the segment reduce riding in front (the live-sequence step: amid_grad_tail_live_f32 runs phase A
// only).  Workgroups [0, nch): chunk c of the sorted list -- if it owns a run that crosses chunk borders (seg_spans.h: the pad row's, a
// few others) it adds the run's partial rows up (the same additions in the same order as segreduce_spans_kernel: the same bits), writes
// the sum to uniq_grad and applies the row's Adam step on the spot; the row workgroups skip those runs.  [nch, nch + dense_blocks): dense
// Adam; the rest: the rows whose runs phase A finished.
template <int VEC>
__global__ __launch_bounds__(256) void optimizer_step_spans_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                   const float* __restrict__ g, long long n, int dense_blocks,
                                                                   float* __restrict__ table, float* __restrict__ m_tab, float* __restrict__ v_tab,
                                                                   int* __restrict__ last, const int* __restrict__ uniq_ids,
                                                                   const int* __restrict__ n_uniq_p, float* __restrict__ uniq_grad,
                                                                   const StepState* __restrict__ stp, float grad_scale,
                                                                   const int* __restrict__ seg_off, const int* __restrict__ seg_of, int n_sorted,
                                                                   const float* __restrict__ partial, int nch) {
    constexpr int D = VEC * 64;
    __shared__ IdleCoef tab[COEF_TAB];
    const StepState st = *stp;
    if ((int)blockIdx.x < nch) {
        __shared__ float red[16][D];
        int u, c_last;
        if (!spans_owner(blockIdx.x, seg_off, seg_of, n_sorted, SEG_CHUNK, u, c_last)) return;
        const long long t = st.step;
        const long long r = uniq_ids[u];
        const long long l = last[r];
        const bool lag = (l > 0 && l < t - 1);                       // (block-uniform: one row.  The pad row, the usual owner, never lags)
        spans_partials<VEC, 4>(red, blockIdx.x, c_last, partial);
        if (lag) fill_coef_table(tab, st);
        __syncthreads();                                             // red is complete
        const AdamCoef cnow = adam_coef_now(st);
        for (int c = threadIdx.x; c < D / 4; c += blockDim.x) {
            const float4 gs = make_float4(spans_total<VEC>(red, 4 * c), spans_total<VEC>(red, 4 * c + 1), spans_total<VEC>(red, 4 * c + 2),
                                          spans_total<VEC>(red, 4 * c + 3));
            st4(uniq_grad + (long long)u * D + 4 * c, gs);
            const long long off = r * D + 4 * c;
            float4 pp = ld4(table + off), mm = ld4(m_tab + off), vv = ld4(v_tab + off);
            if (lag) replay_quad(pp, mm, vv, l + 1, t - 1, st, tab);
            adam_quad(pp, mm, vv, f4scale(gs, grad_scale), cnow);
            st4(table + off, pp); st4(m_tab + off, mm); st4(v_tab + off, vv);
        }
        if (threadIdx.x == 0) last[r] = (int)t;
        return;
    }
    const int bid = blockIdx.x - nch;
    if (bid < dense_blocks) {
        const AdamCoef c = adam_coef_now(st);
        const long long stride = (long long)dense_blocks * blockDim.x * 4;
        for (long long i = ((long long)bid * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
            if (i + 4 <= n) {
                float4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
                adam_quad(pp, mm, vv, f4scale(ld4(g + i), grad_scale), c);
                st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
            } else {
                for (long long k = i; k < n; ++k) adam_elem(p[k], m[k], v[k], g[k] * grad_scale, c);
            }
        }
        return;
    }
    const long long t = st.step;
    const int U = *n_uniq_p;
    const int rb = bid - dense_blocks, n_rb = gridDim.x - nch - dense_blocks;
    if (rb * 8 >= U) return;
    fill_coef_table(tab, st);
    const AdamCoef cnow = adam_coef_now(st);
    const int sub = threadIdx.x & 31;
    const int q = D >> 2;
    for (int u = rb * 8 + (threadIdx.x >> 5); u < U; u += n_rb * 8) {
        if (seg_off[u] / SEG_CHUNK != (seg_off[u + 1] - 1) / SEG_CHUNK) continue;      // crosses chunks: a spans workgroup's (above)
        const long long r = uniq_ids[u];
        const long long l = last[r];
        const bool lag = (l > 0 && l < t - 1);
        for (int c = sub; c < q; c += 32) {
            const long long off = r * D + 4 * c;
            float4 pp = ld4(table + off), mm = ld4(m_tab + off), vv = ld4(v_tab + off);
            if (lag) replay_quad(pp, mm, vv, l + 1, t - 1, st, tab);
            adam_quad(pp, mm, vv, f4scale(ld4(uniq_grad + (long long)u * D + 4 * c), grad_scale), cnow);
            st4(table + off, pp); st4(m_tab + off, mm); st4(v_tab + off, vv);
        }
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) last[r] = (int)t;
    }
}

// The data-parallel optimizer: the whole of "graph B" as ONE launch, reading the world's gathered chunks directly.  Chunk r (at
// gathered + r * chunk_floats) = [id rows: umax ascending unique ids, sentinel-padded | umax gradient rows | the rank's flat dense
// gradient].  Dense role: g = the ranks' dense parts summed in rank order, then Adam.  Row role: entry (r, i) is applied by the
// FIRST rank whose list holds its id (a binary search of every earlier list), with the rows of the later lists that hold the same
// id added in rank order -- the same sums on every rank, hence bit-identical replicas, without the merge / segment-reduce launches.
__device__ __forceinline__ int find_id(const int* __restrict__ ids, int n, int x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ids[mid] < x) lo = mid + 1; else hi = mid;
    }
    return (lo < n && ids[lo] == x) ? lo : -1;
}

__global__ __launch_bounds__(256) void optimizer_gathered_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                 float* __restrict__ g, long long n, int dense_blocks,
                                                                 float* __restrict__ table, float* __restrict__ m_tab, float* __restrict__ v_tab,
                                                                 int* __restrict__ last, const float* __restrict__ gathered, int world, int umax,
                                                                 long long chunk_floats, int id_rows, long long dense_off, int D, int sentinel,
                                                                 const StepState* __restrict__ stp, float grad_scale) {
    __shared__ IdleCoef tab[COEF_TAB];
    const StepState st = *stp;
    if ((int)blockIdx.x < dense_blocks) {
        const AdamCoef c = adam_coef_now(st);
        const long long stride = (long long)dense_blocks * blockDim.x * 4;
        const float* d0 = gathered + dense_off;
        for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
            if (i + 4 <= n) {
                float4 gg;
                if (dense_off < 0) {                                   // the dense gradient was all-reduced into g by the caller
                    gg = ld4(g + i);
                } else {
                    gg = ld4(d0 + i);
                    for (int r = 1; r < world; ++r) {
                        const float4 t = ld4(d0 + r * chunk_floats + i);
                        gg.x += t.x; gg.y += t.y; gg.z += t.z; gg.w += t.w;
                    }
                    st4(g + i, gg);
                }
                float4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
                adam_quad(pp, mm, vv, f4scale(gg, grad_scale), c);
                st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
            } else {
                for (long long k = i; k < n; ++k) {
                    float gs = dense_off < 0 ? g[k] : d0[k];
                    for (int r = 1; dense_off >= 0 && r < world; ++r) gs += d0[r * chunk_floats + k];
                    g[k] = gs;
                    adam_elem(p[k], m[k], v[k], gs * grad_scale, c);
                }
            }
        }
        return;
    }
    const long long t = st.step;
    const int rb = blockIdx.x - dense_blocks, n_rb = gridDim.x - dense_blocks;
    const int total = world * umax;
    if (rb * 8 >= total) return;
    fill_coef_table(tab, st);
    const AdamCoef cnow = adam_coef_now(st);
    const int sub = threadIdx.x & 31;
    const int q = D >> 2;
    for (int u = rb * 8 + (threadIdx.x >> 5); u < total; u += n_rb * 8) {
        const int r0 = u / umax, i0 = u - r0 * umax;
        const int x = ((const int*)(gathered + r0 * chunk_floats))[i0];
        if (x >= sentinel) continue;                                   // padding
        // lane `sub` of the half-wave searches rank sub's list; the half-wave then shares what was found
        const int half = (threadIdx.x >> 5) & 1;
        int j = -1;
        if (sub < world && sub != r0) j = find_id((const int*)(gathered + sub * chunk_floats), umax, x);
        const unsigned int fm = (unsigned int)(__ballot(j >= 0) >> (32 * half));
        if (fm & ((1u << r0) - 1u)) continue;                          // an earlier rank's entry applies this id
        const long long row = x;
        const long long l = last[row];
        const bool lag = (l > 0 && l < t - 1);
        for (int c = sub; c < q; c += 32) {
            float4 gg = ld4(gathered + r0 * chunk_floats + (long long)(id_rows + i0) * D + 4 * c);
            for (int r = r0 + 1; r < world; ++r) {                     // the later ranks' rows of the same id, in rank order
                const int jr = __shfl(j, 32 * half + r);
                if ((fm >> r) & 1u) {
                    const float4 tt = ld4(gathered + r * chunk_floats + (long long)(id_rows + jr) * D + 4 * c);
                    gg.x += tt.x; gg.y += tt.y; gg.z += tt.z; gg.w += tt.w;
                }
            }
            const long long off = row * D + 4 * c;
            float4 pp = ld4(table + off), mm = ld4(m_tab + off), vv = ld4(v_tab + off);
            if (lag) replay_quad(pp, mm, vv, l + 1, t - 1, st, tab);
            adam_quad(pp, mm, vv, f4scale(gg, grad_scale), cnow);
            st4(table + off, pp); st4(m_tab + off, mm); st4(v_tab + off, vv);
        }
        __builtin_amdgcn_wave_barrier();
        if (sub == 0) last[row] = (int)t;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// The gradient tail WITH the optimizer (round 6, amid_grad_tail_opt_f32): the folded step's last two launches -- amid_grad_tail_live_f32 and
// amid_optimizer_step_spans_f32 -- as one.  Every producer of a gradient slice applies Adam to it on the spot:
//   row workers     a half-wave per unique row u whose run lies inside one 64-entry chunk of the sorted list: the run's gradient rows added in
//                   list order (what phase A's chunk wave did), uniq_grad[u] written, the row's Adam step;
//   chunk blocks    the pieces of the runs that CROSS chunk borders (the pad row's ~180 chunks, a few others), one wave per chunk, stored with
//                   agent scope; the block that takes the last ticket finds the crossing runs' owner chunks, adds every run's pieces in
//                   spans_partials' order (the additions of the spans launch, the same bits) and applies the rows -- nobody waits, nobody spins;
//   partial sums    the fixed-order sum of every dense partial (reduce_partials.h) with dense Adam on the finished slice;
//   position rows   pos_sum_block with dense Adam on the slice;
//   the rest        dense Adam on the slots whose gradients an earlier launch finished (the scorer's: sasrec_strip.hip's riders).
// Same additions in the same order and the same Adam arithmetic as the two launches: the same bits (tests/test_gpu_timed_path.py).
struct DenseAdamSink {
    float* p; float* m; float* v; const float* g0; long long n; AdamCoef c; float gs;
    __device__ __forceinline__ void quad(float* dst, float4 g) const {
        const long long off = dst - g0;
        if (off < 0 || off + 4 > n) return;                   // (a sum that is no parameter's gradient: the loss)
        float4 pp = ld4(p + off), mm = ld4(m + off), vv = ld4(v + off);
        adam_quad(pp, mm, vv, f4scale(g, gs), c);
        st4(p + off, pp); st4(m + off, mm); st4(v + off, vv);
    }
    __device__ __forceinline__ void one(float* dst, float g) const {
        const long long off = dst - g0;
        if (off < 0 || off >= n) return;
        adam_elem(p[off], m[off], v[off], g * gs, c);
    }
};

// the data-parallel form (grad_tail_opt_kernel<VEC, false>): nothing is applied -- a finished dense slice is ALSO written into this rank's
// exchange chunk (dst == nullptr: the caller all-reduces the dense gradient itself), the rows only go to uniq_grad (which points into the chunk)
struct DenseShipSink {
    const float* g0; long long n; float* dst;
    __device__ __forceinline__ void quad(float* d, float4 g) const {
        const long long off = d - g0;
        if (dst == nullptr || off < 0 || off + 4 > n) return;
        st4(dst + off, g);
    }
    __device__ __forceinline__ void one(float* d, float g) const {
        const long long off = d - g0;
        if (dst == nullptr || off < 0 || off >= n) return;
        dst[off] = g;
    }
};

struct TailOptArgs {
    const float* grad_rows; const int* pos_sorted; const int* seg_off; const int* seg_of; int n_sorted;
    float* uniq_grad; float* partial; int n_seg, nch; int* ticket;
    const ReduceEntry* entries; const int* blk_off; int n_entries, n_red;
    PosSum ps;
    float* p; float* m; float* v; float* g; long long n_dense; long long left_lo, left_hi; int left_blocks;
    float* table; float* m_tab; float* v_tab; int* last; const int* uniq_ids; const int* n_uniq; int row_blocks;
    const StepState* st; float grad_scale;
    // ship mode: the exchange chunk's id part and dense part
    float* dense_dst; int* out_ids; int n_out, pad_id, id_blocks; int* err;
};

// APPLY: Adam on the spot (the single-GPU step).  !APPLY: the data-parallel step's local half -- the same sums, shipped instead of applied
// (amid_grad_tail_live_dp_f32): rows to uniq_grad = the exchange chunk's row part, dense slices also to its dense part, and id_blocks more
// workgroups pad the unique ids into its id part.
template <int VEC, bool APPLY>
__global__ __launch_bounds__(256) void grad_tail_opt_kernel(const TailOptArgs a) {
    constexpr int D = VEC * 64;
    __shared__ IdleCoef tab[COEF_TAB];
    StepState st = {};
    if constexpr (APPLY) st = *a.st;
    const long long t = st.step;
    int bid = blockIdx.x;
    if (bid < a.n_seg) {
        // ---- chunk blocks: the cut pieces, then a ticket; the last block finishes the crossing runs
        segreduce_chunks_block<VEC, true>(a.grad_rows, a.pos_sorted, a.seg_of, a.n_sorted, nullptr, a.partial, bid, SEG_CHUNK, a.seg_off);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's agent-scope stores are acknowledged
        __shared__ int s_last, s_n;
        __shared__ int owners[256];
        __shared__ float red[16][D];
        __syncthreads();
        if (threadIdx.x == 0) s_last = (atomicAdd(a.ticket, 1) == a.n_seg - 1) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        if (threadIdx.x == 0) *a.ticket = 0;                        // (for the next launch: nobody of this one reads it again)
        AdamCoef cnow = {};
        if constexpr (APPLY) cnow = adam_coef_now(st);
        bool tab_filled = false;
        for (int c0 = 0; c0 < a.nch; c0 += 256) {                   // owner chunks, 256 candidates a round
            if (threadIdx.x == 0) s_n = 0;
            __syncthreads();
            {
                const int c = c0 + (int)threadIdx.x;
                int u, cl;
                if (c < a.nch && spans_owner(c, a.seg_off, a.seg_of, a.n_sorted, SEG_CHUNK, u, cl) && a.seg_off[u + 1] - a.seg_off[u] > SEG_CHUNK)
                    owners[atomicAdd(&s_n, 1)] = c;                 // (a crossing run of at most a chunk's length is a row worker's)
            }
            __syncthreads();
            const int n_own = s_n;
            for (int k = 0; k < n_own; ++k) {                       // (block-uniform; the runs are independent rows: any order)
                const int c = owners[k];
                int u = 0, c_last = 0;
                spans_owner(c, a.seg_off, a.seg_of, a.n_sorted, SEG_CHUNK, u, c_last);
                long long r = 0, l = 0;
                if constexpr (APPLY) { r = a.uniq_ids[u]; l = a.last[r]; }
                const bool lag = APPLY && (l > 0 && l < t - 1);    // (block-uniform: one row.  The pad row, the usual owner, never lags)
                // (the row's parameter and moments requested in front of the pieces: D / 4 <= 64 quads, one per thread of the first wave)
                const int cq0 = threadIdx.x;
                float4 pp0 = make_float4(0.f, 0.f, 0.f, 0.f), mm0 = pp0, vv0 = pp0;
                if constexpr (APPLY) {
                    if (cq0 < D / 4) { const long long off = r * D + 4 * cq0; pp0 = ld4(a.table + off); mm0 = ld4(a.m_tab + off); vv0 = ld4(a.v_tab + off); }
                }
                spans_partials<VEC, 4, true>(red, c, c_last, a.partial);
                if (lag && !tab_filled) { fill_coef_table(tab, st); tab_filled = true; }
                __syncthreads();                                     // red is complete
                for (int cq = threadIdx.x; cq < D / 4; cq += blockDim.x) {
                    const float4 gs = make_float4(spans_total<VEC>(red, 4 * cq), spans_total<VEC>(red, 4 * cq + 1), spans_total<VEC>(red, 4 * cq + 2),
                                                  spans_total<VEC>(red, 4 * cq + 3));
                    st4(a.uniq_grad + (long long)u * D + 4 * cq, gs);
                    if constexpr (APPLY) {
                        const long long off = r * D + 4 * cq;
                        float4 pp = pp0, mm = mm0, vv = vv0;
                        if (lag) replay_quad(pp, mm, vv, l + 1, t - 1, st, tab);
                        adam_quad(pp, mm, vv, f4scale(gs, a.grad_scale), cnow);
                        st4(a.table + off, pp); st4(a.m_tab + off, mm); st4(a.v_tab + off, vv);
                    }
                }
                if constexpr (APPLY) { if (threadIdx.x == 0) a.last[r] = (int)t; }
                __syncthreads();                                     // red is free
            }
        }
        return;
    }
    bid -= a.n_seg;
    auto make_sink = [&]() {
        if constexpr (APPLY) {
            DenseAdamSink k;
            k.p = a.p; k.m = a.m; k.v = a.v; k.g0 = a.g; k.n = a.n_dense; k.c = adam_coef_now(st); k.gs = a.grad_scale;
            return k;
        } else {
            return DenseShipSink{a.g, a.n_dense, a.dense_dst};
        }
    };
    const auto sink = make_sink();
    if (bid < a.n_red) {
        int lo = 0, hi = a.n_entries;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.blk_off[mid] <= bid) lo = mid; else hi = mid;
        }
        reduce_partials_block(a.entries[lo], bid - a.blk_off[lo], a.blk_off[lo + 1] - a.blk_off[lo], sink);
        return;
    }
    bid -= a.n_red;
    if (bid < 2 * a.ps.nblk) {
        pos_sum_block(a.ps, D, bid / a.ps.nblk, bid % a.ps.nblk, a.ps.nblk, sink);
        return;
    }
    bid -= 2 * a.ps.nblk;
    // ---- the dense slots an earlier launch finished: [left_lo, left_hi) of the flat buffer
    if (bid < a.left_blocks) {
        for (long long i = a.left_lo + ((long long)bid * blockDim.x + threadIdx.x) * 4; i < a.left_hi; i += (long long)a.left_blocks * blockDim.x * 4) {
            if (i + 4 <= a.left_hi) sink.quad(a.g + i, ld4(a.g + i));
            else for (long long k = i; k < a.left_hi; ++k) sink.one(a.g + k, a.g[k]);
        }
        return;
    }
    bid -= a.left_blocks;
    if constexpr (!APPLY) {
        // ---- the chunk's id part: the unique ids padded to n_out (segreduce.hip segreduce_spans_pack_kernel's role)
        if (bid < a.id_blocks) {
            const int rr = bid * 256 + (int)threadIdx.x;
            if (rr < a.n_out) a.out_ids[rr] = rr < *a.n_uniq ? a.uniq_ids[rr] : a.pad_id;
            // more unique rows than the caller's bound: rows were written past the chunk's row part -- the step is corrupt; say so
            if (rr == 0 && a.err != nullptr && *a.n_uniq > a.n_out) atomicOr(a.err, AMID_FLAG_UMAX_EXCEEDED);
            return;
        }
        bid -= a.id_blocks;
    }
    {
        // ---- row workers: the runs inside one chunk, summed and applied by a half-wave each
        const int U = *a.n_uniq;
        if (bid * 8 >= U) return;
        AdamCoef cnow = {};
        if constexpr (APPLY) { fill_coef_table(tab, st); cnow = adam_coef_now(st); }
        const int sub = threadIdx.x & 31;
        constexpr int q = D >> 2;
        for (int u = bid * 8 + (threadIdx.x >> 5); u < U; u += a.row_blocks * 8) {
            const int s0 = a.seg_off[u], s1 = a.seg_off[u + 1];
            if (s1 - s0 > SEG_CHUNK) continue;                     // longer than a chunk: the chunk blocks' pieces, the last chunk block's sum (above)
            // a run of at most a chunk's length lies inside one chunk or crosses ONE border: [s0, mid) and [mid, s1)
            const int mid = min(s1, (s0 / SEG_CHUNK + 1) * SEG_CHUNK);
            long long r = 0, l = 0;
            if constexpr (APPLY) { r = a.uniq_ids[u]; l = a.last[r]; }
            const bool lag = APPLY && (l > 0 && l < t - 1);
            for (int c = sub; c < q; c += 32) {
                const long long off = r * D + 4 * c;
                float4 pp = make_float4(0.f, 0.f, 0.f, 0.f), mm = pp, vv = pp;
                if constexpr (APPLY) { pp = ld4(a.table + off); mm = ld4(a.m_tab + off); vv = ld4(a.v_tab + off); }      // (in flight under the run's rows)
                auto piece = [&](int e, const int e_end) {         // rows e .. e_end - 1 added in list order from zero (a chunk wave's sum), eight in flight
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (; e + 8 <= e_end; e += 8) {
                        int ps_[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) ps_[j] = a.pos_sorted[e + j];
                        float4 rw[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) rw[j] = ld4(a.grad_rows + (long long)ps_[j] * D + 4 * c);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc = f4add(acc, rw[j]);
                    }
                    for (; e < e_end; ++e) acc = f4add(acc, ld4(a.grad_rows + (long long)a.pos_sorted[e] * D + 4 * c));
                    return acc;
                };
                float4 gs = piece(s0, mid);
                if (mid < s1) {
                    // the spans launch's additions for a run of two pieces: red[0] = 0 + piece 0, red[1] = 0 + piece 1, total = ((0 + red[0]) +
                    // red[1]) + fourteen zeros (seg_spans.h spans_partials / spans_total) -- spelled out, so that the bits are the same
                    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 r0 = f4add(z, gs), r1 = f4add(z, piece(mid, s1));
                    gs = f4add(f4add(z, r0), r1);
                }
                st4(a.uniq_grad + (long long)u * D + 4 * c, gs);
                if constexpr (APPLY) {
                    if (lag) replay_quad(pp, mm, vv, l + 1, t - 1, st, tab);
                    adam_quad(pp, mm, vv, f4scale(gs, a.grad_scale), cnow);
                    st4(a.table + off, pp); st4(a.m_tab + off, mm); st4(a.v_tab + off, vv);
                }
            }
            if constexpr (APPLY) {
                __builtin_amdgcn_wave_barrier();
                if (sub == 0) a.last[r] = (int)t;
            }
        }
        return;
    }
}

}  // namespace amid

using namespace amid;

extern "C" int amid_step_state_bytes(void) { return (int)sizeof(StepState); }

// host-side fill helper: writes a StepState image into `host_buf` (caller copies it to the device)
extern "C" int amid_step_state_pack(void* host_buf, unsigned long long seed, long long step, double lr, double beta1, double beta2, double eps) {
    AMID_CHECK_ARG(host_buf);
    StepState s;
    s.seed = seed; s.step = step; s.lr = lr; s.beta1 = beta1; s.beta2 = beta2; s.eps = eps; s.ticket = 0; s.pad_ = 0; s.step_done = step;
    *(StepState*)host_buf = s;
    return AMID_OK;
}

extern "C" int amid_step_begin(void* step_state, void* stream) {
    AMID_CHECK_ARG(step_state);
    step_begin_kernel<<<1, 1, 0, (hipStream_t)stream>>>((StepState*)step_state);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static inline int rows_grid(long long n_rows_hint) {
    long long b = (n_rows_hint + 7) / 8;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

extern "C" int amid_lazy_adam_catchup_f32(float* table, float* m, float* v, int* last, const int* uniq_ids, const int* n_uniq,
                                          int n_uniq_max, int D, const void* step_state, void* stream) {
    AMID_CHECK_ARG(table && m && v && last && uniq_ids && n_uniq && step_state && D > 0 && (D % 4) == 0 && n_uniq_max > 0);
    lazy_adam_rows_kernel<0><<<rows_grid(n_uniq_max), 256, 0, (hipStream_t)stream>>>(table, m, v, last, uniq_ids, n_uniq, nullptr, D,
                                                                                      (const StepState*)step_state, 1.0f);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_lazy_adam_apply_f32(float* table, float* m, float* v, int* last, const int* uniq_ids, const int* n_uniq,
                                        int n_uniq_max, const float* uniq_grad, float grad_scale, int D, const void* step_state, void* stream) {
    AMID_CHECK_ARG(table && m && v && last && uniq_ids && n_uniq && uniq_grad && step_state && D > 0 && (D % 4) == 0 && n_uniq_max > 0);
    lazy_adam_rows_kernel<1><<<rows_grid(n_uniq_max), 256, 0, (hipStream_t)stream>>>(table, m, v, last, uniq_ids, n_uniq, uniq_grad, D,
                                                                                      (const StepState*)step_state, grad_scale);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_lazy_adam_flush_f32(float* table, float* m, float* v, int* last, long long n_rows, int D, const void* step_state,
                                        void* stream) {
    AMID_CHECK_ARG(table && m && v && last && step_state && n_rows > 0 && D > 0 && (D % 4) == 0);
    lazy_adam_flush_kernel<<<rows_grid(n_rows), 256, 0, (hipStream_t)stream>>>(table, m, v, last, n_rows, D, (const StepState*)step_state);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_adam_dense_f32(float* p, float* m, float* v, const float* g, long long n, float grad_scale, const void* step_state,
                                   void* stream) {
    AMID_CHECK_ARG(p && m && v && g && step_state && n > 0);
    long long b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    adam_dense_kernel<<<(int)b, 256, 0, (hipStream_t)stream>>>(p, m, v, g, n, (const StepState*)step_state, grad_scale);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_optimizer_step_f32(float* p, float* m, float* v, const float* g, long long n, float* table, float* m_tab, float* v_tab,
                                       int* last, const int* uniq_ids, const int* n_uniq, int n_uniq_max, const float* uniq_grad, int D,
                                       float grad_scale, const void* step_state, void* stream) {
    AMID_CHECK_ARG(p && m && v && g && n > 0 && table && m_tab && v_tab && last && uniq_ids && n_uniq && uniq_grad && step_state && D > 0 &&
                   (D % 4) == 0 && n_uniq_max > 0);
    long long db = (n / 4 + 255) / 256;
    if (db < 1) db = 1;
    if (db > 1024) db = 1024;
    const int rb = rows_grid(n_uniq_max);
    optimizer_step_kernel<<<(int)db + rb, 256, 0, (hipStream_t)stream>>>(p, m, v, g, n, (int)db, table, m_tab, v_tab, last, uniq_ids, n_uniq,
                                                                          uniq_grad, D, (const StepState*)step_state, grad_scale);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_optimizer_step_gathered_f32(float* p, float* m, float* v, float* g, long long n, float* table, float* m_tab, float* v_tab,
                                                int* last, const float* gathered, int world, int umax, long long chunk_floats, int id_rows,
                                                long long dense_off, int D, int sentinel, float grad_scale, const void* step_state,
                                                void* stream) {
    AMID_CHECK_ARG(p && m && v && g && n > 0 && table && m_tab && v_tab && last && gathered && step_state && D > 0 && (D % 4) == 0);
    // dense_off < 0: the chunks carry no dense part -- g already holds the world's summed dense gradient (the caller's all-reduce)
    AMID_CHECK_ARG(world > 0 && world <= 16 && umax > 0 && id_rows * (long long)D >= umax && (chunk_floats % 4) == 0 &&
                   (dense_off < 0 ? chunk_floats >= (long long)(id_rows + umax) * D
                                  : (dense_off >= (long long)(id_rows + umax) * D && (dense_off % 4) == 0 && chunk_floats >= dense_off + n)));
    long long db = (n / 4 + 255) / 256;
    if (db < 1) db = 1;
    if (db > 1024) db = 1024;
    const int rb = rows_grid((long long)world * umax);
    optimizer_gathered_kernel<<<(int)db + rb, 256, 0, (hipStream_t)stream>>>(p, m, v, g, n, (int)db, table, m_tab, v_tab, last, gathered, world,
                                                                              umax, chunk_floats, id_rows, dense_off, D, sentinel,
                                                                              (const StepState*)step_state, grad_scale);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int catchup_positions(float* table, float* m, float* v, int* last, const int* idx, int n_idx, int D, const void* step_state,
                             const void* sort_plan, int sort_phase, void* stream) {
    AMID_CHECK_ARG(table && m && v && last && idx && step_state && D > 0 && (D % 4) == 0 && n_idx > 0);
    // the kernel is a chain of dependent loads (idx -> stamp -> row) and then, for a lagging row, a long serial replay: more waves in
    // flight beat fewer, fatter ones (measured: 2048-block cap 24.8 us at cfg 2 / 388 us at cfg 5, 16384: 22.4 / 341; cfg 4 in steady
    // state, 480-step gaps: 8 positions per block 0.2596 ms per step, 4: 0.2533, 16: 0.2634)
    long long blocks = ((long long)n_idx + 3) / 4;      // a position per wave: a wave that got two lagging rows would replay them one after the other
    if (blocks > 16384) blocks = 16384;
    SortRider rd;
    rd.phase = 0;
    if (sort_plan != nullptr) {
        if (sort_phase != 1 && sort_phase != SORT_CHAIN_PHASE) return AMID_ERR_UNSUPPORTED;      // this launch carries phase 1, or the whole sort
        rd.plan = *(const SortPlan*)sort_plan;
        rd.phase = sort_phase;
        // the chain: 1 024-bin plans of at most SORT_CHAIN_MAX_BLOCKS tiles (every rider resident at once: sort_phases.h)
        if (sort_phase == SORT_CHAIN_PHASE && (rd.plan.g0.bits > 10 || rd.plan.nblk > SORT_CHAIN_MAX_BLOCKS)) return AMID_ERR_UNSUPPORTED;
    }
    lazy_adam_catchup_pos_kernel<<<(int)blocks + rider_blocks_host(rd), 256, 0, (hipStream_t)stream>>>(table, m, v, last, idx, n_idx, D,
                                                                                     (const StepState*)step_state, rd);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_lazy_adam_catchup_positions_f32(float* table, float* m, float* v, int* last, const int* idx, int n_idx, int D,
                                                    const void* step_state, void* stream) {
    return catchup_positions(table, m, v, last, idx, n_idx, D, step_state, nullptr, 0, stream);
}

// the same launch carrying phase `sort_phase` of a sort plan (amid_sort_plan_pack) as extra workgroups: 1, or 6 = ALL five phases chained in this
// one launch (keys below 2^20, at most 64 tiles = 131 072 indices: amid_sort_chain_max_indices; AMID_ERR_UNSUPPORTED otherwise)
extern "C" int amid_sort_chain_max_indices(void) { return SORT_CHAIN_MAX_BLOCKS * SORT_TILE; }
extern "C" int amid_lazy_adam_catchup_positions_sort_f32(float* table, float* m, float* v, int* last, const int* idx, int n_idx, int D,
                                                         const void* step_state, const void* sort_plan, int sort_phase, void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr);
    return catchup_positions(table, m, v, last, idx, n_idx, D, step_state, sort_plan, sort_phase, stream);
}

// The head of a train step on an input pool as ONE launch (step_head_kernel): batch picked by the device step counter, mirrored into the
// plan's input words, full index list, live list, COMPACT index list (idx_c / row_c: B T + B (1 + n_neg) entries -- every sample's
// own-domain sequence, then the items), the lazy-Adam catch-up of the compact list's rows, phase 1 of the sort plan built on
// (idx_c, row_c) (amid_sort_plan_pack), and the step counter's bump (StepState::step; the caller's next launch must be an embed_fwd
// launch, which re-joins StepState::step_done).  replaces: amid_pack_indices_pool_live + amid_lazy_adam_catchup_positions_sort_f32.
static int step_head(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack, int in_words,
                                  int B, int T, int n_neg, long long n_rows, int* idx_all, int* idx_c, int* row_c, int* live, int* err_flag,
                                  float* table, float* m, float* v, int* last, int D, void* step_state, const void* sort_plan, void* stream,
                                  const W16Rider* wrp) {
    AMID_CHECK_ARG(pool && n_pool > 0 && in_pack && idx_all && idx_c && row_c && live && err_flag && table && m && v && last && step_state &&
                   B > 0 && T > 0 && n_neg > 0 && D > 0 && (D % 4) == 0);
    AMID_CHECK_ARG(in_words >= B + B * n_neg + 2 * B * T + B);
    StepHeadArgs a;
    a.pool = pool; a.stride = pool_stride; a.n_pool = n_pool; a.phase = phase; a.in_pack = in_pack; a.in_words = in_words;
    a.B = B; a.T = T; a.n_neg = n_neg; a.n_rows = n_rows; a.idx_all = idx_all; a.idx_c = idx_c; a.row_c = row_c; a.live = live; a.err = err_flag;
    a.table = table; a.m_tab = m; a.v_tab = v; a.last = last; a.D = D; a.st = (StepState*)step_state;
    int npk = (in_words + 1023) / 1024;
    if (npk < 2) npk = 2;
    if (npk > 64) npk = 64;
    a.npk = npk;
    SortRider rd;
    rd.phase = 0;
    const int n_c = B * T + B * (1 + n_neg);
    if (sort_plan != nullptr) {
        rd.plan = *(const SortPlan*)sort_plan;
        rd.phase = 1;
        if (rd.plan.n != n_c) return AMID_ERR_ARG;                  // the plan must be the compact list's
    }
    long long blocks = ((long long)n_c + 3) / 4;                    // a position per wave ...
    // ... up to ONE round of resident workgroups (8 of 256 threads per CU) of the CURRENT device: an attribute query per call (no
    // process-wide cache: a process may drive devices of different sizes from several threads; the query is legal under stream capture)
    int resident = 2048;
    {
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0)
            resident = 8 * n_cu;
    }
    // (fewer, longer catch-up workgroups measured slower: 1 024 blocks 16.8 us as here, 512 19.4, 256 20.1)
    const long long room = resident - rider_blocks_host(rd) - npk;
    if (room >= 256 && blocks > room) blocks = room;
    if (blocks > 16384) blocks = 16384;
    if (wrp != nullptr) {
        const long long room2 = room - (long long)wrp->n * wrp->per;      // (the riders are resident workgroups too)
        if (room2 >= 256 && blocks > room2) blocks = room2;
        step_head_kernel<true><<<rider_blocks_host(rd) + npk + (int)blocks + wrp->n * wrp->per, 256, 0, (hipStream_t)stream>>>(a, rd, *wrp);
    } else {
        step_head_kernel<false><<<rider_blocks_host(rd) + npk + (int)blocks, 256, 0, (hipStream_t)stream>>>(a, rd, W16Rider{});
    }
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
extern "C" int amid_step_head_f32(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack, int in_words,
                                  int B, int T, int n_neg, long long n_rows, int* idx_all, int* idx_c, int* row_c, int* live, int* err_flag,
                                  float* table, float* m, float* v, int* last, int D, void* step_state, const void* sort_plan, void* stream) {
    return step_head(pool, pool_stride, n_pool, phase, in_pack, in_words, B, T, n_neg, n_rows, idx_all, idx_c, row_c, live, err_flag, table, m, v, last,
                     D, step_state, sort_plan, stream, nullptr);
}
// amid_step_head_f32 + the step's weight images by extra workgroups (amid_embed_fwd_w16_f32's riders: w_src = n_w square [D][D] fp32 weights,
// w16_dst [n_w][planes][D][D] bf16 their fragment images, w16t_dst (optional) the same of their transposes): for a step whose gather is the
// prologue of its forward (amid_sas_seq_fwd_gather_*_f32), which re-joins StepState::step_done.  D = 128.
extern "C" int amid_step_head_w16_f32(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack, int in_words,
                                      int B, int T, int n_neg, long long n_rows, int* idx_all, int* idx_c, int* row_c, int* live, int* err_flag,
                                      float* table, float* m, float* v, int* last, int D, void* step_state, const void* sort_plan,
                                      const float* const* w_src, int n_w, int w_planes, void* w16_dst, void* w16t_dst, void* stream) {
    AMID_CHECK_ARG(w_src && n_w > 0 && (w_planes == 1 || w_planes == 3) && w16_dst && D == 128 && n_w * (w16t_dst ? 2 : 1) <= W16_MAX);
    W16Rider wr = {};
    for (int i = 0; i < n_w; ++i) { AMID_CHECK_ARG(w_src[i]); wr.src[i] = w_src[i]; wr.ld[i] = (unsigned short)D; wr.tr[i] = 0; }
    wr.n = wr.n_fwd = n_w;
    if (w16t_dst != nullptr) {
        for (int i = 0; i < n_w; ++i) { wr.src[n_w + i] = w_src[i]; wr.ld[n_w + i] = (unsigned short)D; wr.tr[n_w + i] = 1; }
        wr.n = 2 * n_w;
    }
    wr.dst = (unsigned short*)w16_dst; wr.dstT = (unsigned short*)w16t_dst; wr.planes = w_planes; wr.D = D; wr.per = (D * (D / 8) + 255) / 256;
    return step_head(pool, pool_stride, n_pool, phase, in_pack, in_words, B, T, n_neg, n_rows, idx_all, idx_c, row_c, live, err_flag, table, m, v, last,
                     D, step_state, sort_plan, stream, &wr);
}

// amid_optimizer_step_f32 for a step whose gradient tail ran phase A of the segment reduce only (amid_grad_tail_live_f32): the runs of
// the sorted list (seg_off / seg_of, n_sorted entries; workspace = the tail's) that cross chunk borders are summed from their partial rows
// by extra workgroups of this launch, written to uniq_grad and applied; everything else as amid_optimizer_step_f32.  D = 64 / 128 / 256.
extern "C" int amid_optimizer_step_spans_f32(float* p, float* m, float* v, const float* g, long long n, float* table, float* m_tab, float* v_tab,
                                             int* last, const int* uniq_ids, const int* n_uniq, int n_uniq_max, float* uniq_grad, int D,
                                             float grad_scale, const void* step_state, const int* seg_off, const int* seg_of, int n_sorted,
                                             const void* workspace, void* stream) {
    AMID_CHECK_ARG(p && m && v && g && n > 0 && table && m_tab && v_tab && last && uniq_ids && n_uniq && uniq_grad && step_state &&
                   n_uniq_max > 0 && seg_off && seg_of && n_sorted > 0 && workspace);
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    long long db = (n / 4 + 255) / 256;
    if (db < 1) db = 1;
    if (db > 1024) db = 1024;
    const int rb = rows_grid(n_uniq_max);
    const int nch = (n_sorted + SEG_CHUNK - 1) / SEG_CHUNK;
#define AMID_OPT_LAUNCH(VEC)                                                                                                              \
    optimizer_step_spans_kernel<VEC><<<nch + (int)db + rb, 256, 0, (hipStream_t)stream>>>(p, m, v, g, n, (int)db, table, m_tab, v_tab, last, uniq_ids, \
                                                                                          n_uniq, uniq_grad, (const StepState*)step_state, grad_scale, \
                                                                                          seg_off, seg_of, n_sorted, (const float*)workspace, nch);
    if (D == 64) { AMID_OPT_LAUNCH(1) } else if (D == 128) { AMID_OPT_LAUNCH(2) } else { AMID_OPT_LAUNCH(4) }
#undef AMID_OPT_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}


// amid_grad_tail_live_f32 (hidg = NULL) + amid_optimizer_step_spans_f32 as ONE launch (grad_tail_opt_kernel above).  ticket: one int32 of
// device memory, zero before the first call (the launch leaves it zero).  [left_lo, left_hi): the floats of the flat dense buffer whose
// gradients are already final in g when the launch starts (16-byte aligned bounds; left_lo = left_hi: none); every other dense gradient
// must be the dst of one of `entries` or a position row.  D = 64 / 128 / 256.
static int tail_opt_common(TailOptArgs& a, const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                           void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off, int total_blocks,
                           const int* live, int B, int T, float* dpos0, float* dpos1, float* g, long long n, long long left_lo, long long left_hi,
                           const int* uniq_ids, const int* n_uniq, int n_uniq_max, int* ticket) {
    AMID_CHECK_ARG(grad_rows && pos_sorted && seg_off && seg_of && workspace && uniq_grad && n_idx > 0 && entries_dev && n_entries > 0 &&
                   blk_off && total_blocks > 0);
    // live == NULL: no position rows are summed here (every dense gradient is an entry's: the steps outside the folded form)
    AMID_CHECK_ARG(live == nullptr || (B > 0 && T > 0 && dpos0 && dpos1));
    AMID_CHECK_ARG(g && n > 0 && uniq_ids && n_uniq && n_uniq_max > 0 && ticket);
    AMID_CHECK_ARG(left_lo >= 0 && left_lo <= left_hi && left_hi <= n && (left_lo & 3) == 0);
    AMID_CHECK_ARG(((((unsigned long long)dpos0) | ((unsigned long long)dpos1) | ((unsigned long long)grad_rows) | ((unsigned long long)g)) & 15) == 0);
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    a.grad_rows = grad_rows; a.pos_sorted = pos_sorted; a.seg_off = seg_off; a.seg_of = seg_of; a.n_sorted = n_idx;
    a.uniq_grad = uniq_grad; a.partial = (float*)workspace;
    a.nch = (n_idx + SEG_CHUNK - 1) / SEG_CHUNK; a.n_seg = (a.nch + 3) / 4; a.ticket = ticket;
    a.entries = (const ReduceEntry*)entries_dev; a.blk_off = blk_off; a.n_entries = n_entries; a.n_red = total_blocks;
    a.ps.rows = grad_rows; a.ps.live = live; a.ps.B = B; a.ps.T = T; a.ps.dst[0] = dpos0; a.ps.dst[1] = dpos1;
    a.ps.nblk = live != nullptr ? (T * D + 127) / 128 : 0;
    a.g = g; a.n_dense = n; a.left_lo = left_lo; a.left_hi = left_hi;
    a.left_blocks = (int)((left_hi - left_lo + 1023) / 1024);
    a.uniq_ids = uniq_ids; a.n_uniq = n_uniq; a.row_blocks = rows_grid(n_uniq_max);
    return AMID_OK;
}

extern "C" int amid_grad_tail_opt_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                      void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off,
                                      int total_blocks, const int* live, int B, int T, float* dpos0, float* dpos1, float* p, float* m, float* v,
                                      float* g, long long n, long long left_lo, long long left_hi, float* table, float* m_tab, float* v_tab,
                                      int* last, const int* uniq_ids, const int* n_uniq, int n_uniq_max, float grad_scale,
                                      const void* step_state, int* ticket, void* stream) {
    AMID_CHECK_ARG(p && m && v && table && m_tab && v_tab && last && step_state);
    AMID_CHECK_ARG(((((unsigned long long)p) | ((unsigned long long)m) | ((unsigned long long)v)) & 15) == 0);
    TailOptArgs a = {};
    if (int e = tail_opt_common(a, grad_rows, pos_sorted, seg_off, seg_of, n_idx, D, workspace, uniq_grad, entries_dev, n_entries, blk_off, total_blocks,
                                live, B, T, dpos0, dpos1, g, n, left_lo, left_hi, uniq_ids, n_uniq, n_uniq_max, ticket)) return e;
    a.p = p; a.m = m; a.v = v;
    a.table = table; a.m_tab = m_tab; a.v_tab = v_tab; a.last = last;
    a.st = (const StepState*)step_state; a.grad_scale = grad_scale;
    const int grid = a.n_seg + a.row_blocks + a.n_red + 2 * a.ps.nblk + a.left_blocks;
#define AMID_TO_LAUNCH(VEC) grad_tail_opt_kernel<VEC, true><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    if (D == 64) { AMID_TO_LAUNCH(1) } else if (D == 128) { AMID_TO_LAUNCH(2) } else { AMID_TO_LAUNCH(4) }
#undef AMID_TO_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// The data-parallel form, ONE launch (round 6; it was amid_grad_tail_live_f32's launch + a spans / packing launch): the same sums, shipped
// instead of applied -- uniq_grad (complete when the launch ends) points into this rank's exchange chunk or, n_out = 0, at the plan's own
// buffer; n_out > 0: out_ids [n_out] <- the unique ids padded with pad_id, dense_dst (optional) <- every dense gradient this launch finishes
// and the floats [left_lo, left_hi) of g (the caller keeps the slot padding of dense_dst zero), err_flag gets AMID_FLAG_UMAX_EXCEEDED when
// *n_uniq > n_out.  g = the flat dense gradient buffer the entries' dst point into (n floats).
extern "C" int amid_grad_tail_live_dp1_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                           void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off,
                                           int total_blocks, const int* live, int B, int T, float* dpos0, float* dpos1, float* g, long long n,
                                           long long left_lo, long long left_hi, const int* uniq_ids, const int* n_uniq, int n_uniq_max,
                                           int n_out, int pad_id, int* out_ids, float* dense_dst, int* err_flag, int* ticket, void* stream) {
    AMID_CHECK_ARG(n_out >= 0 && n_out <= n_idx && (n_out == 0 || out_ids) && (dense_dst == nullptr || n_out > 0) &&
                   (((unsigned long long)dense_dst) & 15) == 0);
    TailOptArgs a = {};
    if (int e = tail_opt_common(a, grad_rows, pos_sorted, seg_off, seg_of, n_idx, D, workspace, uniq_grad, entries_dev, n_entries, blk_off, total_blocks,
                                live, B, T, dpos0, dpos1, g, n, left_lo, left_hi, uniq_ids, n_uniq, n_uniq_max, ticket)) return e;
    a.dense_dst = dense_dst; a.out_ids = out_ids; a.n_out = n_out; a.pad_id = pad_id; a.id_blocks = (n_out + 255) / 256; a.err = err_flag;
    if (dense_dst == nullptr) a.left_blocks = 0;
    const int grid = a.n_seg + a.row_blocks + a.n_red + 2 * a.ps.nblk + a.left_blocks + a.id_blocks;
#define AMID_TO_LAUNCH(VEC) grad_tail_opt_kernel<VEC, false><<<grid, 256, 0, (hipStream_t)stream>>>(a);
    if (D == 64) { AMID_TO_LAUNCH(1) } else if (D == 128) { AMID_TO_LAUNCH(2) } else { AMID_TO_LAUNCH(4) }
#undef AMID_TO_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
