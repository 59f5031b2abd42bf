// K1: item-embedding gathers (reference: embItemLayerEnhance.forward, model_seq.py:27-29, called
// four times per forward at model_seq.py:418-421) fused with Log2feats' positional add, embedding
// dropout and the feature-level (==0) timeline mask (model_seq.py:361-366), plus the matching backward
// (dropout/mask on the incoming gradient and the pos_emb gradient).
//
// Layout: one concatenated index array idx_all[N_idx] = [seq_d1 (B*T) | seq_d2 (B*T) | items (B*NI)]
// with items[b] = (i_node[b], neg_samples[b,:]); the gathered rows land in one [N_idx, D] buffer in
// the same order, so the embedding backward can segment-reduce one contiguous gradient buffer.
//
// HBM-bound: per index 8 B (idx as delivered, int64 at the boundary) + D*4 read + D*4 written.
// A row (512 B at D=128) is moved by one half-wave as float4 per lane; each half-wave keeps ROWS_IN_FLIGHT
// independent rows in flight.  Measured on cfg5 S-uniform (417 792 random 512-B rows of a 5.12 GB table, MI355X):
// rows in flight x grid cap = 1 x 4096: 140 us, 4 x 4096: 129 us, 2 x 4096: 115 us, 2 x 16384: 110 us (4.0 TB/s of
// algorithmic read + write) -- many light waves beat few heavy ones here.
#include "common.h"
#include "live_list.h"
#include "rng.h"
#include "sort_phases.h"
#include "adam_replay.h"
#include "weights_image.h"

namespace amid {

constexpr int ROWS_IN_FLIGHT = 2;
constexpr int EMBED_RIF = 4;        // rows a half-wave of the fused embedding forward keeps in flight (its indices come 32 at a time)

// ---------------------------------------------------------------------------------------------
// plain gather: out[i,:] = table[idx[i],:]           (bit-exact; G1)
// ---------------------------------------------------------------------------------------------
template <typename IdxT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, long long n_rows, int D,
                                                          const IdxT* __restrict__ idx, long long n_idx,
                                                          float* __restrict__ out, int* __restrict__ err) {
    const int sub = threadIdx.x & 31;                              // lane within the half-wave
    const long long hw = (long long)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    const long long n_hw = (long long)gridDim.x * (blockDim.x >> 5);
    const int q = D >> 2;                                          // float4 per row
    for (long long r0 = hw * ROWS_IN_FLIGHT; r0 < n_idx; r0 += n_hw * ROWS_IN_FLIGHT) {
        long long src[ROWS_IN_FLIGHT];
#pragma unroll
        for (int u = 0; u < ROWS_IN_FLIGHT; ++u) {
            long long r = r0 + u;
            long long id = (r < n_idx) ? (long long)idx[r] : 0;
            if (id < 0 || id >= n_rows) { if (err && sub == 0 && r < n_idx) atomicOr(err, 1); id = 0; }
            src[u] = id;
        }
        for (int c = sub; c < q; c += 32) {
            float4 v[ROWS_IN_FLIGHT];
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u) v[u] = ld4(table + src[u] * D + 4 * c);
#pragma unroll
            for (int u = 0; u < ROWS_IN_FLIGHT; ++u)
                if (r0 + u < n_idx) st4(out + (r0 + u) * D + 4 * c, v[u]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// live list of a batch: live[0 .. n0) = the batch rows b with domain[b] == 0 (ascending), live[n0 .. B) = those with
// domain[b] != 0, live[B] = n0.  In the fused train step only the sequence (domain_id[b], b) of sample b is ever read by the loss
// (train_sr.py:205-211: the other domain's BCE terms are multiplied by zero), so encoder work is enumerated through this list.
// One workgroup; ballots + popcounts (a few microseconds even at B = 4096).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void live_list_kernel(const long long* __restrict__ domain, int B, int* __restrict__ live) {
    live_list_block(domain, B, live);
}

// ---------------------------------------------------------------------------------------------
// int64 -> int32 index packing at the module boundary (train_sr.py:191-199 hands LongTensors)
// ---------------------------------------------------------------------------------------------
__global__ void pack_indices_kernel(const long long* __restrict__ i_node, const long long* __restrict__ neg,
                                    const long long* __restrict__ seq_d1, const long long* __restrict__ seq_d2,
                                    int B, int T, int n_neg, long long n_rows, int* __restrict__ idx_all, int* __restrict__ err,
                                    StepState* __restrict__ bump, const long long* __restrict__ domain, int* __restrict__ live) {
    if (bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { bump->step += 1; bump->step_done = bump->step; }      // folded amid_step_begin (nobody in this launch reads it)
    if (live != nullptr && blockIdx.x == gridDim.x - 1) live_list_block(domain, B, live);        // folded amid_live_list_i32
    const int M = B * T, NI = 1 + n_neg;
    const int n = 2 * M + B * NI;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        long long v;
        if (i < M) v = seq_d1[i];
        else if (i < 2 * M) v = seq_d2[i - M];
        else {
            int j = i - 2 * M, b = j / NI, k = j % NI;
            v = (k == 0) ? i_node[b] : neg[(long long)b * n_neg + (k - 1)];
        }
        if (v < 0 || v >= n_rows) { atomicOr(err, 1); v = 0; }
        idx_all[i] = (int)v;
    }
}

// Pool-input variant: the batches of an epoch are resident in HBM as `n_pool` packed images ([i_node B][neg B n_neg][seq_d1 M]
// [seq_d2 M][domain, labels, ...]: the plan's input layout).  The kernel picks image (step + phase) % n_pool by the DEVICE step
// counter, so a replayed hipGraph walks the pool with no per-step host copy; it also mirrors the image into the plan's static
// input words (the head kernels read domain / labels there).  Every block reads the step before it takes a ticket and the
// block that takes the last ticket bumps it: no block can see the bumped value.
__global__ void pack_indices_pool_kernel(const long long* __restrict__ pool, long long stride, int n_pool, long long phase,
                                         long long* __restrict__ in_pack, int in_words, int B, int T, int n_neg, long long n_rows,
                                         int* __restrict__ idx_all, int* __restrict__ err, StepState* __restrict__ st, int* __restrict__ live) {
    const long long t_pre = __atomic_load_n(&st->step, __ATOMIC_RELAXED);
    long long which = (t_pre + phase) % n_pool;
    if (which < 0) which += n_pool;
    const long long* __restrict__ src = pool + which * stride;
    const int M = B * T, NI = 1 + n_neg;
    const int n_items = B * n_neg, n_index_words = B + n_items + 2 * M;
    // folded amid_live_list_i32: straight from the image's domain words (read-only input: no ordering against the mirroring below)
    if (live != nullptr && blockIdx.x == gridDim.x - 1) live_list_block(src + n_index_words, B, live);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < in_words; i += gridDim.x * blockDim.x) {
        long long v = src[i];
        in_pack[i] = v;
        if (i < n_index_words) {
            int dst;
            if (i < B) dst = 2 * M + i * NI;
            else if (i < B + n_items) { const int j = i - B; dst = 2 * M + (j / n_neg) * NI + 1 + (j % n_neg); }
            else dst = i - B - n_items;
            if (v < 0 || v >= n_rows) { atomicOr(err, 1); v = 0; }
            idx_all[dst] = (int)v;
        }
    }
    __syncthreads();
    // (no __threadfence: the only access that must precede the ticket is this block's READ of the step, and its value has been
    // consumed -- it addresses every load above; a fence per block writes the L2 back and was most of this launch's time)
    if (threadIdx.x == 0 && atomicInc(&st->ticket, gridDim.x - 1) == gridDim.x - 1) { st->step = t_pre + 1; st->step_done = t_pre + 1; }
}

// ---------------------------------------------------------------------------------------------
// fused SASRec embedding forward.
//   seq rows  r <  2M : x = E[idx] + P_g[t]; tm = (x == 0); x *= dropout; x = tm ? 0 : x
//   item rows r >= 2M : x = E[idx]
// tmq[r][c] (uint8, one per float4 column quad) keeps the 4 "== 0" bits for the later per-layer
// re-masking (model_seq.py:383) and for backward.  pos == nullptr (BERT4Rec: no positional table,
// no embedding dropout, no mask) degrades to the plain gather for every row.
// ---------------------------------------------------------------------------------------------
// FOLD: the lazy-Adam replay and the sort rider compiled in (amid_embed_fwd_replay_f32).  The plain build must not carry them: with both
// in one kernel the gather ran at 178 VGPRs / 6.4 KB of LDS instead of 109 / 0 and cfg 5's K1 went from 52.8 to 81.4 us.
// W16: the first workgroups write this step's bf16 fragment images of the encoder weights (the one-launch forward on bf16 pieces reads them;
// as a launch of its own the 72 planes cost 4.9 us of a 0.355 ms step)
// n tiles of D x D floats: tile i = src[i][r * ld[i] + c] (tr[i] = 0) or its transpose; the first n_fwd images go to dst, the others to dstT
// (SASRec: 24 weights + the same 24 transposed into a buffer of their own; BERT4Rec: 96 tiles of its 16 weights, bert_strip.hip)
template <int RIF, bool FOLD, bool W16 = false>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ table, const int* __restrict__ idx_all,
                                                        const float* __restrict__ pos0, const float* __restrict__ pos1,
                                                        int B, int T, int D, int n_item_rows,
                                                        float* __restrict__ xg, unsigned char* __restrict__ tmq,
                                                        const RngState* __restrict__ rng, int train, unsigned thr16, float scale,
                                                        const int* __restrict__ live, int* __restrict__ idx_c, int* __restrict__ row_c, int chunk,
                                                        const float* __restrict__ m_tab, const float* __restrict__ v_tab,
                                                        const int* __restrict__ last, const StepState* __restrict__ adam_st, const SortRider rd,
                                                        const W16Rider wr) {
    // rider: the first workgroups run phase 1 of the step's index sort (sort_phases.h) beside the gather (it rode in the catch-up
    // launch while that launch existed)
    int nrb = 0;
    if constexpr (W16) {
        nrb = wr.n * wr.per;
        if ((int)blockIdx.x < nrb) {
            const int wi = blockIdx.x / wr.per;
            unsigned short* out = wi < wr.n_fwd ? wr.dst + (size_t)wi * wr.planes * wr.D * wr.D
                                                : wr.dstT + (size_t)(wi - wr.n_fwd) * wr.planes * wr.D * wr.D;
            weights_image_block(wr.src[wi], out, wr.D, wr.tr[wi], wr.planes, blockIdx.x - wi * wr.per, wr.per, wr.ld[wi]);
            return;
        }
    }
    if constexpr (FOLD) {
        const int nsb = rider_blocks(rd);
        __shared__ __attribute__((aligned(16))) int sort_hist[OS_BINS_MAX];
        if ((int)blockIdx.x - nrb < nsb) { sort_phase_ct<1>(rd.plan, blockIdx.x - nrb, sort_hist); return; }
        nrb += nsb;
    }
    const int bid = blockIdx.x - nrb;
    // the step's counters re-joined (rng.h StepState::step_done): behind a one-launch step head `step` is one ahead of `step_done`; no
    // block of this launch reads step_done, so one thread may write it
    if (rng != nullptr && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) const_cast<RngState*>(rng)->step_done = rng->step;
    const int sub = threadIdx.x & 31;
    const int hw = bid * (blockDim.x >> 5) + (threadIdx.x >> 5);
    const int n_hw = (gridDim.x - nrb) * (blockDim.x >> 5);
    const int q = D >> 2;
    const int M = B * T;
    const int n_idx = 2 * M + n_item_rows;
    // live != nullptr: only the sequences of the live list are gathered (B of the 2 B): the walk covers B * T "virtual" sequence
    // rows + the item rows; everything else (index layout, dropout counters, output rows) is unchanged
    const int n_walk = live != nullptr ? M + n_item_rows : n_idx;
    const int n0 = live != nullptr ? live[B] : 0;
    unsigned long long seed = 0;
    unsigned step = 0;
    if (train) { seed = rng->seed; step = (unsigned)rng->step; }
    // last != nullptr: the lazy-Adam catch-up folded into the gather.  A row whose stamp lags (0 < last < t - 1) owes zero-gradient Adam
    // steps last + 1 .. t - 1 (adam.hip): they are replayed HERE, in registers, for the value the forward reads -- nothing is written;
    // the optimizer launch of this step, which visits exactly the rows that were gathered, replays the same steps again (the same bits:
    // adam_replay.h) in front of the real step and stamps the row.  No launch of its own, no ordering between workgroups.
    StepState stv = {};
    long long t_now = 0;
    bool any_lag = false;
    [[maybe_unused]] IdleCoef* ctab = nullptr;
    if constexpr (FOLD) {
        __shared__ IdleCoef ctab_s[COEF_TAB];
        ctab = ctab_s;
        stv = *adam_st;
        t_now = stv.step;
        bool mine_lag = false;                          // first sweep: does any position of this block lag at all? (usually not)
        for (int c0 = hw * chunk; c0 < n_walk; c0 += n_hw * chunk) {
            const int mine = c0 + sub;
            if (sub < chunk && mine < n_walk) {
                int r = mine;
                if (live != nullptr) {
                    if (r < M) { const int sq = r / T; r = (sq >= n0 ? M : 0) + live[sq] * T + (r - sq * T); }
                    else r += M;
                }
                const int l = last[idx_all[r]];
                if (l > 0 && l < t_now - 1) mine_lag = true;
            }
        }
        any_lag = __syncthreads_or(mine_lag ? 1 : 0) != 0;
        if (any_lag) fill_coef_table(ctab, stv);
    }
    // A half-wave owns a chunk of `chunk` (4 .. 32) walk positions: lane j < chunk resolves position j's (row, id) -- the dependent
    // chain live -> index -> table row is paid once per chunk, in parallel across the lanes, not once per row --, then the rows are
    // moved RIF at a time with every lane on its float4 column.  Long lists take chunks of 32, short ones small chunks (more waves).
    const int half = threadIdx.x & 32;
    for (int c0 = hw * chunk; c0 < n_walk; c0 += n_hw * chunk) {
        const int mine = c0 + sub;
        int my_row = 0, my_src = 0, my_last = 0;
        if (sub < chunk && mine < n_walk) {
            int r = mine;
            if (live != nullptr) {
                if (r < M) { const int sq = r / T; r = (sq >= n0 ? M : 0) + live[sq] * T + (r - sq * T); }
                else r += M;
            }
            my_row = r;
            my_src = idx_all[r];
            if constexpr (FOLD) { if (any_lag) my_last = last[my_src]; }
            // the walk over the live sequences is the step's compact index list: the id at every walk position and the row of the
            // full layout its gradient will stand in (what the sort, the segment reduce and the row Adam of the step then run on)
            if (idx_c != nullptr) { idx_c[mine] = my_src; row_c[mine] = r; }
        }
        // p = 0.5 at D = 128: ONE Philox call decides a whole row (rng.h).  Lane j draws the call of ITS row; the row loop below
        // fetches the word a lane's column quad needs from the row's lane -- one call per row instead of one per lane and row.
        const bool row_calls = train && D == 128 && spec_bits(thr16) == 1 && pos0 != nullptr;
        uint4 my_bits = make_uint4(0u, 0u, 0u, 0u);
        if (row_calls && sub < chunk && mine < n_walk && my_row < 2 * M) {
            const int g = my_row >= M;
            my_bits = rng_call(seed, (unsigned long long)(my_row - g * M), site_id(g, 0, SITE_EMB), step);
        }
        const int n_here = min(chunk, n_walk - c0);
        for (int u0 = 0; u0 < n_here; u0 += RIF) {
            long long src[RIF];
            int row[RIF];
            [[maybe_unused]] int lst[RIF];
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                src[u] = __shfl(my_src, half + ((u0 + u) & 31), 64);
                row[u] = __shfl(my_row, half + ((u0 + u) & 31), 64);
                if constexpr (FOLD) lst[u] = any_lag ? __shfl(my_last, half + ((u0 + u) & 31), 64) : 0;
            }
            unsigned kbits[RIF];                       // the 32-bit word of row u's call that holds this lane's column quad (c = sub)
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                kbits[u] = 0u;
                if (row_calls) {
                    const int from = half + ((u0 + u) & 31);
                    const unsigned wx = __shfl(my_bits.x, from, 64), wy = __shfl(my_bits.y, from, 64);
                    const unsigned wz = __shfl(my_bits.z, from, 64), ww = __shfl(my_bits.w, from, 64);
                    const int wsel = sub >> 3;
                    kbits[u] = wsel == 0 ? wx : wsel == 1 ? wy : wsel == 2 ? wz : ww;
                }
            }
            for (int c = sub; c < q; c += 32) {
                float4 v[RIF];
#pragma unroll
                for (int u = 0; u < RIF; ++u) {
                    // non-temporal: a table row is read once per step and the table is far larger than the caches -- inside the cfg 5 step,
                    // behind the catch-up's footprint, 59.1 -> 55.2 us (profiles/tools/probe/k1_instep.py)
                    const f32x4 t = __builtin_nontemporal_load((const f32x4*)(table + src[u] * D + 4 * c));
                    v[u] = make_float4(t[0], t[1], t[2], t[3]);
                }
                if constexpr (FOLD) if (any_lag) {
#pragma unroll
                    for (int u = 0; u < RIF; ++u) {
                        if (u0 + u < n_here && lst[u] > 0 && lst[u] < t_now - 1) {      // (uniform over the half-wave: it shares the row)
                            float4 mq = ld4(m_tab + src[u] * D + 4 * c), vq = ld4(v_tab + src[u] * D + 4 * c);
                            replay_quad(v[u], mq, vq, (long long)lst[u] + 1, t_now - 1, stv, ctab);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < RIF; ++u) {
                    const int r = row[u];
                    if (u0 + u >= n_here) continue;
                    float4 x = v[u];
                    if (r < 2 * M && pos0 != nullptr) {
                        const int g = r >= M;
                        const int local = r - g * M;
                        const int t = local % T;
                        const float4 p = ld4((g ? pos1 : pos0) + (long long)t * D + 4 * c);
                        x = f4add(x, p);
                        const unsigned bits = (x.x == 0.f ? 1u : 0u) | (x.y == 0.f ? 2u : 0u) | (x.z == 0.f ? 4u : 0u) | (x.w == 0.f ? 8u : 0u);
                        if (row_calls) {
                            const unsigned kw = (kbits[u] >> ((c & 7) * 4)) | (spec_thr(thr16) == 0 ? 0xFu : 0u);      // keep bits of this quad
                            x = make_float4((kw & 1u) ? x.x * scale : 0.f, (kw & 2u) ? x.y * scale : 0.f, (kw & 4u) ? x.z * scale : 0.f,
                                            (kw & 8u) ? x.w * scale : 0.f);
                        } else if (train) {
                            const float4 m = dropout_mult4(seed, site_id(g, 0, SITE_EMB), step, (unsigned long long)local * D + 4 * c, thr16, scale);
                            x = f4mul(x, m);
                        }
                        if (bits) {
                            if (bits & 1u) x.x = 0.f;
                            if (bits & 2u) x.y = 0.f;
                            if (bits & 4u) x.z = 0.f;
                            if (bits & 8u) x.w = 0.f;
                        }
                        tmq[(long long)r * q + c] = (unsigned char)bits;
                    }
                    st4(xg + (long long)r * D + 4 * c, x);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// embedding backward, seq rows only: block (t, g) walks b = 0..B-1
//   dxe[r] = dx0[r] * ~tm * dropout          (in place: the buffer then feeds the segment reduce)
//   dP_g[t] = sum_b dxe[g,b,t]               (fixed-order tree => bitwise reproducible)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_bwd_kernel(float* __restrict__ dxg, const unsigned char* __restrict__ tmq,
                                                        int B, int T, int D, int nsplit, float* __restrict__ dpos_part,
                                                        const RngState* __restrict__ rng, int train, unsigned thr16, float scale,
                                                        const long long* __restrict__ row_domain, const SortRider rd) {
    extern __shared__ __attribute__((aligned(16))) float red[];      // [8][D]
    // rider: with rd.phase set the launch has one more z-slice, whose first rd.plan.nblk workgroups run the last phase of the step's
    // index sort (sort_phases.h: run heads)
    if (rd.phase != 0 && (int)blockIdx.z == nsplit) {
        const int rb = blockIdx.y * gridDim.x + blockIdx.x;
        if (rb < rd.plan.nblk) sort_phase_ct<5>(rd.plan, rb, red);
        return;
    }
    const int t = blockIdx.x, g = blockIdx.y, z = blockIdx.z;
    const int sub = threadIdx.x & 31, rg = threadIdx.x >> 5;        // 8 row groups
    const int q = D >> 2;
    const int M = B * T;
    const int per = (B + nsplit - 1) / nsplit;
    const int b_beg = z * per, b_end = min(B, b_beg + per);
    unsigned long long seed = 0;
    unsigned step = 0;
    if (train) { seed = rng->seed; step = (unsigned)rng->step; }
    if (D == 128 && (!train || spec_bits(thr16) == 1)) {
        // one float4 column per lane, the row group's rows four at a time: every load of a batch is issued before the first use, and
        // the batch's dropout words come from ONE Philox call per row (p = 0.5: a call decides a whole row, rng.h), drawn by lane i
        // for row i and handed round -- not one call per lane and row
        const int c = sub, half = threadIdx.x & 32;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b0 = b_beg + rg; b0 < b_end; b0 += 32) {
            float4 v[4];
            unsigned bits[4], kw[4];
            bool in[4], livef[4];
            uint4 mine = make_uint4(~0u, ~0u, ~0u, ~0u);
            if (train && sub < 4 && b0 + 8 * sub < b_end)
                mine = rng_call(seed, (unsigned long long)((b0 + 8 * sub) * T + t), site_id(g, 0, SITE_EMB), step);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + 8 * i;
                in[i] = b < b_end;
                livef[i] = in[i] && !(row_domain != nullptr && (row_domain[b] != 0 ? 1 : 0) != g);
                const long long r = (long long)g * M + (long long)(in[i] ? b : b_beg) * T + t;
                v[i] = livef[i] ? ld4(dxg + r * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                bits[i] = livef[i] ? tmq[r * q + c] : 0u;
                const unsigned wx = __shfl(mine.x, half + i, 64), wy = __shfl(mine.y, half + i, 64);
                const unsigned wz = __shfl(mine.z, half + i, 64), ww = __shfl(mine.w, half + i, 64);
                const int wsel = c >> 3;
                kw[i] = ((wsel == 0 ? wx : wsel == 1 ? wy : wsel == 2 ? wz : ww) >> ((c & 7) * 4)) | (spec_thr(thr16) == 0 ? 0xFu : 0u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!in[i]) continue;
                const long long r = (long long)g * M + (long long)(b0 + 8 * i) * T + t;
                float4 x = v[i];
                if (train) x = make_float4((kw[i] & 1u) ? x.x * scale : 0.f, (kw[i] & 2u) ? x.y * scale : 0.f, (kw[i] & 4u) ? x.z * scale : 0.f,
                                           (kw[i] & 8u) ? x.w * scale : 0.f);
                if (bits[i] & 1u) x.x = 0.f;
                if (bits[i] & 2u) x.y = 0.f;
                if (bits[i] & 4u) x.z = 0.f;
                if (bits[i] & 8u) x.w = 0.f;
                if (!livef[i]) x = make_float4(0.f, 0.f, 0.f, 0.f);    // no gradient reached this sequence: the zeros it stands for
                st4(dxg + r * D + 4 * c, x);
                acc = f4add(acc, x);
            }
        }
        st4(red + rg * D + 4 * c, acc);
    } else
    for (int c = sub; c < q; c += 32) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b = b_beg + rg; b < b_end; b += 8) {
            const int local = b * T + t;
            const long long r = (long long)g * M + local;
            if (row_domain != nullptr && (row_domain[b] != 0 ? 1 : 0) != g) {      // no gradient reached this sequence: the buffer holds
                st4(dxg + r * D + 4 * c, make_float4(0.f, 0.f, 0.f, 0.f));          // nothing for it (the row-tile kernels skipped it)
                continue;
            }
            float4 v = ld4(dxg + r * D + 4 * c);
            if (train) v = f4mul(v, dropout_mult4(seed, site_id(g, 0, SITE_EMB), step, (unsigned long long)local * D + 4 * c, thr16, scale));
            const unsigned bits = tmq[r * q + c];
            if (bits) {
                if (bits & 1u) v.x = 0.f;
                if (bits & 2u) v.y = 0.f;
                if (bits & 4u) v.z = 0.f;
                if (bits & 8u) v.w = 0.f;
            }
            st4(dxg + r * D + 4 * c, v);
            acc = f4add(acc, v);
        }
        st4(red + rg * D + 4 * c, acc);
    }
    __syncthreads();
    float* dp = dpos_part + (((long long)z * 2 + g) * T + t) * D;     // [nsplit][2][T][D]
    for (int e = threadIdx.x; e < D; e += blockDim.x) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k * D + e];
        dp[e] = s;
    }
}

// key mask of BERT4Rec: keep[b][t] = seq[b][t] > 0 (reference model_seq.py:288)
__global__ __launch_bounds__(256) void key_keep_kernel(const long long* __restrict__ seq, long long n, unsigned char* __restrict__ keep) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) keep[i] = seq[i] > 0 ? 1 : 0;
}

// the same mask tiled `reps` times along the key axis: keep[b][r * T + t] = seq[b][t] > 0 (model_seq.py:286, :294 `.repeat(1, n, 2)`)
__global__ __launch_bounds__(256) void key_keep_tiled_kernel(const long long* __restrict__ seq, int B, int T, int reps,
                                                             unsigned char* __restrict__ keep) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)B * T * reps;
    if (i >= n) return;
    const int b = (int)(i / ((long long)T * reps));
    const int t = (int)(i % T);
    keep[i] = seq[(long long)b * T + t] > 0 ? 1 : 0;
}

}  // namespace amid

using namespace amid;

// embed_fwd_kernel: a half-wave per chunk of rows; the largest chunk that still fills every wave slot of the chip (8 blocks per CU:
// the gather wants as many rows in flight as the chip can hold -- measured at cfg 5, 221 k rows: chunks of 32 / 864 blocks 78.8 us)
static inline int embed_chunk(long long n_rows_to_move) {
    int chunk = 32;
    while (chunk > EMBED_RIF && (n_rows_to_move / chunk + 7) / 8 < 2048) chunk >>= 1;
    return chunk;
}
static inline int embed_grid(long long n_rows_to_move, int chunk) {
    long long blocks = ((n_rows_to_move + chunk - 1) / chunk + 7) / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > 16384) blocks = 16384;      // (caps of 1024 / 2048 / 4096 blocks with chunks of 4 or 8 measured the same: profiles/tools/probe/k1_ab.py)
    return (int)blocks;
}

static inline int gather_grid(long long n_rows_to_move) {
    long long hw_needed = (n_rows_to_move + ROWS_IN_FLIGHT - 1) / ROWS_IN_FLIGHT;
    long long blocks = (hw_needed + 7) / 8;
    if (blocks < 1) blocks = 1;
    if (blocks > 16384) blocks = 16384;   // 64 blocks per CU, grid-stride beyond
    return (int)blocks;
}

extern "C" int amid_gather_rows_f32(const float* table, long long n_rows, int D, const void* idx, int idx_is_i64,
                                    long long n_idx, float* out, int* err_flag, void* stream) {
    AMID_CHECK_ARG(D > 0 && (D % 4) == 0 && n_idx >= 0);
    if (n_idx == 0) return AMID_OK;                 // empty lookup: nothing to move, pointers may be null
    AMID_CHECK_ARG(table && idx && out);
    hipStream_t s = (hipStream_t)stream;
    if (idx_is_i64)
        gather_rows_kernel<long long><<<gather_grid(n_idx), 256, 0, s>>>(table, n_rows, D, (const long long*)idx, n_idx, out, err_flag);
    else
        gather_rows_kernel<int><<<gather_grid(n_idx), 256, 0, s>>>(table, n_rows, D, (const int*)idx, n_idx, out, err_flag);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int pack_indices(const long long* i_node, const long long* neg, const long long* seq_d1, const long long* seq_d2,
                        int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag, void* step_state_to_bump,
                        const long long* domain, int* live, void* stream) {
    AMID_CHECK_ARG(i_node && neg && seq_d1 && seq_d2 && idx_all && err_flag && B > 0 && T > 0 && n_neg >= 0);
    int n = 2 * B * T + B * (1 + n_neg);
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    pack_indices_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(i_node, neg, seq_d1, seq_d2, B, T, n_neg, n_rows, idx_all, err_flag,
                                                                  (StepState*)step_state_to_bump, domain, live);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_pack_indices(const long long* i_node, const long long* neg, const long long* seq_d1, const long long* seq_d2,
                                 int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag, void* step_state_to_bump,
                                 void* stream) {
    return pack_indices(i_node, neg, seq_d1, seq_d2, B, T, n_neg, n_rows, idx_all, err_flag, step_state_to_bump, nullptr, nullptr, stream);
}

// the same + amid_live_list_i32(domain, B, live) in the same launch
extern "C" int amid_pack_indices_live(const long long* i_node, const long long* neg, const long long* seq_d1, const long long* seq_d2,
                                      int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag, void* step_state_to_bump,
                                      const long long* domain, int* live, void* stream) {
    AMID_CHECK_ARG(domain && live);
    return pack_indices(i_node, neg, seq_d1, seq_d2, B, T, n_neg, n_rows, idx_all, err_flag, step_state_to_bump, domain, live, stream);
}

static int pack_indices_pool(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack,
                             int in_words, int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag,
                             void* step_state, int* live, void* stream) {
    AMID_CHECK_ARG(pool && in_pack && idx_all && err_flag && step_state && n_pool > 0 && B > 0 && T > 0 && n_neg >= 0);
    AMID_CHECK_ARG(in_words >= B + B * n_neg + 2 * B * T + (live ? B : 0) && pool_stride >= in_words);
    int blocks = (in_words + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    pack_indices_pool_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(pool, pool_stride, n_pool, phase, in_pack, in_words, B, T, n_neg,
                                                                       n_rows, idx_all, err_flag, (StepState*)step_state, live);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_pack_indices_pool(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack,
                                      int in_words, int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag,
                                      void* step_state, void* stream) {
    return pack_indices_pool(pool, pool_stride, n_pool, phase, in_pack, in_words, B, T, n_neg, n_rows, idx_all, err_flag, step_state, nullptr, stream);
}

// the same + amid_live_list_i32 on the image's domain words (the B words behind the index words) in the same launch
extern "C" int amid_pack_indices_pool_live(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack,
                                           int in_words, int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag,
                                           void* step_state, int* live, void* stream) {
    AMID_CHECK_ARG(live != nullptr);
    return pack_indices_pool(pool, pool_stride, n_pool, phase, in_pack, in_words, B, T, n_neg, n_rows, idx_all, err_flag, step_state, live, stream);
}

static int embed_fwd(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                     int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop, const int* live,
                     int* idx_c, int* row_c, void* stream, const float* m_tab = nullptr, const float* v_tab = nullptr, const int* last = nullptr,
                     const void* adam_state = nullptr, const void* sort_plan = nullptr, int sort_phase = 0, const float* const* w_src = nullptr,
                     int n_w = 0, int w_planes = 0, void* w16_dst = nullptr, void* w16t_dst = nullptr, const int* w_ld = nullptr,
                     const int* w_tr = nullptr) {
    AMID_CHECK_ARG(last == nullptr || (m_tab && v_tab && adam_state));
    W16Rider wr = {};
    if (w_src != nullptr) {
        AMID_CHECK_ARG(n_w > 0 && (w_planes == 1 || w_planes == 3) && w16_dst && D == 128 && last == nullptr && sort_plan == nullptr);
        AMID_CHECK_ARG((w_ld == nullptr) == (w_tr == nullptr) && n_w * (w_ld == nullptr && w16t_dst ? 2 : 1) <= W16_MAX && (w_ld == nullptr || !w16t_dst));
        for (int i = 0; i < n_w; ++i) {
            AMID_CHECK_ARG(w_src[i] && (w_ld == nullptr || (w_ld[i] >= D && w_ld[i] < 65536)));
            wr.src[i] = w_src[i]; wr.ld[i] = (unsigned short)(w_ld ? w_ld[i] : D); wr.tr[i] = (unsigned char)(w_tr && w_tr[i] ? 1 : 0);
        }
        wr.n = wr.n_fwd = n_w;
        if (w16t_dst != nullptr) {                     // the same matrices transposed, into their own buffer
            for (int i = 0; i < n_w; ++i) { wr.src[n_w + i] = w_src[i]; wr.ld[n_w + i] = (unsigned short)D; wr.tr[n_w + i] = 1; }
            wr.n = 2 * n_w;
        }
        wr.dst = (unsigned short*)w16_dst; wr.dstT = (unsigned short*)w16t_dst; wr.planes = w_planes; wr.D = D; wr.per = (D * (D / 8) + 255) / 256;
    }
    SortRider rd;
    rd.phase = 0;
    if (sort_plan != nullptr) {
        if (sort_phase != 1) return AMID_ERR_UNSUPPORTED;          // this launch carries phase 1
        rd.plan = *(const SortPlan*)sort_plan;
        rd.phase = sort_phase;
    }
    AMID_CHECK_ARG((idx_c == nullptr) == (row_c == nullptr) && (idx_c == nullptr || live != nullptr));
    AMID_CHECK_ARG(table && idx_all && xg && B > 0 && T > 0 && D > 0 && (D % 4) == 0 && n_item_rows >= 0);
    AMID_CHECK_ARG((pos0 == nullptr) == (pos1 == nullptr));
    AMID_CHECK_ARG(pos0 == nullptr || tmq != nullptr);
    AMID_CHECK_ARG(!train || rng_state != nullptr);
    const long long n_walk = (live != nullptr ? 1LL : 2LL) * B * T + n_item_rows;
    const int tr = (train && pos0 != nullptr && p_drop > 0.f) ? 1 : 0;
    const int chunk = embed_chunk(n_walk);
    if (last != nullptr || rd.phase != 0)
        embed_fwd_kernel<EMBED_RIF, true><<<embed_grid(n_walk, chunk) + rider_blocks_host(rd), 256, 0, (hipStream_t)stream>>>(
            table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, (const RngState*)rng_state, tr, keep_thr16(p_drop),
            tr ? 1.0f / (1.0f - p_drop) : 1.0f, live, idx_c, row_c, chunk, m_tab, v_tab, last, (const StepState*)adam_state, rd, wr);
    else if (wr.n > 0)
        embed_fwd_kernel<EMBED_RIF, false, true><<<embed_grid(n_walk, chunk) + wr.n * wr.per, 256, 0, (hipStream_t)stream>>>(
            table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, (const RngState*)rng_state, tr, keep_thr16(p_drop),
            tr ? 1.0f / (1.0f - p_drop) : 1.0f, live, idx_c, row_c, chunk, m_tab, v_tab, last, (const StepState*)adam_state, rd, wr);
    else
        embed_fwd_kernel<EMBED_RIF, false><<<embed_grid(n_walk, chunk), 256, 0, (hipStream_t)stream>>>(
            table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, (const RngState*)rng_state, tr, keep_thr16(p_drop),
            tr ? 1.0f / (1.0f - p_drop) : 1.0f, live, idx_c, row_c, chunk, m_tab, v_tab, last, (const StepState*)adam_state, rd, wr);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_embed_fwd_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                                  int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop,
                                  void* stream) {
    return embed_fwd(table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, rng_state, train, p_drop, nullptr, nullptr, nullptr, stream);
}

// K1 over the live sequences only (live: amid_live_list_i32): the rows of the other B sequences of xg / tmq are left untouched
extern "C" int amid_embed_fwd_live_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                                       int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop,
                                       const int* live, void* stream) {
    AMID_CHECK_ARG(live != nullptr);
    return embed_fwd(table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, rng_state, train, p_drop, live, nullptr, nullptr, stream);
}

// the same, also writing the step's compact index list (B T + n_item_rows entries): idx_c[i] = the id at position i of the walk over
// the live sequences and the items, row_c[i] = that position's row in the full [2 B T + items] layout
extern "C" int amid_embed_fwd_live_compact_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T,
                                               int D, int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train,
                                               float p_drop, const int* live, int* idx_c, int* row_c, void* stream) {
    AMID_CHECK_ARG(live != nullptr && idx_c != nullptr && row_c != nullptr);
    return embed_fwd(table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, rng_state, train, p_drop, live, idx_c, row_c, stream);
}

// K1 with the lazy-Adam catch-up folded in (and, optionally, phase 1 of the step's index sort riding as extra workgroups): every
// gathered row whose stamp last[id] lags behind step t - 1 (t = adam_state's step) has its pending zero-gradient Adam steps replayed in
// registers for the value written to xg; table, m, v and last are only READ (amid_optimizer_step_f32 replays the same steps in front
// of the real step).  live / idx_c / row_c: as the live and compact entry points (NULL: every sequence / no compact list).
extern "C" int amid_embed_fwd_replay_f32(const float* table, const float* m_tab, const float* v_tab, const int* last, const int* idx_all,
                                         const float* pos0, const float* pos1, int B, int T, int D, int n_item_rows, float* xg,
                                         unsigned char* tmq, const void* rng_state, int train, float p_drop, const int* live, int* idx_c,
                                         int* row_c, const void* adam_state, const void* sort_plan, int sort_phase, void* stream) {
    AMID_CHECK_ARG(m_tab && v_tab && last && adam_state);
    return embed_fwd(table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, rng_state, train, p_drop, live, idx_c, row_c, stream, m_tab,
                     v_tab, last, adam_state, sort_plan, sort_phase);
}

// K1 (live / idx_c / row_c as the three entry points above: NULL = every sequence / no compact list) with this step's bf16 fragment
// images of n_w square [D][D] weights written by extra workgroups of the same launch: w16_dst [n_w][planes][D][D] bf16, planes = 1 or 3
// (amid_sas_weights_bf16_planes is the launch this saves); w16t_dst != NULL: the images of the weights' TRANSPOSES too, same layout (the
// backward strips' operands).  D = 128.
extern "C" int amid_embed_fwd_w16_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                                      int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop,
                                      const int* live, int* idx_c, int* row_c, const float* const* w_src, int n_w, int w_planes, void* w16_dst,
                                      void* w16t_dst, void* stream) {
    AMID_CHECK_ARG(w_src != nullptr);
    return embed_fwd(table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, rng_state, train, p_drop, live, idx_c, row_c, stream, nullptr,
                     nullptr, nullptr, nullptr, nullptr, 0, w_src, n_w, w_planes, w16_dst, w16t_dst);
}

// ... with the riders' tiles spelled out: tile i = w_src[i][r * w_ld[i] + c] or (w_tr[i] != 0) its transpose, r, c < D = 128, n_w <= 96 tiles,
// images to w16_dst [n_w][planes][D][D] bf16 (amid_bert_weight_images_f32 is the launch this saves: BERT4Rec's strips on bf16 pieces)
extern "C" int amid_embed_fwd_tiles_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                                        int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop,
                                        const int* live, const float* const* w_src, const int* w_ld, const int* w_tr, int n_w, int w_planes,
                                        void* w16_dst, void* stream) {
    AMID_CHECK_ARG(w_src != nullptr && w_ld != nullptr && w_tr != nullptr);
    return embed_fwd(table, idx_all, pos0, pos1, B, T, D, n_item_rows, xg, tmq, rng_state, train, p_drop, live, nullptr, nullptr, stream, nullptr,
                     nullptr, nullptr, nullptr, nullptr, 0, w_src, n_w, w_planes, w16_dst, nullptr, w_ld, w_tr);
}

extern "C" int amid_live_list_i32(const long long* domain, int B, int* live, void* stream) {
    AMID_CHECK_ARG(domain && live && B > 0);
    live_list_kernel<<<1, 256, 0, (hipStream_t)stream>>>(domain, B, live);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int embed_bwd(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part, const void* rng_state,
                     int train, float p_drop, const long long* row_domain, const void* sort_plan, int sort_phase, void* stream) {
    AMID_CHECK_ARG(dxg && tmq && dpos_part && B > 0 && T > 0 && D > 0 && (D % 4) == 0 && nsplit > 0 && nsplit <= B);
    AMID_CHECK_ARG(!train || rng_state != nullptr);
    const int tr = (train && p_drop > 0.f) ? 1 : 0;
    SortRider rd;
    rd.phase = 0;
    if (sort_plan != nullptr) {
        if (sort_phase != 5) return AMID_ERR_UNSUPPORTED;                  // this launch carries phase 5 (run heads)
        rd.plan = *(const SortPlan*)sort_plan;
        rd.phase = 5;
        if (rd.plan.nblk > 2 * T) return AMID_ERR_UNSUPPORTED;             // the riders sit in one extra z-slice of T x 2 workgroups
    }
    embed_bwd_kernel<<<dim3(T, 2, nsplit + (rd.phase ? 1 : 0)), 256, 8 * D * sizeof(float), (hipStream_t)stream>>>(dxg, tmq, B, T, D, nsplit, dpos_part,
                                                                                               (const RngState*)rng_state, tr,
                                                                                               keep_thr16(p_drop),
                                                                                               tr ? 1.0f / (1.0f - p_drop) : 1.0f, row_domain, rd);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_embed_bwd_f32(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part,
                                  const void* rng_state, int train, float p_drop, void* stream) {
    return embed_bwd(dxg, tmq, B, T, D, nsplit, dpos_part, rng_state, train, p_drop, nullptr, nullptr, 0, stream);
}

// behind the *_rows backward kernels (sasrec_bwd.hip): the rows of the sequences (g, b) with (row_domain[b] != 0) != g were never
// written -- they are set to the zeros they stand for here, without being read
extern "C" int amid_embed_bwd_rows_f32(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part,
                                       const void* rng_state, int train, float p_drop, const long long* row_domain, void* stream) {
    AMID_CHECK_ARG(row_domain != nullptr);
    return embed_bwd(dxg, tmq, B, T, D, nsplit, dpos_part, rng_state, train, p_drop, row_domain, nullptr, 0, stream);
}

// either of the two (row_domain optional) carrying phase 5 of a sort plan (amid_sort_plan_pack: the run heads) as extra workgroups
extern "C" int amid_embed_bwd_sort_f32(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part,
                                       const void* rng_state, int train, float p_drop, const long long* row_domain, const void* sort_plan,
                                       int sort_phase, void* stream) {
    AMID_CHECK_ARG(sort_plan != nullptr);
    return embed_bwd(dxg, tmq, B, T, D, nsplit, dpos_part, rng_state, train, p_drop, row_domain, sort_plan, sort_phase, stream);
}

extern "C" int amid_key_keep_tiled_u8(const long long* seq, int B, int T, int reps, unsigned char* keep, void* stream) {
    AMID_CHECK_ARG(seq && keep && B > 0 && T > 0 && reps > 0);
    const long long n = (long long)B * T * reps;
    key_keep_tiled_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(seq, B, T, reps, keep);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_key_keep_u8(const long long* seq, long long n, unsigned char* keep, void* stream) {
    AMID_CHECK_ARG(seq && keep && n > 0);
    key_keep_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(seq, n, keep);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
