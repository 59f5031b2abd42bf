// K3: embedding gradient as a segment reduce over the sorted inverted index (sortuniq.hip):
//   uniq_grad[u,:] = sum over e in [seg_off[u], seg_off[u+1]) of grad_rows[pos_sorted[e], :]
// Replaces the reference's four dense EmbeddingBackward index_adds into zero-filled [n_rows, D]
// gradients (autograd of model_seq.py:418-421; 24 % of the reference step is the zero fill alone,
// SURVEY.md section 3).  No atomics: a row's addends are summed in sorted (position) order, so the
// result is bitwise reproducible.
//
// HBM-bound: reads N*(D*4 + 4) B, writes U*D*4 B.  Skew (one pad row owns 80-90 % of the
// positions) is handled by cutting the sorted list into fixed 64-entry chunks, one wave each:
// a run that lies inside one chunk is finished there; a run that crosses chunk borders leaves one
// partial per chunk, and the chunk in which the run starts adds the partials up in chunk order.
#include "common.h"
#include "reduce_partials.h"
#include "seg_spans.h"
#include "tail_parts.h"
#include "scorer_sum.h"

namespace amid {

template <int VEC>
__global__ __launch_bounds__(256) void segreduce_chunks_kernel(const float* __restrict__ grad_rows, const int* __restrict__ pos_sorted,
                                                               const int* __restrict__ seg_of, int n, float* __restrict__ uniq_grad,
                                                               float* __restrict__ partial, int chunk) {
    segreduce_chunks_block<VEC>(grad_rows, pos_sorted, seg_of, n, uniq_grad, partial, blockIdx.x, chunk);
}

// The two independent, bandwidth-bound ends of backward in ONE launch: blocks [0, n_seg) run phase A of the segment reduce, blocks
// [n_seg, n_seg + red_bx * n_entries) run the fixed-order reduction of the partial buffers (dense gradients + loss).  (As two
// kernels on two streams inside the step's graph they did run side by side, but the fork and the join each cost ~10 us of idle
// timeline -- rocprofv3 trace of a replayed step -- which ate the whole gain.)
template <int VEC>
__global__ __launch_bounds__(256) void grad_tail_kernel(const float* __restrict__ grad_rows, const int* __restrict__ pos_sorted,
                                                        const int* __restrict__ seg_of, int n, float* __restrict__ uniq_grad,
                                                        float* __restrict__ partial, int n_seg, const ReduceEntry* __restrict__ entries,
                                                        int red_bx, const int* __restrict__ blk_off, int n_entries) {
    if ((int)blockIdx.x < n_seg) { segreduce_chunks_block<VEC>(grad_rows, pos_sorted, seg_of, n, uniq_grad, partial, blockIdx.x); return; }
    const int rb = blockIdx.x - n_seg;
    if (blk_off == nullptr) { reduce_partials_block(entries[rb / red_bx], rb % red_bx, red_bx); return; }
    // blk_off [n_entries + 1]: entry e owns the blocks [blk_off[e], blk_off[e + 1]) -- as many as its size needs (with one grid
    // row of red_bx blocks per entry, most blocks of the many short entries -- biases, LayerNorm vectors -- found nothing to do)
    int lo = 0, hi = n_entries;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_off[mid] <= rb) lo = mid; else hi = mid;
    }
    reduce_partials_block(entries[lo], rb - blk_off[lo], blk_off[lo + 1] - blk_off[lo]);
}

// ---- the gradient tail of the live-sequence train step (amid_grad_tail_live_f32) ----------------------------------------------------------
// grad_tail_kernel's two roles over the step's COMPACT sorted list, plus a third: the position rows' gradients (tail_parts.h pos_sum_block).
// Phase B of the segment reduce rides in the optimizer launch (adam.hip optimizer_step_spans_kernel).
template <int VEC>
__global__ __launch_bounds__(256) void grad_tail_live_kernel(const float* __restrict__ grad_rows, const int* __restrict__ pos_sorted,
                                                             const int* __restrict__ seg_of, int n, float* __restrict__ uniq_grad,
                                                             float* __restrict__ partial, int n_seg, const ReduceEntry* __restrict__ entries,
                                                             const int* __restrict__ blk_off, int n_entries, int n_red, const PosSum ps,
                                                             const ScorerSum ss) {
    // the scorer sums first: their workgroups run the longest chains (B samples in eight groups)
    if ((int)blockIdx.x < ss.nblk) {
        __shared__ f32x4 sred[8 * 33];
        scorer_sum_block(ss, blockIdx.x, (scorer_lds_f4*)sred);
        return;
    }
    const int bid = blockIdx.x - ss.nblk;
    if (bid < n_seg) { segreduce_chunks_block<VEC>(grad_rows, pos_sorted, seg_of, n, uniq_grad, partial, bid); return; }
    const int rb = bid - n_seg;
    if (rb >= n_red) {
        const int pb = rb - n_red;
        pos_sum_block(ps, VEC * 64, pb / ps.nblk, pb % ps.nblk, ps.nblk);
        return;
    }
    int lo = 0, hi = n_entries;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (blk_off[mid] <= rb) lo = mid; else hi = mid;
    }
    reduce_partials_block(entries[lo], rb - blk_off[lo], blk_off[lo + 1] - blk_off[lo]);
}

// phase B: the chunk in which a border-crossing run STARTS owns its final sum (seg_spans.h)
template <int VEC>
__device__ __forceinline__ void segreduce_spans_block(int c, const int* __restrict__ seg_off, const int* __restrict__ seg_of, int n,
                                                      const float* __restrict__ partial, float* __restrict__ uniq_grad, int chunk) {
    const int D = VEC * 64;
    __shared__ float red[16][VEC * 64];
    int u, c_last;
    if (!spans_owner(c, seg_off, seg_of, n, chunk, u, c_last)) return;
    spans_partials<VEC, 16>(red, c, c_last, partial);
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 1024) uniq_grad[(long long)u * D + d] = spans_total<VEC>(red, d);
}

template <int VEC>
__global__ __launch_bounds__(1024) void segreduce_spans_kernel(const int* __restrict__ seg_off, const int* __restrict__ seg_of, int n,
                                                               const float* __restrict__ partial, float* __restrict__ uniq_grad,
                                                               int chunk = SEG_CHUNK) {
    segreduce_spans_block<VEC>(blockIdx.x, seg_off, seg_of, n, partial, uniq_grad, chunk);
}

// The data-parallel step's grad tail: phase B of the segment reduce, and behind it the packing of this rank's exchange chunk -- the
// unique ids padded to n_out with pad_id (blocks [nch, nch + id_blocks)) and a copy of the flat dense gradient, which the first
// launch of the tail has just summed (the remaining blocks).  The gradient rows need no copy: the segment reduce writes them into
// the chunk.  Rows past n_uniq stay stale; their ids say "padding".
template <int VEC>
__global__ __launch_bounds__(1024) void segreduce_spans_pack_kernel(const int* __restrict__ seg_off, const int* __restrict__ seg_of, int n,
                                                                    const float* __restrict__ partial, float* __restrict__ uniq_grad, int nch,
                                                                    const int* __restrict__ ids, const int* __restrict__ n_uniq, int n_out,
                                                                    int pad_id, int* __restrict__ out_ids, int id_blocks,
                                                                    const float* __restrict__ dense_src, float* __restrict__ dense_dst,
                                                                    long long dense_n, int* __restrict__ err) {
    const int b = blockIdx.x;
    if (b < nch) {
        segreduce_spans_block<VEC>(b, seg_off, seg_of, n, partial, uniq_grad, SEG_CHUNK);
        return;
    }
    if (b < nch + id_blocks) {
        const int r = (b - nch) * 1024 + threadIdx.x;
        if (r < n_out) out_ids[r] = r < *n_uniq ? ids[r] : pad_id;
        // more unique rows than the caller's bound: the segment reduce has written rows past the chunk's row part (into its dense
        // tail) -- the step is corrupt; say so (the host checks the flag at its next synchronisation point)
        if (r == 0 && err != nullptr && *n_uniq > n_out) atomicOr(err, AMID_FLAG_UMAX_EXCEEDED);
        return;
    }
    const long long nb = gridDim.x - nch - id_blocks;
    for (long long i = ((long long)(b - nch - id_blocks) * 1024 + threadIdx.x) * 4; i < dense_n; i += nb * 4096) {
        if (i + 4 <= dense_n) st4(dense_dst + i, ld4(dense_src + i));
        else for (long long k = i; k < dense_n; ++k) dense_dst[k] = dense_src[k];
    }
}

// Data-parallel exchange: a rank's (unique ids, gradient rows) padded to the world's largest count with (pad_id, zero row)
// pairs (amid_amd/dist.py); pad_id < 0 = repeat the first id (adds exact zeros to a real row).  One half-wave per row.
// Blocks past n_pad (optional) run fixed-order sums of reduce_partials.h: the copy of the flat dense gradient behind the rows.
__global__ __launch_bounds__(256) void sparse_pad_kernel(const int* __restrict__ ids, const float* __restrict__ rows,
                                                         const int* __restrict__ n_uniq, int n_out, int D, int pad_id,
                                                         int* __restrict__ out_ids, float* __restrict__ out_rows, int n_pad,
                                                         const ReduceEntry* __restrict__ entries, int red_bx) {
    if ((int)blockIdx.x >= n_pad) {
        const int rb = blockIdx.x - n_pad;
        reduce_partials_block(entries[rb / red_bx], rb % red_bx, red_bx);
        return;
    }
    const int sub = threadIdx.x & 31;
    const int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    if (r >= n_out) return;
    const int n = *n_uniq;
    const bool live = r < n;
    if (sub == 0) out_ids[r] = live ? ids[r] : (pad_id < 0 ? ids[0] : pad_id);
    for (int c = sub; c < (D >> 2); c += 32)
        st4(out_rows + (long long)r * D + 4 * c, live ? ld4(rows + (long long)r * D + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f));
}

}  // namespace amid

using namespace amid;

extern "C" long long amid_segreduce_workspace_bytes(int n_idx, int D) {
    const long long nch = (n_idx + SEG_CHUNK_SHORT - 1) / SEG_CHUNK_SHORT;       // the finer of the two chunkings
    return nch * 2 * D * 4 + 256;
}

extern "C" int amid_embgrad_segreduce_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of,
                                          int n_idx, int D, void* workspace, float* uniq_grad, void* stream) {
    AMID_CHECK_ARG(grad_rows && pos_sorted && seg_off && seg_of && workspace && uniq_grad && n_idx > 0);
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    // standalone use = the data-parallel merge (ids repeat at most `world` times) and the module-level gather backward: short lists
    // take the fine chunking, 4x the waves for the same rows (3 072 rows: 11.9 -> see DESIGN.md)
    const int chunk = n_idx <= 65536 ? SEG_CHUNK_SHORT : SEG_CHUNK;
    const int nch = (n_idx + chunk - 1) / chunk;
    float* partial = (float*)workspace;
#define AMID_SEG_LAUNCH(VEC)                                                                                                  \
    segreduce_chunks_kernel<VEC><<<(nch + 3) / 4, 256, 0, s>>>(grad_rows, pos_sorted, seg_of, n_idx, uniq_grad, partial, chunk); \
    segreduce_spans_kernel<VEC><<<nch, 1024, 0, s>>>(seg_off, seg_of, n_idx, partial, uniq_grad, chunk);
    if (D == 64) { AMID_SEG_LAUNCH(1) } else if (D == 128) { AMID_SEG_LAUNCH(2) } else { AMID_SEG_LAUNCH(4) }
#undef AMID_SEG_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// amid_embgrad_segreduce_f32 and amid_reduce_partials_f32 (sasrec_bwd.hip) with their first phases in ONE launch
// (spans = false, amid_grad_tail_nospans_f32: phase B of the segment reduce is left to amid_optimizer_step_spans_f32)
static int grad_tail(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                     void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count,
                     const int* blk_off, int total_blocks, void* stream, bool spans) {
    AMID_CHECK_ARG(blk_off == nullptr || total_blocks > 0);
    AMID_CHECK_ARG(grad_rows && pos_sorted && seg_off && seg_of && workspace && uniq_grad && n_idx > 0 && entries_dev && n_entries > 0 &&
                   max_count > 0);
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int nch = (n_idx + SEG_CHUNK - 1) / SEG_CHUNK, n_seg = (nch + 3) / 4;
    int bx = (max_count + 127) / 128;      // aligned entries move 128 elements per block pass; the (small) others loop
    if (bx > 512) bx = 512;
    float* partial = (float*)workspace;
    const ReduceEntry* en = (const ReduceEntry*)entries_dev;
#define AMID_TAIL_LAUNCH(VEC)                                                                                                       \
    grad_tail_kernel<VEC><<<n_seg + (blk_off ? total_blocks : bx * n_entries), 256, 0, s>>>(grad_rows, pos_sorted, seg_of, n_idx, uniq_grad, \
                                                                                           partial, n_seg, en, bx, blk_off, n_entries);     \
    if (spans) segreduce_spans_kernel<VEC><<<nch, 1024, 0, s>>>(seg_off, seg_of, n_idx, partial, uniq_grad);
    if (D == 64) { AMID_TAIL_LAUNCH(1) } else if (D == 128) { AMID_TAIL_LAUNCH(2) } else { AMID_TAIL_LAUNCH(4) }
#undef AMID_TAIL_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
extern "C" int amid_grad_tail_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                  void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count,
                                  const int* blk_off, int total_blocks, void* stream) {
    return grad_tail(grad_rows, pos_sorted, seg_off, seg_of, n_idx, D, workspace, uniq_grad, entries_dev, n_entries, max_count, blk_off, total_blocks, stream, true);
}
// ... without the second launch: the runs of the sorted list that cross 64-entry chunks stay as partial rows in `workspace`, and the caller's
// next launch is amid_optimizer_step_spans_f32 (same seg_off / seg_of / n_idx / workspace), which sums them in the same order and applies
// them on the spot -- one launch less in every single-GPU train step whose optimizer follows its gradient tail (round 5)
extern "C" int amid_grad_tail_nospans_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                          void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count,
                                          const int* blk_off, int total_blocks, void* stream) {
    return grad_tail(grad_rows, pos_sorted, seg_off, seg_of, n_idx, D, workspace, uniq_grad, entries_dev, n_entries, max_count, blk_off, total_blocks, stream, false);
}

// The gradient tail of the live-sequence step in ONE launch (grad_tail_live_kernel): phase A of the segment reduce over the step's
// compact sorted list (pos_sorted holds rows of grad_rows), the fixed-order partial sums of `entries` (blk_off: blocks per entry, as
// amid_grad_tail_f32) and the position rows' gradients dpos[g] [T, D] summed over the live sequences (live: amid_live_list_i32) straight
// from grad_rows.  Phase B of the segment reduce is NOT run here: amid_optimizer_step_spans_f32 finishes the runs that cross chunks.
// hidg != NULL (amid_head_fwd_bwd_own_vec_f32's per-sample hidden gradients; u [2, B, D], items [B, NI, D]): the scorer's weight gradients
// dW1 [hid, 2 D], db1 [hid], dW2 [hid], db2 [1] are formed here too (D <= 128).
extern "C" int amid_grad_tail_live_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                       void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off,
                                       int total_blocks, const int* live, int B, int T, float* dpos0, float* dpos1, const float* hidg,
                                       const float* u, const float* items, int NI, int hid, float* dW1, float* db1, float* dW2, float* db2,
                                       void* stream) {
    AMID_CHECK_ARG(grad_rows && pos_sorted && seg_off && seg_of && workspace && uniq_grad && n_idx > 0 && entries_dev && n_entries > 0 &&
                   blk_off && total_blocks > 0 && live && B > 0 && T > 0 && dpos0 && dpos1);
    AMID_CHECK_ARG(hidg == nullptr || (u && items && NI > 0 && hid > 0 && dW1 && db1 && dW2 && db2 && D <= 128 &&
                                       ((((unsigned long long)dW1) | ((unsigned long long)u) | ((unsigned long long)items)) & 15) == 0));
    AMID_CHECK_ARG(((((unsigned long long)dpos0) | ((unsigned long long)dpos1) | ((unsigned long long)grad_rows)) & 15) == 0);
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int nch = (n_idx + SEG_CHUNK - 1) / SEG_CHUNK, n_seg = (nch + 3) / 4;
    PosSum ps;
    ps.rows = grad_rows; ps.live = live; ps.B = B; ps.T = T; ps.dst[0] = dpos0; ps.dst[1] = dpos1;
    ps.nblk = (T * D + 127) / 128;
    const ScorerSum ss = scorer_sum_args(hidg, u, items, B, NI, D, hid, dW1, db1, dW2, db2);
    float* partial = (float*)workspace;
    const ReduceEntry* en = (const ReduceEntry*)entries_dev;
#define AMID_TAIL_LAUNCH(VEC)                                                                                                       \
    grad_tail_live_kernel<VEC><<<ss.nblk + n_seg + total_blocks + 2 * ps.nblk, 256, 0, s>>>(grad_rows, pos_sorted, seg_of, n_idx, uniq_grad, partial, \
                                                                                            n_seg, en, blk_off, n_entries, total_blocks, ps, ss);
    if (D == 64) { AMID_TAIL_LAUNCH(1) } else if (D == 128) { AMID_TAIL_LAUNCH(2) } else { AMID_TAIL_LAUNCH(4) }
#undef AMID_TAIL_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

// The folded step's gradient tail for a caller that SHIPS the row gradients (the data-parallel step, round 6): amid_grad_tail_live_f32's
// launch, then phase B of the segment reduce as its own launch -- the optimizer that would finish it runs on another rank's chunks too --
// either alone (n_out = 0: uniq_grad complete, what an eager exchange or the local-gradients graph reads) or with the packing of this
// rank's exchange chunk riding in it (amid_grad_tail_pack_f32's second launch: ids padded to n_out, the flat dense gradient copied behind
// the rows; uniq_grad points into the chunk).  Same additions in the same order as amid_grad_tail_live_f32 + amid_optimizer_step_spans_f32.
extern "C" int amid_grad_tail_live_dp_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                          void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off,
                                          int total_blocks, const int* live, int B, int T, float* dpos0, float* dpos1, const int* uniq_ids,
                                          const int* n_uniq, int n_out, int pad_id, int* out_ids, const float* dense_src, float* dense_dst,
                                          long long dense_n, int* err_flag, void* stream) {
    AMID_CHECK_ARG(grad_rows && pos_sorted && seg_off && seg_of && workspace && uniq_grad && n_idx > 0 && entries_dev && n_entries > 0 &&
                   blk_off && total_blocks > 0 && live && B > 0 && T > 0 && dpos0 && dpos1);
    AMID_CHECK_ARG(((((unsigned long long)dpos0) | ((unsigned long long)dpos1) | ((unsigned long long)grad_rows)) & 15) == 0);
    AMID_CHECK_ARG(n_out >= 0 && n_out <= n_idx && dense_n >= 0);
    AMID_CHECK_ARG(n_out == 0 || (uniq_ids && n_uniq && out_ids &&
                                  (dense_n == 0 || (dense_src && dense_dst && ((((unsigned long long)dense_src) | ((unsigned long long)dense_dst)) & 15) == 0))));
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int nch = (n_idx + SEG_CHUNK - 1) / SEG_CHUNK, n_seg = (nch + 3) / 4;
    PosSum ps;
    ps.rows = grad_rows; ps.live = live; ps.B = B; ps.T = T; ps.dst[0] = dpos0; ps.dst[1] = dpos1;
    ps.nblk = (T * D + 127) / 128;
    const ScorerSum ss = scorer_sum_args(nullptr, nullptr, nullptr, B, 0, D, 0, nullptr, nullptr, nullptr, nullptr);
    float* partial = (float*)workspace;
    const ReduceEntry* en = (const ReduceEntry*)entries_dev;
    const int id_blocks = (n_out + 1023) / 1024;
    long long cb = (dense_n / 4 + 1023) / 1024;
    if (cb < 1) cb = 1;
    if (cb > 256) cb = 256;
    if (dense_n == 0 || n_out == 0) cb = 0;
#define AMID_TAIL_LAUNCH(VEC)                                                                                                       \
    grad_tail_live_kernel<VEC><<<ss.nblk + n_seg + total_blocks + 2 * ps.nblk, 256, 0, s>>>(grad_rows, pos_sorted, seg_of, n_idx, uniq_grad, partial, \
                                                                                            n_seg, en, blk_off, n_entries, total_blocks, ps, ss); \
    if (n_out == 0) segreduce_spans_kernel<VEC><<<nch, 1024, 0, s>>>(seg_off, seg_of, n_idx, partial, uniq_grad);                   \
    else segreduce_spans_pack_kernel<VEC><<<nch + id_blocks + (int)cb, 1024, 0, s>>>(seg_off, seg_of, n_idx, partial, uniq_grad, nch, uniq_ids, n_uniq, \
                                                                                   n_out, pad_id, out_ids, id_blocks, dense_src, dense_dst, dense_n, err_flag);
    if (D == 64) { AMID_TAIL_LAUNCH(1) } else if (D == 128) { AMID_TAIL_LAUNCH(2) } else { AMID_TAIL_LAUNCH(4) }
#undef AMID_TAIL_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_grad_tail_pack_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                       void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count,
                                       const int* uniq_ids, const int* n_uniq, int n_out, int pad_id, int* out_ids, const float* dense_src,
                                       float* dense_dst, long long dense_n, int* err_flag, const int* blk_off, int total_blocks, void* stream) {
    AMID_CHECK_ARG(blk_off == nullptr || total_blocks > 0);
    AMID_CHECK_ARG(grad_rows && pos_sorted && seg_off && seg_of && workspace && uniq_grad && n_idx > 0 && entries_dev && n_entries > 0 &&
                   max_count > 0);
    // dense_n == 0: the chunk carries no dense part (the caller all-reduces the dense gradient itself)
    AMID_CHECK_ARG(uniq_ids && n_uniq && out_ids && n_out > 0 && n_out <= n_idx && dense_n >= 0 &&
                   (dense_n == 0 || (dense_src && dense_dst && ((((unsigned long long)dense_src) | ((unsigned long long)dense_dst)) & 15) == 0)));
    if (!(D == 64 || D == 128 || D == 256)) return AMID_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int nch = (n_idx + SEG_CHUNK - 1) / SEG_CHUNK, n_seg = (nch + 3) / 4;
    int bx = (max_count + 127) / 128;
    if (bx > 512) bx = 512;
    float* partial = (float*)workspace;
    const ReduceEntry* en = (const ReduceEntry*)entries_dev;
    const int id_blocks = (n_out + 1023) / 1024;
    long long cb = (dense_n / 4 + 1023) / 1024;
    if (cb < 1) cb = 1;
    if (cb > 256) cb = 256;
    if (dense_n == 0) cb = 0;
#define AMID_TAIL_LAUNCH(VEC)                                                                                                       \
    grad_tail_kernel<VEC><<<n_seg + (blk_off ? total_blocks : bx * n_entries), 256, 0, s>>>(grad_rows, pos_sorted, seg_of, n_idx, uniq_grad, \
                                                                                           partial, n_seg, en, bx, blk_off, n_entries);     \
    segreduce_spans_pack_kernel<VEC><<<nch + id_blocks + (int)cb, 1024, 0, s>>>(seg_off, seg_of, n_idx, partial, uniq_grad, nch, uniq_ids, n_uniq, \
                                                                              n_out, pad_id, out_ids, id_blocks, dense_src, dense_dst, dense_n, err_flag);
    if (D == 64) { AMID_TAIL_LAUNCH(1) } else if (D == 128) { AMID_TAIL_LAUNCH(2) } else { AMID_TAIL_LAUNCH(4) }
#undef AMID_TAIL_LAUNCH
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int sparse_pad(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int n_out, int D, int pad_id, int* out_ids,
                      float* out_rows, const void* entries_dev, int n_entries, int max_count, void* stream) {
    AMID_CHECK_ARG(uniq_ids && uniq_rows && n_uniq && out_ids && out_rows && n_out > 0 && D > 0 && (D % 4) == 0);
    AMID_CHECK_ARG(n_entries == 0 || (entries_dev && n_entries > 0 && max_count > 0));
    const int n_pad = (n_out + 7) / 8;
    int bx = n_entries ? (max_count + 127) / 128 : 0;
    if (bx > 512) bx = 512;
    sparse_pad_kernel<<<n_pad + bx * n_entries, 256, 0, (hipStream_t)stream>>>(uniq_ids, uniq_rows, n_uniq, n_out, D, pad_id, out_ids, out_rows,
                                                                              n_pad, (const ReduceEntry*)entries_dev, bx > 0 ? bx : 1);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_sparse_pad_f32(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int n_out, int D, int pad_id,
                                   int* out_ids, float* out_rows, void* stream) {
    return sparse_pad(uniq_ids, uniq_rows, n_uniq, n_out, D, pad_id, out_ids, out_rows, nullptr, 0, 0, stream);
}

// the same padding with `n_entries` fixed-order sums (amid_reduce_entry_pack tables) in the same launch
extern "C" int amid_sparse_pad_sum_f32(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int n_out, int D, int pad_id,
                                       int* out_ids, float* out_rows, const void* entries_dev, int n_entries, int max_count,
                                       void* stream) {
    AMID_CHECK_ARG(entries_dev && n_entries > 0);
    return sparse_pad(uniq_ids, uniq_rows, n_uniq, n_out, D, pad_id, out_ids, out_rows, entries_dev, n_entries, max_count, stream);
}
