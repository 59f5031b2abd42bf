// Index preparation for the sparse side of the step: a stable LSD radix sort of (item id, position)
// pairs followed by run detection.  Produces, for the N_idx gathered indices of one step,
//   pos_sorted[N]  positions ordered by (id, position)         -> inverted index for the segment reduce
//   uniq_ids[U]    distinct ids, ascending                     -> rows the lazy Adam has to touch
//   seg_off[U+1]   run starts in pos_sorted (seg_off[U] = N)
//   seg_of[N]      run index of every sorted entry (saves the segment reduce a binary search per chunk)
//   n_uniq         U (device scalar; everything downstream reads it from memory => graph-replay safe)
// There is no reference counterpart: the reference lets autograd build four dense 458 MB
// gradients (nn.Embedding(sparse=False), model_seq.py:25) and runs dense Adam over them
// (train_sr.py:480).  Sorting fixes the summation order of every row gradient => reproducible.
//
// 8-bit digits, ceil(bits(n_rows-1)/8) passes; per pass: per-block digit counts -> one-block scan ->
// stable scatter.  Ranking inside a wave uses ballot-based match-any, so a wave full of the pad id
// (83-91 % of all indices) costs one LDS add per wave, not 64 serialized atomics.
#include "common.h"
#include "reduce_partials.h"
#include "sort_phases.h"

namespace amid {


__device__ __forceinline__ unsigned long long match_digit(unsigned d, bool valid) {
    // lanes of this wave holding the same 8-bit digit (invalid lanes match nobody)
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
    }
    return peers;
}

// counts[digit * nblk + blk]
__global__ __launch_bounds__(SORT_THREADS) void radix_count_kernel(const int* __restrict__ keys, int n, int shift, int nblk,
                                                                   int* __restrict__ counts) {
    __shared__ int hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const int base = blockIdx.x * SORT_TILE + w * (64 * SORT_ITEMS);
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = base + i * 64 + lane;
        const bool valid = k < n;
        const unsigned d = valid ? (((unsigned)keys[k] >> shift) & 0xFFu) : 0u;
        const unsigned long long peers = match_digit(d, valid);
        if (valid && (__ffsll((long long)peers) - 1) == lane) atomicAdd(&hist[d], __popcll(peers));
    }
    __syncthreads();
    counts[threadIdx.x * nblk + blockIdx.x] = hist[threadIdx.x];
}

// exclusive scan of counts[0 .. 256*nblk) in place (digit-major order = final output order)
__global__ __launch_bounds__(1024) void radix_scan_kernel(int* __restrict__ counts, int total) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = lane_id(), w = wave_id();
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < total) ? counts[i] : 0;
        int x = v;                                   // inclusive scan inside the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (i < total) counts[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(const int* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                                     int* __restrict__ keys_out, int* __restrict__ vals_out, int n,
                                                                     int shift, int nblk, const int* __restrict__ offsets, int first_pass) {
    __shared__ int whist[SORT_WAVES][256];          // per-wave digit totals, then running output cursors
    const int w = wave_id(), lane = lane_id();
    for (int i = threadIdx.x; i < SORT_WAVES * 256; i += SORT_THREADS) (&whist[0][0])[i] = 0;
    __syncthreads();
    const int base = blockIdx.x * SORT_TILE + w * (64 * SORT_ITEMS);
    int key[SORT_ITEMS];
    unsigned long long peers[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = base + i * 64 + lane;
        const bool valid = k < n;
        key[i] = valid ? keys_in[k] : 0;
        const unsigned d = ((unsigned)key[i] >> shift) & 0xFFu;
        peers[i] = match_digit(d, valid);
        if (valid && (__ffsll((long long)peers[i]) - 1) == lane) atomicAdd(&whist[w][d], __popcll(peers[i]));
    }
    __syncthreads();
    {   // cursor[w][d] = global offset of (d, this block) + totals of earlier waves
        const int d = threadIdx.x;
        int run = offsets[d * nblk + blockIdx.x];
#pragma unroll
        for (int k = 0; k < SORT_WAVES; ++k) {
            const int c = whist[k][d];
            whist[k][d] = run;
            run += c;
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = base + i * 64 + lane;
        const bool valid = k < n;
        const unsigned d = ((unsigned)key[i] >> shift) & 0xFFu;
        int dst = 0;
        if (valid) dst = whist[w][d] + __popcll(peers[i] & lt);
        __builtin_amdgcn_wave_barrier();            // every lane has read the cursor before the leader bumps it
        if (valid && (__ffsll((long long)peers[i]) - 1) == lane) whist[w][d] += __popcll(peers[i]);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[dst] = key[i];
            vals_out[dst] = (first_pass && vals_in == nullptr) ? k : vals_in[k];
        }
    }
}

// run heads of the sorted keys: per-block head counts
__global__ __launch_bounds__(256) void heads_count_kernel(const int* __restrict__ keys, int n, int* __restrict__ blk_heads) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool head = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
    const unsigned long long m = __ballot(head);
    if (lane_id() == 0 && m) atomicAdd(&cnt, __popcll(m));
    __syncthreads();
    if (threadIdx.x == 0) blk_heads[blockIdx.x] = cnt;
}

// one block: exclusive scan of blk_heads (in place), total -> n_uniq, seg_off[total] = n
// keys / sentinel (optional): when the largest key equals `sentinel` (padding of the data-parallel merge) its run is not counted
__global__ __launch_bounds__(1024) void heads_scan_kernel(int* __restrict__ blk_heads, int nblk, int* __restrict__ n_uniq,
                                                          int* __restrict__ seg_off, int n, const int* __restrict__ keys = nullptr,
                                                          int sentinel = -1) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = lane_id(), w = wave_id();
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < nblk) ? blk_heads[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        const int carry = carry_s;
        if (i < nblk) blk_heads[i] = carry + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *n_uniq = (keys != nullptr && keys[n - 1] == sentinel) ? carry_s - 1 : carry_s;
        seg_off[carry_s] = n;
    }
}

__global__ __launch_bounds__(256) void heads_write_kernel(const int* __restrict__ keys, int n, const int* __restrict__ blk_base,
                                                          int* __restrict__ uniq_ids, int* __restrict__ seg_off, int* __restrict__ seg_of) {
    __shared__ int wcnt[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id(), w = wave_id();
    const bool head = (i < n) && (i == 0 || keys[i] != keys[i - 1]);
    const unsigned long long m = __ballot(head);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int off = blk_base[blockIdx.x];
    for (int k = 0; k < w; ++k) off += wcnt[k];
    const int u = off + __popcll(m & ((2ull << lane) - 1ull)) - 1;      // run index of sorted entry i (heads up to and including lane)
    if (i < n) seg_of[i] = u;
    if (head) {
        uniq_ids[u] = keys[i];
        seg_off[u] = i;
    }
}

// the three kernels above as ONE workgroup for short lists (n <= HEADS_FUSED_MAX: every per-step index list of the cfg 1-4 shapes
// and the data-parallel merge up to 16 ranks x 4096 rows): thread t owns the contiguous span [t * per, (t + 1) * per), counts its run
// heads, the 1024 counts are scanned through LDS, and a second pass over the same span writes uniq_ids / seg_off / seg_of.  Three
// dependent ~5 us launches become one of about that length (they sit on the critical path of the data-parallel step's merge).
constexpr int HEADS_FUSED_MAX = 8192;       // beyond: the three short kernels (one 1024-thread workgroup walking 26 k entries ran 58 us and
                                            // held a CU against the one-round kernels of the main stream)
__global__ __launch_bounds__(1024) void heads_fused_kernel(const int* __restrict__ keys, int n, int* __restrict__ n_uniq,
                                                           int* __restrict__ uniq_ids, int* __restrict__ seg_off, int* __restrict__ seg_of,
                                                           int use_sentinel, int sentinel) {
    __shared__ int wsum[16];
    const int lane = lane_id(), w = wave_id();
    const int per = (n + 1023) >> 10;
    const int i0 = min((int)threadIdx.x * per, n), i1 = min(i0 + per, n);
    int cnt = 0;
    {
        int prev = (i0 > 0 && i0 < n) ? keys[i0 - 1] : 0;
        for (int i = i0; i < i1; ++i) {
            const int k = keys[i];
            cnt += (i == 0 || k != prev) ? 1 : 0;
            prev = k;
        }
    }
    int x = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0, total = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) woff += wsum[k]; total += wsum[k]; }
    int u = woff + x - cnt - 1;                          // run index of the entry before this span
    {
        int prev = (i0 > 0 && i0 < n) ? keys[i0 - 1] : 0;
        for (int i = i0; i < i1; ++i) {
            const int k = keys[i];
            if (i == 0 || k != prev) { ++u; uniq_ids[u] = k; seg_off[u] = i; }
            seg_of[i] = u;
            prev = k;
        }
    }
    if (threadIdx.x == 0) {
        *n_uniq = (use_sentinel && keys[n - 1] == sentinel) ? total - 1 : total;
        seg_off[total] = n;
    }
}

// the four phases as launches of their own (bodies: sort_phases.h)
template <int BINS>
__global__ __launch_bounds__(SORT_THREADS) void os_count_kernel(const int* __restrict__ keys, OsGeom g, int* __restrict__ state,
                                                                int* __restrict__ counts0, int* __restrict__ stot0, int stot0_copy, int n_stot0,
                                                                int* __restrict__ counts1, long long n_counts1, int* __restrict__ stot1,
                                                                int n_stot1, int* __restrict__ hstatus) {
    __shared__ int hist[BINS];
    os_count_block<BINS>(blockIdx.x, gridDim.x, ArrayKeys{keys}, g, state, counts0, stot0, stot0_copy, n_stot0, counts1, n_counts1, stot1, n_stot1, hstatus, hist);
}
template <bool COUNT_NEXT, int BINS>
__global__ __launch_bounds__(SORT_THREADS) void os_scatter_kernel(const int* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                                  int* __restrict__ keys_out, int* __restrict__ vals_out, OsGeom g,
                                                                  const int* __restrict__ counts, const int* __restrict__ stot, int nsup,
                                                                  int* __restrict__ counts_next, int* __restrict__ stot_next,
                                                                  const int* __restrict__ state, int stot_copy) {
    // pass 0 reads the stot_0 copy this call fills (os_count_block)
    const int* st = COUNT_NEXT ? stot + (long long)(state[3] & 1) * stot_copy : stot;
    __shared__ SortScatterLds<BINS> lds;
    os_scatter_block<COUNT_NEXT, BINS>(blockIdx.x, gridDim.x, keys_in, vals_in, keys_out, vals_out, g, counts, st, nsup, counts_next, stot_next, lds);
}
__global__ __launch_bounds__(SORT_THREADS) void os_heads_kernel(const int* __restrict__ keys, int n, int ntiles, int* __restrict__ state,
                                                                unsigned* __restrict__ hstatus, int* __restrict__ n_uniq,
                                                                int* __restrict__ uniq_ids, int* __restrict__ seg_off, int* __restrict__ seg_of) {
    __shared__ SortHeadsLds lds;
    os_heads_block(keys, n, ntiles, state, hstatus, n_uniq, uniq_ids, seg_off, seg_of, lds);
}

// Stable merge of `world` sorted lists of `len` keys each (list r = keys[r * len ..]): the merged position of entry (r, i) with
// key x is i + sum over the other lists of (# keys < x), or (# keys <= x) for lists of lower rank -- ties go in rank order, so the
// result equals a stable sort of the concatenation.  All binary searches of a thread advance in lock-step (independent loads).
constexpr int MERGE_MAX_WORLD = 16;
// key_stride: distance (in ints) between two ranks' lists; the merged position table holds row_base + r * row_stride + i for
// entry (r, i), i.e. the row of that entry's gradient in whatever layout the gathered buffer has (packed exchange, dist.py).
__device__ __forceinline__ void merge_rank_block(const int* __restrict__ keys, int world, int len, long long key_stride,
                                                 int row_base, int row_stride, int* __restrict__ keys_sorted,
                                                 int* __restrict__ pos_sorted, int block) {
    const int e = block * 256 + threadIdx.x;
    if (e >= world * len) return;
    const int r = e / len, i = e - r * len;
    const int x = keys[r * key_stride + i];
    int lo[MERGE_MAX_WORLD], hi[MERGE_MAX_WORLD];
#pragma unroll
    for (int q = 0; q < MERGE_MAX_WORLD; ++q) { lo[q] = 0; hi[q] = len; }
    for (int span = len; span > 0; span >>= 1) {               // ceil(log2(len)) + 1 rounds close every interval
#pragma unroll
        for (int q = 0; q < MERGE_MAX_WORLD; ++q) {
            if (q < world && q != r && lo[q] < hi[q]) {
                const int mid = (lo[q] + hi[q]) >> 1;
                const int v = keys[q * key_stride + mid];
                const bool left = (q < r) ? (v <= x) : (v < x);      // lower ranks: upper bound; higher ranks: lower bound
                if (left) lo[q] = mid + 1; else hi[q] = mid;
            }
        }
    }
    int p = i;
#pragma unroll
    for (int q = 0; q < MERGE_MAX_WORLD; ++q)
        if (q < world && q != r) {
            while (lo[q] < hi[q]) {                                  // (defensive: the loop above already closed it)
                const int mid = (lo[q] + hi[q]) >> 1;
                const int v = keys[q * key_stride + mid];
                const bool left = (q < r) ? (v <= x) : (v < x);
                if (left) lo[q] = mid + 1; else hi[q] = mid;
            }
            p += lo[q];
        }
    keys_sorted[p] = x;
    pos_sorted[p] = row_base + r * row_stride + i;
}

// blocks [0, n_merge): the merge; the rest (optional): fixed-order sums of reduce_partials.h -- the rank-ordered sum of the dense
// gradients that travelled behind the sparse rows is independent of the merge and rides in the same launch
__global__ __launch_bounds__(256) void merge_rank_kernel(const int* __restrict__ keys, int world, int len, long long key_stride,
                                                         int row_base, int row_stride, int* __restrict__ keys_sorted,
                                                         int* __restrict__ pos_sorted, int n_merge, const ReduceEntry* __restrict__ entries,
                                                         int red_bx) {
    if ((int)blockIdx.x < n_merge) { merge_rank_block(keys, world, len, key_stride, row_base, row_stride, keys_sorted, pos_sorted, blockIdx.x); return; }
    const int rb = blockIdx.x - n_merge;
    reduce_partials_block(entries[rb / red_bx], rb % red_bx, red_bx);
}

}  // namespace amid

using namespace amid;

static inline int sort_nblk(int n) { return (n + SORT_TILE - 1) / SORT_TILE; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// four-launch sort.  Head of the workspace, independent of n_idx: [state | stot0 (OS_SUPER_MAX rows)]; behind the 8-bit sort's
// arrays: [stot1 | counts0 | counts1 | hstatus]
static inline size_t os_nsuper(size_t ntiles) { return (ntiles + OS_SUPER - 1) / OS_SUPER; }
static inline size_t os_head_bytes() { return align256(OS_STATE_INTS * 4) + 2 * align256((size_t)OS_SUPER_MAX * OS_BINS_MAX * 4); }      // two stot_0 copies
static inline size_t os_bytes(size_t ntiles) {
    return align256(os_nsuper(ntiles) * OS_BINS_MAX * 4) + 2 * align256(ntiles * OS_BINS_MAX * 4) + align256(ntiles * 4);
}

extern "C" long long amid_sort_unique_workspace_bytes(int n_idx) {
    if (n_idx <= 0) return 256;
    const size_t nblk = sort_nblk(n_idx);
    size_t b = os_head_bytes();
    b += 4 * align256((size_t)n_idx * 4);            // keys a/b, vals a/b
    b += align256(256 * nblk * 4);                   // digit counts / offsets
    b += align256(((size_t)n_idx + 255) / 256 * 4);  // per-block head counts
    b += os_bytes(nblk);                             // state of the four-launch sort
    return (long long)b;
}

// Lists of at least this many indices take the four-launch sort.  Shorter ones (every per-step list of the cfg 1-4 shapes) keep the
// 8-bit passes: that sort runs on a side stream beside the encoder, where its many 5-9 us launches of 13 small blocks cost nothing,
// while the four ~15 us launches of the two-pass sort hold their CUs long enough to delay the one-round kernels of the main stream
// (measured at cfg 2: 0.408 -> 0.432 ms per step).  At cfg 5 (418 k indices) the sort is exposed: 574 -> 125 us.
static int g_four_launch_min = 65536;

extern "C" int amid_sort_set_four_launch_min(int n_idx) {
    const int prev = g_four_launch_min;
    if (n_idx >= 0) g_four_launch_min = n_idx;
    return prev;
}

// the four-phase sort's pointers into the caller's workspace (layout: amid_sort_unique_workspace_bytes)
static int make_plan(SortPlan& sp, const int* idx, const int* rows, int n_idx, long long n_rows, void* workspace, int* pos_sorted, int* uniq_ids,
                     int* seg_off, int* seg_of, int* n_uniq) {
    const int nblk = sort_nblk(n_idx);
    const size_t nsup = os_nsuper(nblk);
    int bits = 1;
    while (bits < 31 && (1LL << bits) < n_rows) ++bits;
    if (bits > 24 || nsup > (size_t)OS_SUPER_MAX) return AMID_ERR_UNSUPPORTED;
    if (bits < 2) bits = 2;
    const int b0 = (bits + 1) / 2, b1 = bits - b0;
    char* const ws_head = (char*)workspace;
    char* ws = ws_head + os_head_bytes();
    const size_t kb = align256((size_t)n_idx * 4);
    char* os = ws + 4 * kb + align256((size_t)256 * nblk * 4) + align256(((size_t)n_idx + 255) / 256 * 4);
    sp.idx = idx; sp.rows = rows; sp.n = n_idx; sp.nblk = nblk; sp.nsup = (int)nsup;
    sp.g0 = OsGeom{n_idx, nblk, 0, b0, b0, b1};
    sp.g1 = OsGeom{n_idx, nblk, b0, b1, 0, b0};
    sp.state = (int*)ws_head;
    sp.stot0 = (int*)(ws_head + align256(OS_STATE_INTS * 4));
    sp.stot1 = (int*)os;
    sp.counts0 = (int*)((char*)sp.stot1 + align256(nsup * OS_BINS_MAX * 4));
    sp.counts1 = (int*)((char*)sp.counts0 + align256((size_t)nblk * OS_BINS_MAX * 4));
    sp.hstatus = (unsigned*)((char*)sp.counts1 + align256((size_t)nblk * OS_BINS_MAX * 4));
    sp.keys0 = (int*)ws; sp.keys1 = (int*)(ws + kb); sp.vals0 = (int*)(ws + 2 * kb);
    sp.pos_sorted = pos_sorted; sp.uniq_ids = uniq_ids; sp.seg_off = seg_off; sp.seg_of = seg_of; sp.n_uniq = n_uniq;
    sp.n_counts1 = (long long)nblk << b1; sp.n_stot1 = (int)(nsup << b1); sp.n_zero_a = (int)(nsup << b0);
    sp.stot0_copy = (int)(align256((size_t)OS_SUPER_MAX * OS_BINS_MAX * 4) / 4);
    return AMID_OK;
}

extern "C" int amid_sort_plan_bytes(void) { return (int)sizeof(SortPlan); }

// The sort of amid_sort_unique_i32 / amid_sort_unique_rows_i32 (rows optional) as a PLAN: nothing is launched; launches of the train
// step that take a plan + a phase (1 .. 4, in this order, each in a later launch of the same stream than the one before) run the
// sort as extra workgroups of their own (keys below 2^24 and at most OS_SUPER_MAX supertiles; AMID_ERR_UNSUPPORTED otherwise).  host_buf: amid_sort_plan_bytes().
extern "C" int amid_sort_plan_pack(void* host_buf, const int* idx, const int* rows, int n_idx, long long n_rows, void* workspace,
                                   int* pos_sorted, int* uniq_ids, int* seg_off, int* seg_of, int* n_uniq) {
    AMID_CHECK_ARG(host_buf && idx && workspace && pos_sorted && uniq_ids && seg_off && seg_of && n_uniq && n_idx > 0 && n_rows > 0);
    SortPlan sp;
    if (int e = make_plan(sp, idx, rows, n_idx, n_rows, workspace, pos_sorted, uniq_ids, seg_off, seg_of, n_uniq)) return e;
    *(SortPlan*)host_buf = sp;
    return AMID_OK;
}

static int sort_unique(const int* idx, const int* rows, int n_idx, long long n_rows, void* workspace, int* pos_sorted, int* uniq_ids,
                       int* seg_off, int* seg_of, int* n_uniq, void* stream) {
    AMID_CHECK_ARG(idx && workspace && pos_sorted && uniq_ids && seg_off && seg_of && n_uniq && n_idx > 0 && n_rows > 0);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = sort_nblk(n_idx);
    char* const ws_head = (char*)workspace;                    // [state | stot0]: fixed, whatever n_idx
    char* ws = ws_head + os_head_bytes();
    const size_t kb = align256((size_t)n_idx * 4);
    int* keys[2] = {(int*)ws, (int*)(ws + kb)};
    int* vals[2] = {(int*)(ws + 2 * kb), (int*)(ws + 3 * kb)};
    int* counts = (int*)(ws + 4 * kb);
    int* blk_heads = (int*)(ws + 4 * kb + align256((size_t)256 * nblk * 4));
    int bits = 1;
    while (bits < 31 && (1LL << bits) < n_rows) ++bits;
    if (bits <= 24 && n_idx >= g_four_launch_min && os_nsuper(nblk) <= (size_t)OS_SUPER_MAX) {      // four launches (sort_phases.h)
        SortPlan sp;
        if (make_plan(sp, idx, rows, n_idx, n_rows, workspace, pos_sorted, uniq_ids, seg_off, seg_of, n_uniq) != AMID_OK) return AMID_ERR_ARG;
#define AMID_OS_LAUNCH(BINS)                                                                                                            \
        os_count_kernel<BINS><<<nblk, SORT_THREADS, 0, s>>>(sp.idx, sp.g0, sp.state, sp.counts0, sp.stot0, sp.stot0_copy, sp.n_zero_a, sp.counts1,   \
                                                            sp.n_counts1, sp.stot1, sp.n_stot1, (int*)sp.hstatus);                       \
        os_scatter_kernel<true, BINS><<<nblk, SORT_THREADS, 0, s>>>(sp.idx, sp.rows, sp.keys0, sp.vals0, sp.g0, sp.counts0, sp.stot0, sp.nsup,   \
                                                                    sp.counts1, sp.stot1, sp.state, sp.stot0_copy);                        \
        os_scatter_kernel<false, BINS><<<nblk, SORT_THREADS, 0, s>>>(sp.keys0, sp.vals0, sp.keys1, sp.pos_sorted, sp.g1, sp.counts1, sp.stot1,  \
                                                                     sp.nsup, nullptr, nullptr, sp.state, 0);
        if (sp.g0.bits <= 10) { AMID_OS_LAUNCH(1024) } else { AMID_OS_LAUNCH(4096) }
#undef AMID_OS_LAUNCH
        os_heads_kernel<<<nblk, SORT_THREADS, 0, s>>>(sp.keys1, sp.n, sp.nblk, sp.state, sp.hstatus, sp.n_uniq, sp.uniq_ids, sp.seg_off, sp.seg_of);
        AMID_LAUNCH_CHECK();
        return AMID_OK;
    }
    const int passes = (bits + 7) / 8;
    const int* kin = idx;
    const int* vin = rows;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        int* kout = keys[p & 1];
        int* vout = (p == passes - 1) ? pos_sorted : vals[p & 1];
        radix_count_kernel<<<nblk, SORT_THREADS, 0, s>>>(kin, n_idx, shift, nblk, counts);
        radix_scan_kernel<<<1, 1024, 0, s>>>(counts, 256 * nblk);
        radix_scatter_kernel<<<nblk, SORT_THREADS, 0, s>>>(kin, vin, kout, vout, n_idx, shift, nblk, counts, p == 0);
        kin = kout;
        vin = vout;
    }
    const int hblk = (n_idx + 255) / 256;
    if (n_idx <= HEADS_FUSED_MAX) {
        heads_fused_kernel<<<1, 1024, 0, s>>>(kin, n_idx, n_uniq, uniq_ids, seg_off, seg_of, 0, 0);
    } else {
        heads_count_kernel<<<hblk, 256, 0, s>>>(kin, n_idx, blk_heads);
        heads_scan_kernel<<<1, 1024, 0, s>>>(blk_heads, hblk, n_uniq, seg_off, n_idx);
        heads_write_kernel<<<hblk, 256, 0, s>>>(kin, n_idx, blk_heads, uniq_ids, seg_off, seg_of);
    }
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_sort_unique_i32(const int* idx, int n_idx, long long n_rows, void* workspace, int* pos_sorted, int* uniq_ids,
                                    int* seg_off, int* seg_of, int* n_uniq, void* stream) {
    return sort_unique(idx, nullptr, n_idx, n_rows, workspace, pos_sorted, uniq_ids, seg_off, seg_of, n_uniq, stream);
}

// the same with a payload: pos_sorted holds rows[i] instead of i (the sort of a compact index list whose entries' gradient rows
// stand elsewhere: amid_lazy_adam_catchup_live_f32); ties keep the order of the list
extern "C" int amid_sort_unique_rows_i32(const int* idx, const int* rows, int n_idx, long long n_rows, void* workspace, int* pos_sorted,
                                         int* uniq_ids, int* seg_off, int* seg_of, int* n_uniq, void* stream) {
    AMID_CHECK_ARG(rows != nullptr);
    return sort_unique(idx, rows, n_idx, n_rows, workspace, pos_sorted, uniq_ids, seg_off, seg_of, n_uniq, stream);
}

// Data-parallel merge (amid_amd/dist.py): `world` lists of `len` keys, each non-decreasing (a rank's unique ids in ascending order,
// then `sentinel` padding with sentinel > every id) -> the same outputs as amid_sort_unique_i32 on the concatenation, in 4 launches
// instead of a full radix sort; the sentinel run, if any, is left out of n_uniq.  Workspace: amid_sort_unique_workspace_bytes(world*len).
static int merge_sorted_lists(const int* keys, int world, int len, long long key_stride, int row_base, int row_stride, int sentinel,
                              void* workspace, int* pos_sorted, int* uniq_ids, int* seg_off, int* seg_of, int* n_uniq,
                              const void* entries_dev, int n_entries, int max_count, void* stream) {
    AMID_CHECK_ARG(keys && workspace && pos_sorted && uniq_ids && seg_off && seg_of && n_uniq && world > 0 && world <= MERGE_MAX_WORLD &&
                   len > 0 && key_stride >= len && row_base >= 0 && row_stride >= len);
    AMID_CHECK_ARG(n_entries == 0 || (entries_dev && n_entries > 0 && max_count > 0));
    hipStream_t s = (hipStream_t)stream;
    const int n = world * len;
    char* ws = (char*)workspace + os_head_bytes();             // (the head belongs to the four-launch sort)
    const size_t kb = align256((size_t)n * 4);
    int* keys_sorted = (int*)ws;
    int* blk_heads = (int*)(ws + 4 * kb + align256((size_t)256 * sort_nblk(n) * 4));
    const int n_merge = (n + 255) / 256;
    int bx = n_entries ? (max_count + 127) / 128 : 0;
    if (bx > 512) bx = 512;
    merge_rank_kernel<<<n_merge + bx * n_entries, 256, 0, s>>>(keys, world, len, key_stride, row_base, row_stride, keys_sorted, pos_sorted, n_merge,
                                                             (const ReduceEntry*)entries_dev, bx > 0 ? bx : 1);
    const int hblk = (n + 255) / 256;
    if (n <= HEADS_FUSED_MAX) {
        heads_fused_kernel<<<1, 1024, 0, s>>>(keys_sorted, n, n_uniq, uniq_ids, seg_off, seg_of, 1, sentinel);
    } else {
        heads_count_kernel<<<hblk, 256, 0, s>>>(keys_sorted, n, blk_heads);
        heads_scan_kernel<<<1, 1024, 0, s>>>(blk_heads, hblk, n_uniq, seg_off, n, keys_sorted, sentinel);
        heads_write_kernel<<<hblk, 256, 0, s>>>(keys_sorted, n, blk_heads, uniq_ids, seg_off, seg_of);
    }
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_merge_sorted_lists_i32(const int* keys, int world, int len, long long key_stride, int row_base, int row_stride,
                                           int sentinel, void* workspace, int* pos_sorted, int* uniq_ids, int* seg_off, int* seg_of,
                                           int* n_uniq, void* stream) {
    return merge_sorted_lists(keys, world, len, key_stride, row_base, row_stride, sentinel, workspace, pos_sorted, uniq_ids, seg_off, seg_of,
                              n_uniq, nullptr, 0, 0, stream);
}

// the same merge with `n_entries` fixed-order sums (amid_reduce_entry_pack tables, as amid_reduce_partials_f32) in its first launch
extern "C" int amid_merge_sorted_lists_sum_i32(const int* keys, int world, int len, long long key_stride, int row_base, int row_stride,
                                               int sentinel, void* workspace, int* pos_sorted, int* uniq_ids, int* seg_off, int* seg_of,
                                               int* n_uniq, const void* entries_dev, int n_entries, int max_count, void* stream) {
    AMID_CHECK_ARG(entries_dev && n_entries > 0);
    return merge_sorted_lists(keys, world, len, key_stride, row_base, row_stride, sentinel, workspace, pos_sorted, uniq_ids, seg_off, seg_of,
                              n_uniq, entries_dev, n_entries, max_count, stream);
}
