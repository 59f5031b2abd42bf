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

namespace amid {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;                       // rounds of 64 keys per wave
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS; // 2048 keys per block
constexpr int SORT_WAVES = SORT_THREADS / 64;

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

// ---------------------------------------------------------------------------------------------------------------------------------
// Two-pass sort in FOUR launches (keys below 2^24: every table up to 16.7 M rows).  Digits of ceil(bits / 2) and floor(bits / 2) bits
// (cfg 2: 10 + 10, cfg 5: 12 + 12 -- up to 4096 bins); tiles of 2048 keys, supertiles of 16 tiles.  For each pass p two tables say
// where a tile's keys of a bin go: stot_p[supertile][bin] and counts_p[tile][bin]; a scatter block sweeps the stot_p rows (all of
// them: the bin totals, whose exclusive scan is the bin's base; the earlier supertiles': its share) and the counts_p rows of the
// earlier tiles of its own supertile -- at most n_super + 15 coalesced, cached rows; nothing waits for another block, nothing is
// scanned by a single block.
//   launch 1  os_count     counts_0 (plain stores), stot_0 (atomics); zeroes counts_1 / stot_1 and launch 4's status words
//   launch 2  os_scatter   stable scatter by digit 0; every key also bumps counts_1 / stot_1 of the tile its DESTINATION lies in
//                          (pass 1's tiles are contiguous slices of this pass's output)
//   launch 3  os_scatter   stable scatter by digit 1 -> sorted keys + positions; zeroes stot_0 for the next call
//   launch 4  os_heads     run heads: per-tile count, a wave-parallel look-back over the earlier tiles' status words gives the tile's
//                          first run index (tile ids are handed out by an atomic counter: a tile only waits for tiles that already
//                          run); writes uniq_ids / seg_off / seg_of / n_uniq
// stot_0 is the one table that must be zero when a call starts: it sits at a FIXED place at the head of the workspace (whatever
// n_idx the workspace is used with), is zero-filled once with the workspace and re-zeroed by launch 3 of every call.
constexpr int OS_BINS_MAX = 4096;
constexpr int OS_STATE_INTS = 64;                 // [2] tile counter of launch 4
constexpr int OS_SUPER = 16;                      // tiles per supertile
constexpr int OS_SUPER_MAX = 256;                 // supertiles the fixed stot_0 area holds (4096 tiles = 8.4 M indices; beyond: 8-bit passes)

struct OsGeom {
    int n, ntiles;
    int shift, bits;                              // this pass's digit
    int next_shift, next_bits;                    // the other pass's digit
};

__device__ __forceinline__ unsigned long long match_bits(unsigned d, bool valid, int bits) {
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const unsigned long long m = __ballot((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
    }
    return peers;
}

// BINS: LDS is sized for 1024 bins when both digits have at most 10 bits (every table below 2^20 rows: 12 KB in the scatter) -- small
// enough to share a CU with a one-workgroup-per-CU kernel of the main stream (the fused forward holds 144 of the 160 KB).
template <int BINS>
__global__ __launch_bounds__(SORT_THREADS) void os_count_kernel(const int* __restrict__ keys, OsGeom g, int* __restrict__ state,
                                                                int* __restrict__ counts0, int* __restrict__ stot0, int* __restrict__ counts1,
                                                                long long n_counts1, int* __restrict__ stot1, int n_stot1,
                                                                int* __restrict__ hstatus) {
    __shared__ int hist0[BINS];
    const int bins0 = 1 << g.bits;
    for (int d = threadIdx.x; d < bins0; d += SORT_THREADS) hist0[d] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const int base = blockIdx.x * SORT_TILE + w * (64 * SORT_ITEMS);
    const unsigned mask0 = (unsigned)bins0 - 1u;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = base + i * 64 + lane;
        const bool valid = k < g.n;
        const unsigned d0 = valid ? (((unsigned)keys[k] >> g.shift) & mask0) : 0u;
        const unsigned long long peers = match_bits(d0, valid, g.bits);
        if (valid && (__ffsll((long long)peers) - 1) == lane) atomicAdd(&hist0[d0], __popcll(peers));
    }
    __syncthreads();
    const int sup = blockIdx.x / OS_SUPER;
    for (int d = threadIdx.x; d < bins0; d += SORT_THREADS) {
        const int c = hist0[d];
        counts0[(long long)blockIdx.x * bins0 + d] = c;
        if (c) atomicAdd(&stot0[(long long)sup * bins0 + d], c);
    }
    // housekeeping for the later launches of this call
    for (long long i = (long long)blockIdx.x * SORT_THREADS + threadIdx.x; i < n_counts1; i += (long long)gridDim.x * SORT_THREADS) counts1[i] = 0;
    for (int i = blockIdx.x * SORT_THREADS + threadIdx.x; i < n_stot1; i += gridDim.x * SORT_THREADS) stot1[i] = 0;
    for (int i = blockIdx.x * SORT_THREADS + threadIdx.x; i < g.ntiles; i += gridDim.x * SORT_THREADS) hstatus[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) state[2] = 0;
}

// stable scatter of one pass.  COUNT_NEXT (pass 0): also the next pass's counts / stot (global atomics).  !COUNT_NEXT (pass 1):
// zeroes stot_0 (zero_a), the one table that must be zero when the next call starts.
template <bool COUNT_NEXT, int BINS>
__global__ __launch_bounds__(SORT_THREADS) void os_scatter_kernel(const int* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                                  int* __restrict__ keys_out, int* __restrict__ vals_out, OsGeom g,
                                                                  const int* __restrict__ counts, const int* __restrict__ stot, int nsup,
                                                                  int* __restrict__ counts_next, int* __restrict__ stot_next,
                                                                  int* __restrict__ zero_a, int n_zero_a) {
    __shared__ unsigned short woff[SORT_WAVES][BINS];           // per-wave digit counts, then running offsets inside the tile's bin
    __shared__ int tile_base[BINS];                             // output position of the tile's first key of every bin
    __shared__ int wsum[SORT_WAVES];
    const int bins = 1 << g.bits;
    const unsigned mask = (unsigned)bins - 1u;
    const int w = wave_id(), lane = lane_id();
    for (int i = threadIdx.x; i < SORT_WAVES * BINS / 2; i += SORT_THREADS) ((unsigned*)&woff[0][0])[i] = 0u;
    __syncthreads();
    const int kb = blockIdx.x * SORT_TILE + w * (64 * SORT_ITEMS);
    int key[SORT_ITEMS];
    unsigned long long peers[SORT_ITEMS];
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = kb + i * 64 + lane;
        const bool valid = k < g.n;
        key[i] = valid ? keys_in[k] : 0;
        const unsigned d = ((unsigned)key[i] >> g.shift) & mask;
        peers[i] = match_bits(d, valid, g.bits);
        // one leader per digit group and round, rounds in program order: no two lanes ever update the same counter at once
        if (valid && (__ffsll((long long)peers[i]) - 1) == lane) woff[w][d] = (unsigned short)(woff[w][d] + __popcll(peers[i]));
        __builtin_amdgcn_wave_barrier();
    }
    // where the tile's keys of every bin start: thread t owns the bins [t * per, (t + 1) * per).  One sweep over the supertile rows
    // gives both the bin totals (all rows: for the scan over the bins) and the earlier supertiles' share (rows below this tile's
    // supertile); the earlier tiles of the own supertile follow.  A thread reads its bins of a row as int4s (a wave covers a
    // contiguous KB), two rows in flight.
    constexpr int PERMAX = BINS / SORT_THREADS;
    const int per = bins / SORT_THREADS > 0 ? bins / SORT_THREADS : 1;
    const int d0 = threadIdx.x * per;
    const int sup = blockIdx.x / OS_SUPER;
    int tot[PERMAX], pre[PERMAX];
#pragma unroll
    for (int k = 0; k < PERMAX; ++k) { tot[k] = 0; pre[k] = 0; }
    if (d0 < bins) {
        if ((per & 3) == 0) {
            auto add_row = [&](const int* __restrict__ rowp, bool early, bool total) {
#pragma unroll
                for (int k = 0; k < PERMAX; k += 4) {
                    if (k < per) {
                        const int4 a = *(const int4*)(rowp + d0 + k);
                        if (total) { tot[k] += a.x; tot[k + 1] += a.y; tot[k + 2] += a.z; tot[k + 3] += a.w; }
                        if (early) { pre[k] += a.x; pre[k + 1] += a.y; pre[k + 2] += a.z; pre[k + 3] += a.w; }
                    }
                }
            };
            for (int s2 = 0; s2 < nsup; ++s2) add_row(stot + (long long)s2 * bins, s2 < sup, true);
            for (int t = sup * OS_SUPER; t < (int)blockIdx.x; ++t) add_row(counts + (long long)t * bins, true, false);
        } else {
#pragma unroll
            for (int k = 0; k < PERMAX; ++k) {
                if (k < per) {
                    for (int s2 = 0; s2 < nsup; ++s2) { const int a = stot[(long long)s2 * bins + d0 + k]; tot[k] += a; if (s2 < sup) pre[k] += a; }
                    for (int t = sup * OS_SUPER; t < (int)blockIdx.x; ++t) pre[k] += counts[(long long)t * bins + d0 + k];
                }
            }
        }
    }
    int mine = 0;
#pragma unroll
    for (int k = 0; k < PERMAX; ++k) mine += tot[k];
    int x = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();                                             // (also: every wave's woff counts are complete)
    if (d0 < bins) {
        int run = x - mine;
        for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
        for (int k = 0; k < PERMAX; ++k) {
            if (k < per) {
                const int d = d0 + k;
                tile_base[d] = run + pre[k];                     // base(d) + the earlier tiles' keys of the bin
                run += tot[k];
                unsigned wrun = 0;
#pragma unroll
                for (int q = 0; q < SORT_WAVES; ++q) {
                    const unsigned c = woff[q][d];
                    woff[q][d] = (unsigned short)wrun;
                    wrun += c;
                }
            }
        }
    }
    __syncthreads();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned nmask = (1u << g.next_bits) - 1u;
    const int nbins = 1 << g.next_bits;
    int pend_cell = -1, pend_n = 0;                 // wave-uniform: the leading cell of the last rounds and its carried count
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = kb + i * 64 + lane;
        const bool valid = k < g.n;
        const unsigned d = ((unsigned)key[i] >> g.shift) & mask;
        int dst = 0;
        if (valid) dst = tile_base[d] + woff[w][d] + __popcll(peers[i] & lt);
        __builtin_amdgcn_wave_barrier();            // every lane has read the offset before the leader bumps it
        if (valid && (__ffsll((long long)peers[i]) - 1) == lane) woff[w][d] = (unsigned short)(woff[w][d] + __popcll(peers[i]));
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            keys_out[dst] = key[i];
            vals_out[dst] = vals_in ? vals_in[k] : k;
        }
        if (COUNT_NEXT) {
            // counts_next[dst / tile][digit 1] (+ the supertile's stot_next).  Equal keys share digit 1 and land next to each other, and
            // the pad id is 80-90 % of a real batch: the lanes that share the first lane's cell are counted together and the count is
            // CARRIED across the rounds while the leading cell stays the same (same-address atomics serialise in L2: one per wave
            // and cell instead of one per round); the other lanes add one each.
            const int tile1 = dst / SORT_TILE;
            const int d1 = (int)(((unsigned)key[i] >> g.next_shift) & nmask);
            const int cell = valid ? tile1 * nbins + d1 : -1;
            const int lead_cell = __builtin_amdgcn_readfirstlane(cell);
            const unsigned long long same = __ballot(valid && cell == lead_cell);
            if (lead_cell >= 0) {
                if (lead_cell == pend_cell) pend_n += __popcll(same);
                else {
                    if (pend_n && lane == 0) {
                        atomicAdd(&counts_next[pend_cell], pend_n);
                        atomicAdd(&stot_next[(long long)(pend_cell / nbins / OS_SUPER) * nbins + (pend_cell % nbins)], pend_n);
                    }
                    pend_cell = lead_cell;
                    pend_n = __popcll(same);
                }
            }
            if (valid && cell != lead_cell) {
                atomicAdd(&counts_next[cell], 1);
                atomicAdd(&stot_next[(long long)(tile1 / OS_SUPER) * nbins + d1], 1);
            }
        }
    }
    if (COUNT_NEXT) {           // the waves' carried counts: equal cells of the four waves merged, then one atomic each
        __shared__ int pc[SORT_WAVES], pn[SORT_WAVES];
        if (lane == 0) { pc[w] = pend_cell; pn[w] = pend_n; }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int a = 0; a < SORT_WAVES; ++a) {
                int n_a = pn[a];
                if (!n_a) continue;
#pragma unroll
                for (int b = a + 1; b < SORT_WAVES; ++b)
                    if (pn[b] && pc[b] == pc[a]) { n_a += pn[b]; pn[b] = 0; }
                atomicAdd(&counts_next[pc[a]], n_a);
                atomicAdd(&stot_next[(long long)(pc[a] / nbins / OS_SUPER) * nbins + (pc[a] % nbins)], n_a);
            }
        }
    }
    if (!COUNT_NEXT) {
        for (int i2 = blockIdx.x * SORT_THREADS + threadIdx.x; i2 < n_zero_a; i2 += gridDim.x * SORT_THREADS) zero_a[i2] = 0;
    }
}

// run heads of the sorted keys in one launch: thread t of a tile owns 8 consecutive entries
constexpr unsigned OS_ST_AGG = 1u << 30, OS_ST_PRE = 2u << 30, OS_ST_VAL = (1u << 30) - 1u;
__global__ __launch_bounds__(SORT_THREADS) void os_heads_kernel(const int* __restrict__ keys, int n, int ntiles, int* __restrict__ state,
                                                                unsigned* __restrict__ hstatus, int* __restrict__ n_uniq,
                                                                int* __restrict__ uniq_ids, int* __restrict__ seg_off, int* __restrict__ seg_of) {
    __shared__ int tile_s, excl_s;
    __shared__ int wsum[SORT_WAVES];
    if (threadIdx.x == 0) tile_s = atomicAdd(&state[2], 1);
    __syncthreads();
    const int tile = tile_s;
    const int lane = lane_id(), w = wave_id();
    const int i0 = tile * SORT_TILE + threadIdx.x * SORT_ITEMS;
    int k[SORT_ITEMS];
    int prev = (i0 > 0 && i0 < n) ? keys[i0 - 1] : 0;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int i = i0 + j;
        k[j] = i < n ? keys[i] : 0;
        cnt += (i < n && (i == 0 || k[j] != (j ? k[j - 1] : prev))) ? 1 : 0;
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
#pragma unroll
    for (int q = 0; q < SORT_WAVES; ++q) { if (q < w) woff += wsum[q]; total += wsum[q]; }
    if (w == 0) {                                       // wave 0: publish the tile's count, look back, publish the inclusive prefix
        int excl = 0;
        if (tile > 0) {
            if (lane == 0) __hip_atomic_store(&hstatus[tile], OS_ST_AGG | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int t = tile - 1;
            while (true) {
                const int idx = t - lane;
                const unsigned s = idx >= 0 ? __hip_atomic_load(&hstatus[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_ST_PRE;
                const unsigned long long nr = __ballot((s >> 30) == 0u), pre = __ballot((s >> 30) == 2u);
                const int first_pre = pre ? __ffsll((long long)pre) - 1 : 64;
                const int first_nr = nr ? __ffsll((long long)nr) - 1 : 64;
                if (first_nr < first_pre) { __builtin_amdgcn_s_sleep(2); continue; }     // a tile before the nearest prefix is not ready
                const int upto = first_pre < 64 ? first_pre : 63;
                int v = lane <= upto ? (int)(s & OS_ST_VAL) : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                excl += v;
                if (first_pre < 64) break;
                t -= 64;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&hstatus[tile], OS_ST_PRE | (unsigned)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            excl_s = excl;
            if (tile == ntiles - 1) { *n_uniq = excl + total; seg_off[excl + total] = n; }
        }
    }
    __syncthreads();
    int u = excl_s + woff + x - cnt - 1;                // run index of the entry before this thread's span
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const int i = i0 + j;
        if (i < n) {
            if (i == 0 || k[j] != (j ? k[j - 1] : prev)) { ++u; uniq_ids[u] = k[j]; seg_off[u] = i; }
            seg_of[i] = u;
        }
    }
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
static inline size_t os_head_bytes() { return align256(OS_STATE_INTS * 4) + align256((size_t)OS_SUPER_MAX * OS_BINS_MAX * 4); }
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
    if (bits <= 24 && n_idx >= g_four_launch_min && os_nsuper(nblk) <= (size_t)OS_SUPER_MAX) {      // four launches (see os_count_kernel)
        char* os = ws + 4 * kb + align256((size_t)256 * nblk * 4) + align256(((size_t)n_idx + 255) / 256 * 4);
        const size_t nsup = os_nsuper(nblk);
        int* state = (int*)ws_head;
        int* stot0 = (int*)(ws_head + align256(OS_STATE_INTS * 4));
        int* stot1 = (int*)os;
        int* counts0 = (int*)((char*)stot1 + align256(nsup * OS_BINS_MAX * 4));
        int* counts1 = (int*)((char*)counts0 + align256((size_t)nblk * OS_BINS_MAX * 4));
        unsigned* hstatus = (unsigned*)((char*)counts1 + align256((size_t)nblk * OS_BINS_MAX * 4));
        if (bits < 2) bits = 2;
        const int b0 = (bits + 1) / 2, b1 = bits - b0;
        OsGeom g0{n_idx, nblk, 0, b0, b0, b1}, g1{n_idx, nblk, b0, b1, 0, b0};
#define AMID_OS_LAUNCH(BINS)                                                                                                              \
        os_count_kernel<BINS><<<nblk, SORT_THREADS, 0, s>>>(idx, g0, state, counts0, stot0, counts1, (long long)nblk << b1, stot1,               \
                                                            (int)(nsup << b1), (int*)hstatus);                                            \
        os_scatter_kernel<true, BINS><<<nblk, SORT_THREADS, 0, s>>>(idx, rows, keys[0], vals[0], g0, counts0, stot0, (int)nsup, counts1, stot1, \
                                                                    nullptr, 0);                                                          \
        os_scatter_kernel<false, BINS><<<nblk, SORT_THREADS, 0, s>>>(keys[0], vals[0], keys[1], pos_sorted, g1, counts1, stot1, (int)nsup,      \
                                                                     nullptr, nullptr, stot0, (int)(nsup << b0));
        if (b0 <= 10) { AMID_OS_LAUNCH(1024) } else { AMID_OS_LAUNCH(4096) }
#undef AMID_OS_LAUNCH
        os_heads_kernel<<<nblk, SORT_THREADS, 0, s>>>(keys[1], n_idx, nblk, state, hstatus, n_uniq, uniq_ids, seg_off, seg_of);
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
