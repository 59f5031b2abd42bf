// The four phases of the two-pass index sort (csrc/sortuniq.hip) as device functions of a virtual block index, so that they can run
// either as launches of their own (sortuniq.hip: long lists, on the side stream) or as extra workgroups in front of main-stream
// launches of the train step (the sort then costs no launch, no fork and no join: sasrec_strip.hip, adam.hip).
#pragma once
#include "common.h"

namespace amid {

constexpr int SORT_THREADS = 256;
constexpr int SORT_ITEMS = 8;                       // rounds of 64 keys per wave
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS; // 2048 keys per block
constexpr int SORT_WAVES = SORT_THREADS / 64;

// ---------------------------------------------------------------------------------------------------------------------------------
// Two-pass sort in FOUR launches (keys below 2^24: every table up to 16.7 M rows).  Digits of ceil(bits / 2) and floor(bits / 2) bits
// (cfg 2: 10 + 10, cfg 5: 12 + 12 -- up to 4096 bins); tiles of 2048 keys, supertiles of 16 tiles.  For each pass p two tables say
// where a tile's keys of a bin go: stot_p[supertile][bin] and counts_p[tile][bin]; a scatter block sweeps the stot_p rows (all of
// them: the bin totals, whose exclusive scan is the bin's base; the earlier supertiles': its share) and the counts_p rows of the
// earlier tiles of its own supertile -- at most n_super + 15 coalesced, cached rows; nothing waits for another block, nothing is
// scanned by a single block.
//   launch 1  os_count     counts_0 (plain stores), stot_0 (atomics); zeroes counts_1 / stot_1, launch 4's status words and the
//                          stot_0 copy of the next call
//   launch 2  os_scatter   stable scatter by digit 0; every key also bumps counts_1 / stot_1 of the tile its DESTINATION lies in
//                          (pass 1's tiles are contiguous slices of this pass's output)
//   launch 3  os_scatter   stable scatter by digit 1 -> sorted keys + positions
//   launch 4  os_heads     run heads: per-tile count, a wave-parallel look-back over the earlier tiles' status words gives the tile's
//                          first run index (tile ids are handed out by an atomic counter: a tile only waits for tiles that already
//                          run); writes uniq_ids / seg_off / seg_of / n_uniq; flips the stot_0 copy
// stot_0 is the one table that must be zero when a call starts: its two copies sit at a FIXED place at the head of the workspace
// (whatever n_idx the workspace is used with), zero-filled once with the workspace; calls alternate between them and launch 1
// re-zeroes the copy the next call will use, over the extent the previous call recorded for it (state[4 + copy]).
constexpr int OS_BINS_MAX = 4096;
constexpr int OS_STATE_INTS = 64;                 // [2] tile counter of launch 4, [3] stot_0 copy of this call, [4 + c] dirty extent of copy c
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

// The phases' LDS, handed in by the caller: launches of their own keep it as a static allocation (sortuniq.hip); riders alias the head of the
// host kernel's dynamic allocation (the rider workgroups use nothing else of it) -- so a kernel that rides the 4 096-bin build (tables of
// 2^20 rows and more: cfg 5's 10 M) does not add 48 KB of static LDS to every workgroup of its launch (round 6).
// (the count phases: int hist[bins])
template <int BINS> struct SortScatterLds {             // os_scatter_block
    unsigned short woff[SORT_WAVES][BINS];              // per-wave digit counts, then running offsets inside the tile's bin
    int tile_base[BINS];                                // output position of the tile's first key of every bin
    int wsum[SORT_WAVES]; int pc[SORT_WAVES]; int pn[SORT_WAVES];
};
struct SortHeadsLds { int tile_s, excl_s; int wsum[SORT_WAVES]; };                           // os_heads_block

// BINS: LDS is sized for 1024 bins when both digits have at most 10 bits (every table below 2^20 rows: 12 KB in the scatter) -- small
// enough to share a CU with a one-workgroup-per-CU kernel of the main stream (the fused forward holds 144 of the 160 KB).
// keys of the first pass as a function of the position: the list itself, or -- the one-launch step head, where the list is written by other
// workgroups of the same launch -- straight from the batch's packed image (adam.hip PoolBatch)
struct ArrayKeys { const int* __restrict__ p; __device__ __forceinline__ int operator()(int k) const { return p[k]; } };
template <int BINS, class Keys>
__device__ __forceinline__ void os_count_block(const int blk, const int nblk, const Keys keys, OsGeom g, int* __restrict__ state,
                                                                int* __restrict__ counts0, int* __restrict__ stot0, int stot0_copy, int n_stot0,
                                                                int* __restrict__ counts1, long long n_counts1, int* __restrict__ stot1,
                                                                int n_stot1, int* __restrict__ hstatus, int* const hist0 /* LDS [BINS] */) {
    const int bins0 = 1 << g.bits;
    for (int d = threadIdx.x; d < bins0; d += SORT_THREADS) hist0[d] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const int base = blk * SORT_TILE + w * (64 * SORT_ITEMS);
    const unsigned mask0 = (unsigned)bins0 - 1u;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = base + i * 64 + lane;
        const bool valid = k < g.n;
        const unsigned d0 = valid ? (((unsigned)keys(k) >> g.shift) & mask0) : 0u;
        const unsigned long long peers = match_bits(d0, valid, g.bits);
        if (valid && (__ffsll((long long)peers) - 1) == lane) atomicAdd(&hist0[d0], __popcll(peers));
    }
    __syncthreads();
    // stot_0 is accumulated with atomics, so it must be zero when a call starts: there are two copies, used by alternate calls
    // (state[3], flipped by the last tile of launch 4); this launch fills one and zeroes the other for the next call
    const int par = state[3] & 1;
    int* __restrict__ st_use = stot0 + (long long)par * stot0_copy;
    int* __restrict__ st_zero = stot0 + (long long)(1 - par) * stot0_copy;
    const int sup = blk / OS_SUPER;
    for (int d = threadIdx.x; d < bins0; d += SORT_THREADS) {
        const int c = hist0[d];
        counts0[(long long)blk * bins0 + d] = c;
        if (c) atomicAdd(&st_use[(long long)sup * bins0 + d], c);
    }
    // The other copy is dirty over what the PREVIOUS call (calls alternate between the copies) accumulated into it -- that call's
    // n_stot0, which is not this call's when one workspace serves lists of different lengths (a plan's compact list and its full
    // autograd list): every call records its extent in state[4 + copy], and the next one zeroes exactly that.
    const int ext_other = state[4 + (1 - par)];
    for (int i = blk * SORT_THREADS + threadIdx.x; i < ext_other; i += nblk * SORT_THREADS) st_zero[i] = 0;
    if (blk == 0 && threadIdx.x == 0) state[4 + par] = n_stot0;       // (no block of this launch reads this word)
    // housekeeping for the later launches of this call
    for (long long i = (long long)blk * SORT_THREADS + threadIdx.x; i < n_counts1; i += (long long)nblk * SORT_THREADS) counts1[i] = 0;
    for (int i = blk * SORT_THREADS + threadIdx.x; i < n_stot1; i += nblk * SORT_THREADS) stot1[i] = 0;
    for (int i = blk * SORT_THREADS + threadIdx.x; i < g.ntiles; i += nblk * SORT_THREADS) hstatus[i] = 0;
    if (blk == 0 && threadIdx.x == 0) state[2] = 0;
}

// pass 1's per-tile digit counts as a phase of its own (the riders' schedule: there the counting costs no launch, and the global
// atomics with which os_scatter_block<true> counts on the fly are most of that launch's time at cfg 2 -- 5.8 k scattered atomics,
// 23 of its 34 us).  counts_1 by plain stores, stot_1 (zeroed by os_count_block) by one atomic per tile and non-empty bin.
template <int BINS>
__device__ __forceinline__ void os_count1_block(const int blk, const int* __restrict__ keys, OsGeom g, int* __restrict__ counts1,
                                                int* __restrict__ stot1, int* const hist1 /* LDS [BINS] */) {
    const int bins = 1 << g.bits;
    for (int d = threadIdx.x; d < bins; d += SORT_THREADS) hist1[d] = 0;
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const int base = blk * SORT_TILE + w * (64 * SORT_ITEMS);
    const unsigned mask = (unsigned)bins - 1u;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const int k = base + i * 64 + lane;
        const bool valid = k < g.n;
        const unsigned d = valid ? (((unsigned)keys[k] >> g.shift) & mask) : 0u;
        const unsigned long long peers = match_bits(d, valid, g.bits);
        if (valid && (__ffsll((long long)peers) - 1) == lane) atomicAdd(&hist1[d], __popcll(peers));
    }
    __syncthreads();
    const int sup = blk / OS_SUPER;
    for (int d = threadIdx.x; d < bins; d += SORT_THREADS) {
        const int c = hist1[d];
        counts1[(long long)blk * bins + d] = c;
        if (c) atomicAdd(&stot1[(long long)sup * bins + d], c);
    }
}

// stable scatter of one pass.  COUNT_NEXT (pass 0): also the next pass's counts / stot (global atomics).
template <bool COUNT_NEXT, int BINS>
__device__ __forceinline__ void os_scatter_block(const int blk, const int nblk, const int* __restrict__ keys_in, const int* __restrict__ vals_in,
                                                                  int* __restrict__ keys_out, int* __restrict__ vals_out, OsGeom g,
                                                                  const int* __restrict__ counts, const int* __restrict__ stot, int nsup,
                                                                  int* __restrict__ counts_next, int* __restrict__ stot_next, SortScatterLds<BINS>& lds) {
    unsigned short (&woff)[SORT_WAVES][BINS] = lds.woff;
    int* const tile_base = lds.tile_base;
    int* const wsum = lds.wsum;
    const int bins = 1 << g.bits;
    const unsigned mask = (unsigned)bins - 1u;
    const int w = wave_id(), lane = lane_id();
    for (int i = threadIdx.x; i < SORT_WAVES * BINS / 2; i += SORT_THREADS) ((unsigned*)&woff[0][0])[i] = 0u;
    __syncthreads();
    const int kb = blk * SORT_TILE + w * (64 * SORT_ITEMS);
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
    const int sup = blk / OS_SUPER;
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
            for (int t = sup * OS_SUPER; t < blk; ++t) add_row(counts + (long long)t * bins, true, false);
        } else {
#pragma unroll
            for (int k = 0; k < PERMAX; ++k) {
                if (k < per) {
                    for (int s2 = 0; s2 < nsup; ++s2) { const int a = stot[(long long)s2 * bins + d0 + k]; tot[k] += a; if (s2 < sup) pre[k] += a; }
                    for (int t = sup * OS_SUPER; t < blk; ++t) pre[k] += counts[(long long)t * bins + d0 + k];
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
        int* const pc = lds.pc; int* const pn = lds.pn;
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
}

// run heads of the sorted keys in one launch: thread t of a tile owns 8 consecutive entries
constexpr unsigned OS_ST_AGG = 1u << 30, OS_ST_PRE = 2u << 30, OS_ST_VAL = (1u << 30) - 1u;
__device__ __forceinline__ void os_heads_block(const int* __restrict__ keys, int n, int ntiles, int* __restrict__ state,
                                                                unsigned* __restrict__ hstatus, int* __restrict__ n_uniq,
                                                                int* __restrict__ uniq_ids, int* __restrict__ seg_off, int* __restrict__ seg_of,
                                                                SortHeadsLds& lds, long long chain_base = -1) {
    int& tile_s = lds.tile_s; int& excl_s = lds.excl_s;
    int* const wsum = lds.wsum;
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
            if (tile == ntiles - 1) {
                *n_uniq = excl + total; seg_off[excl + total] = n; state[3] ^= 1;       // (the next call's stot_0 copy)
                if (chain_base >= 0) state[7] = (int)((unsigned)chain_base + 4u * (unsigned)ntiles);      // (sort_chain_block: the next call's barrier base)
            }
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


// Everything the four phases need, packed on the host (amid_sort_plan_pack) and handed to kernels by value.
struct SortPlan {
    const int* idx; const int* rows;              // keys; optional payload (pos_sorted then holds rows[i] instead of i)
    int n, nblk, nsup;
    OsGeom g0, g1;
    int* state; int* stot0; int* stot1; int* counts0; int* counts1; unsigned* hstatus;
    int* keys0; int* vals0; int* keys1;
    int* pos_sorted; int* uniq_ids; int* seg_off; int* seg_of; int* n_uniq;
    long long n_counts1; int n_stot1, n_zero_a, stot0_copy;       // stot0: two copies of stot0_copy ints (see os_count_block)
};

// phase PHASE (1 .. 5) of the sort as block `blk` (< sp.nblk) of a 256-thread workgroup, chosen at compile time, on `lds_raw`: LDS of at
// least sort_phase_lds_bytes<PHASE>() bytes that the rider workgroup owns (16-byte aligned).  The bin count follows the plan: 1 024 bins for
// digits of at most 10 bits (every table below 2^20 rows), 4 096 otherwise (keys below 2^24).
// (Kernels with dynamic LDS used to carry static LDS for ONE phase each: with the static LDS of two phases beside a dynamic allocation hipcc 7.2
// dies in instruction selection -- "Illegal instruction detected: Operand has incorrect register class V_CMP_NE_U32_e32 0, $src_shared_base";
// they alias their dynamic allocation now.)
template <int PHASE> constexpr size_t sort_phase_lds_bytes() {
    return PHASE == 5 ? sizeof(SortHeadsLds) : (PHASE == 2 || PHASE == 4) ? sizeof(SortScatterLds<OS_BINS_MAX>) : sizeof(int) * OS_BINS_MAX;
}
template <int BINS, int PHASE>
__device__ __forceinline__ void sort_phase_bins(const SortPlan& sp, int blk, void* lds_raw) {
    // the riders' schedule, five phases: count 0, scatter 0, count 1, scatter 1, run heads
    if constexpr (PHASE == 1)
        os_count_block<BINS>(blk, sp.nblk, ArrayKeys{sp.idx}, sp.g0, sp.state, sp.counts0, sp.stot0, sp.stot0_copy, sp.n_zero_a, sp.counts1, sp.n_counts1, sp.stot1,
                             sp.n_stot1, (int*)sp.hstatus, (int*)lds_raw);
    else if constexpr (PHASE == 2)
        os_scatter_block<false, BINS>(blk, sp.nblk, sp.idx, sp.rows, sp.keys0, sp.vals0, sp.g0, sp.counts0,
                                      sp.stot0 + (long long)(sp.state[3] & 1) * sp.stot0_copy, sp.nsup, nullptr, nullptr, *(SortScatterLds<BINS>*)lds_raw);
    else if constexpr (PHASE == 3)
        os_count1_block<BINS>(blk, sp.keys0, sp.g1, sp.counts1, sp.stot1, (int*)lds_raw);
    else if constexpr (PHASE == 4)
        os_scatter_block<false, BINS>(blk, sp.nblk, sp.keys0, sp.vals0, sp.keys1, sp.pos_sorted, sp.g1, sp.counts1, sp.stot1, sp.nsup, nullptr, nullptr,
                                      *(SortScatterLds<BINS>*)lds_raw);
    else
        os_heads_block(sp.keys1, sp.n, sp.nblk, sp.state, sp.hstatus, sp.n_uniq, sp.uniq_ids, sp.seg_off, sp.seg_of, *(SortHeadsLds*)lds_raw);
}
template <int PHASE>
__device__ __forceinline__ void sort_phase_ct(const SortPlan& sp, int blk, void* lds_raw) {
    if constexpr (PHASE == 5) sort_phase_bins<1024, 5>(sp, blk, lds_raw);
    else if (sp.g0.bits <= 10) sort_phase_bins<1024, PHASE>(sp, blk, lds_raw);        // (block-uniform; g0 has the wider digit)
    else sort_phase_bins<OS_BINS_MAX, PHASE>(sp, blk, lds_raw);
}

// ALL FIVE phases in ONE launch (round 6): the plan's nblk rider workgroups run the phases back to back and meet at a barrier of their own
// between them -- for the steps that have no five launches to ride in (the one-launch backward at T <= 32, BERT4Rec, the comp modules), whose
// sort ran as twelve small launches on a side stream (a fork and a join per step: ~ 6 us of idle main stream each inside a replayed graph).
// The riders are the launch's FIRST workgroups and there are at most SORT_CHAIN_MAX_BLOCKS of them: they are dispatched before any other
// workgroup of the launch and need 12 KB of LDS each, so all of them are resident when the first one waits (the same assumption the
// look-back of os_heads_block makes about earlier tiles).  Barrier: state[6] counts arrivals for ever (wrap-safe unsigned differences);
// state[7] holds its value when the call began -- written by the call before (the last tile of the heads phase, when every rider is past the
// last barrier); both start at zero with the workspace.  Memory: every rider's stores are released at agent scope before it arrives and
// the CU's caches are invalidated after it leaves (a rider reads what riders on other XCDs wrote in the phase before; a build without the
// two fences did not come back on the hardware).
constexpr int SORT_CHAIN_MAX_BLOCKS = 64, SORT_CHAIN_PHASE = 6;
__device__ __forceinline__ void sort_chain_barrier(int* __restrict__ state, unsigned base, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add((unsigned*)&state[6], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while ((__hip_atomic_load((unsigned*)&state[6], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - base) < target) __builtin_amdgcn_s_sleep(2);
        __threadfence();
    }
    __syncthreads();
}
// lds_raw: sizeof(SortScatterLds<1024>) bytes (1 024-bin plans only: sp.g0.bits <= 10)
__device__ __forceinline__ void sort_chain_block(const SortPlan& sp, int blk, void* lds_raw) {
    const unsigned base = (unsigned)__hip_atomic_load(&sp.state[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned n = (unsigned)sp.nblk;
    sort_phase_bins<1024, 1>(sp, blk, lds_raw);
    sort_chain_barrier(sp.state, base, n);
    sort_phase_bins<1024, 2>(sp, blk, lds_raw);
    sort_chain_barrier(sp.state, base, 2 * n);
    sort_phase_bins<1024, 3>(sp, blk, lds_raw);
    sort_chain_barrier(sp.state, base, 3 * n);
    sort_phase_bins<1024, 4>(sp, blk, lds_raw);
    sort_chain_barrier(sp.state, base, 4 * n);
    // (the heads phase: the tile that takes the LAST ticket also records the barrier counter's value for the next call)
    os_heads_block(sp.keys1, sp.n, sp.nblk, sp.state, sp.hstatus, sp.n_uniq, sp.uniq_ids, sp.seg_off, sp.seg_of, *(SortHeadsLds*)lds_raw, (long long)base);
}

// riders: a launch of the train step with `plan.nblk` extra 256-thread workgroups IN FRONT of its own (blockIdx < nblk) that run one
// of the FIVE phases of the step's index sort (sort_phase_ct; keys below 2^24); phase 0 = no rider
struct SortRider { SortPlan plan; int phase; };
__device__ __forceinline__ int rider_blocks(const SortRider& r) { return r.phase ? r.plan.nblk : 0; }
static inline int rider_blocks_host(const SortRider& r) { return r.phase ? r.plan.nblk : 0; }

}  // namespace amid
