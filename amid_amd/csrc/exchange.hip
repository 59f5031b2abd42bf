// Owner-bucketed sparse exchange (no counterpart in the reference, which is single-GPU: train_sr.py:473).
// A rank's segment-reduced table gradient -- ascending unique ids + one row each -- is split by owner = id % world into `world`
// packed chunks ([id rows | bmax gradient rows], the layout amid_merge_sorted_lists_i32 reads), one per destination rank of the
// all-to-all.  Every bucket stays ascending (the split is stable), unused slots carry the sentinel id and a zero row.
#include <hip/hip_runtime.h>

#include "amid_hip.h"
#include "common.h"

namespace amid {

constexpr int OB_BLOCK = 256;   // entries per block
constexpr int OB_MAXW = 16;     // ranks (amid_merge_sorted_lists_i32 merges up to 16 lists)

// block_counts[blk][o] = entries of block blk owned by o
__global__ __launch_bounds__(OB_BLOCK) void owner_count_kernel(const int* __restrict__ ids, const int* __restrict__ n_uniq, int cap, int world,
                                                               int* __restrict__ block_counts) {
    __shared__ int cnt[OB_MAXW];
    if (threadIdx.x < OB_MAXW) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int n = min(*n_uniq, cap), i = blockIdx.x * OB_BLOCK + threadIdx.x;
    if (i < n) atomicAdd(&cnt[ids[i] % world], 1);
    __syncthreads();
    if (threadIdx.x < OB_MAXW) block_counts[blockIdx.x * OB_MAXW + threadIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of every owner's column over the blocks (wave o scans owner o); counts[o] = the owner's total, counts[world] = 0
__global__ __launch_bounds__(64 * OB_MAXW) void owner_scan_kernel(int* __restrict__ block_counts, int nblk, int world, int* __restrict__ counts) {
    const int o = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) counts[world] = 0;                 // overflow flag of the fill
    if (o >= world) return;
    int run = 0;
    for (int base = 0; base < nblk; base += 64) {
        const int b = base + lane;
        const int v = b < nblk ? block_counts[b * OB_MAXW + o] : 0;
        int incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (b < nblk) block_counts[b * OB_MAXW + o] = run + incl - v;
        run += __shfl(incl, 63);
    }
    if (lane == 0) counts[o] = run;
}

// blocks [0, nblk): move the entries of one 256-entry block into their buckets; blocks [nblk, ...): sentinel ids + zero rows for
// the unused slots of every bucket
__global__ __launch_bounds__(OB_BLOCK) void owner_fill_kernel(const int* __restrict__ ids, const float* __restrict__ rows, const int* __restrict__ n_uniq,
                                                              int cap, int D, int world, int bmax, int sentinel, const int* __restrict__ block_off,
                                                              int nblk, float* __restrict__ out, long long chunk_floats, int id_rows,
                                                              int* __restrict__ counts) {
    const int q = D >> 2;                                    // float4 per row
    if ((int)blockIdx.x >= nblk) {                           // ---- padding: 8 slots per block
        const int per = (bmax + 7) / 8, pb = blockIdx.x - nblk;
        const int o = pb / per, s0 = (pb % per) * 8;
        const int have = min(counts[o], bmax);
        float* chunk = out + (long long)o * chunk_floats;
        for (int e = threadIdx.x; e < 8 * q; e += OB_BLOCK) {
            const int s = s0 + e / q;
            if (s < bmax && s >= have) {
                if (e % q == 0) ((int*)chunk)[s] = sentinel;
                ((float4*)(chunk + (long long)(id_rows + s) * D))[e % q] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        return;
    }
    __shared__ int wave_cnt[OB_BLOCK / 64][OB_MAXW];
    __shared__ int dst[OB_BLOCK];                            // destination row (chunk-relative, in rows of D floats from `out`) or -1
    const int n = min(*n_uniq, cap), i = blockIdx.x * OB_BLOCK + threadIdx.x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool live = i < n;
    const int id = live ? ids[i] : 0;
    const int own = live ? id % world : -1;
    int rank_in_wave = 0;
    for (int o = 0; o < world; ++o) {                        // stable rank of the entry among its wave's entries of the same owner
        const unsigned long long m = __ballot(own == o);
        if (own == o) rank_in_wave = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[w][o] = __popcll(m);
    }
    __syncthreads();
    int slot = -1;
    if (live) {
        int before = block_off[blockIdx.x * OB_MAXW + own];
        for (int ww = 0; ww < w; ++ww) before += wave_cnt[ww][own];
        slot = before + rank_in_wave;
        if (slot >= bmax) {                                  // the caller's bound was too small: flagged, never written out of range
            atomicOr(&counts[world], 1);
            slot = -1;
        } else {
            ((int*)(out + (long long)own * chunk_floats))[slot] = id;
        }
    }
    dst[threadIdx.x] = slot;
    __syncthreads();
    for (int e = threadIdx.x; e < OB_BLOCK * q; e += OB_BLOCK) {       // the block's rows, a row per q consecutive lanes
        const int r = e / q, s = dst[r];
        if (s < 0) continue;
        const int src = blockIdx.x * OB_BLOCK + r;
        const int o = ids[src] % world;
        ((float4*)(out + (long long)o * chunk_floats + (long long)(id_rows + s) * D))[e % q] = ((const float4*)(rows + (long long)src * D))[e % q];
    }
}

}  // namespace amid

using namespace amid;

extern "C" long long amid_owner_workspace_bytes(int cap) {
    if (cap <= 0) return 0;
    return ((long long)((cap + OB_BLOCK - 1) / OB_BLOCK) * OB_MAXW + 64) * (long long)sizeof(int);
}

extern "C" int amid_owner_count_i32(const int* uniq_ids, const int* n_uniq, int cap, int world, void* workspace, int* counts, void* stream) {
    AMID_CHECK_ARG(uniq_ids && n_uniq && workspace && counts && cap > 0 && world > 0 && world <= OB_MAXW);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (cap + OB_BLOCK - 1) / OB_BLOCK;
    owner_count_kernel<<<nblk, OB_BLOCK, 0, s>>>(uniq_ids, n_uniq, cap, world, (int*)workspace);
    owner_scan_kernel<<<1, 64 * OB_MAXW, 0, s>>>((int*)workspace, nblk, world, counts);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

extern "C" int amid_owner_buckets_f32(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int cap, int D, int world, int bmax,
                                      int sentinel, const void* workspace, float* out, long long chunk_floats, int id_rows, int* counts,
                                      void* stream) {
    AMID_CHECK_ARG(uniq_ids && uniq_rows && n_uniq && workspace && out && counts && cap > 0 && world > 0 && world <= OB_MAXW && bmax > 0);
    AMID_CHECK_ARG(D > 0 && (D % 4) == 0 && id_rows * (long long)D >= bmax && chunk_floats >= (long long)(id_rows + bmax) * D);
    const int nblk = (cap + OB_BLOCK - 1) / OB_BLOCK;
    const int pad_blocks = world * ((bmax + 7) / 8);
    owner_fill_kernel<<<nblk + pad_blocks, OB_BLOCK, 0, (hipStream_t)stream>>>(uniq_ids, uniq_rows, n_uniq, cap, D, world, bmax, sentinel,
                                                                              (const int*)workspace, nblk, out, chunk_floats, id_rows, counts);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
