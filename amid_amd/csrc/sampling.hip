// Negative sampling on the device (next-3 of SURVEY.md 8(f); reference: dataset_seq.py:188 / :206 and :198 / :215 --
// `random.sample(item_pool_d - set(own sequence), k)` per sample inside __getitem__, the 2.5 k samples/s host bottleneck).
// One wave per row: each round every lane draws one candidate uniformly from the row's domain pool (Philox, keyed by seed,
// epoch and row), rejects it if it is in the row's own sequence, already accepted, or drawn by a lower lane in this round,
// and the survivors are appended in lane order until k are accepted: uniform without replacement over pool \ own.
#include "common.h"
#include "rng.h"

namespace amid {

constexpr unsigned SITE_NEG = 0x4E45;      // RNG site of the sampler (no dropout site uses it)

struct NegArgs {
    const long long* pool[2]; int n_pool[2];   // sorted unique item ids of each domain (dataset_seq.py:141-142)
    const long long* own;                      // concatenated own-sequence item ids of every row
    const int* own_off;                        // [N + 1]
    const long long* domain;                   // [N]
    long long* out;                            // [N, k]
    int N, k;
    unsigned long long seed; unsigned epoch;
};

__global__ __launch_bounds__(256) void sample_negatives_kernel(const NegArgs a) {
    extern __shared__ long long acc_all[];                     // [4][k] accepted ids of the 4 rows of this block
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int r = blockIdx.x * 4 + wv;
    if (r >= a.N) return;
    long long* acc = acc_all + (size_t)wv * a.k;
    const int dom = a.domain[r] != 0;
    const long long* pool = a.pool[dom];
    const unsigned n = (unsigned)a.n_pool[dom];
    const long long* own = a.own + a.own_off[r];
    const int n_own = a.own_off[r + 1] - a.own_off[r];
    int have = 0;
    for (unsigned round = 0; have < a.k && round < 4096u; ++round) {
        const uint4 rn = rng_call(a.seed, ((unsigned long long)r << 20) | ((unsigned long long)round << 6) | lane, SITE_NEG, a.epoch);
        // Lemire's multiply-shift maps 32 random bits onto [0, n) (bias < n / 2^32, far below sampling noise)
        const long long cand = pool[(unsigned)(((unsigned long long)rn.x * n) >> 32)];
        bool ok = true;
        for (int i = 0; i < n_own && ok; ++i) ok = own[i] != cand;
        for (int i = 0; i < have && ok; ++i) ok = acc[i] != cand;
        // duplicates inside the round: a lane loses to any lower lane holding the same candidate
        for (int l = 0; l < 64; ++l) {
            const long long other = __shfl(cand, l, 64);
            if (l < lane && other == cand) ok = false;
        }
        const unsigned long long m = __ballot(ok);
        const int slot = have + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && slot < a.k) { acc[slot] = cand; a.out[(long long)r * a.k + slot] = cand; }
        have = min(a.k, have + __popcll(m));
    }
    if (have < a.k && lane == 0) a.out[(long long)r * a.k] = -1;                     // pool exhausted: reported by the host wrapper
}

}  // namespace amid

using namespace amid;

extern "C" int amid_sample_negatives_i64(const long long* pool_d1, int n_pool_d1, const long long* pool_d2, int n_pool_d2,
                                         const long long* own_items, const int* own_off, const long long* domain_id, int N, int k,
                                         unsigned long long seed, unsigned epoch, long long* out, void* stream) {
    AMID_CHECK_ARG(pool_d1 && pool_d2 && own_items && own_off && domain_id && out && N > 0 && k > 0 && n_pool_d1 > 0 && n_pool_d2 > 0);
    const size_t lds = (size_t)4 * k * sizeof(long long);
    if (lds > 64 * 1024) return AMID_ERR_UNSUPPORTED;
    NegArgs a;
    a.pool[0] = pool_d1; a.pool[1] = pool_d2; a.n_pool[0] = n_pool_d1; a.n_pool[1] = n_pool_d2;
    a.own = own_items; a.own_off = own_off; a.domain = domain_id; a.out = out; a.N = N; a.k = k; a.seed = seed; a.epoch = epoch;
    sample_negatives_kernel<<<(N + 3) / 4, 256, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}
