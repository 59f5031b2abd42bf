// The gradient tail's roles as device functions (segreduce.hip: the tail launches; adam.hip: the tail with the optimizer folded in):
// phase A of the segment reduce over 64-entry chunks of the sorted list and the position rows' gradients summed from the gradient rows.
// Reference: the four dense EmbeddingBackward index_adds of autograd through model_seq.py:418-421 and pos_emb's (:361-362).
#pragma once
#include "common.h"
#include "seg_spans.h"

namespace amid {

// a consumer of finished gradient slices (the folded optimizer: Adam on the spot); the default does nothing
struct NoSink {
    __device__ __forceinline__ void quad(float*, float4) const {}
    __device__ __forceinline__ void one(float*, float) const {}
};

// how a chunk's partial rows (the pieces of runs that cross chunk borders) leave the wave: plain stores when the next LAUNCH reads them,
// agent-scope stores when another workgroup of the SAME launch does (they write through to where every XCD's loads of that scope look)
template <int VEC>
__device__ __forceinline__ void store_row_agent(float* __restrict__ base, long long row, int D, int lane, const RowVec<VEC>& r) {
    float* p = base + row * D + lane * VEC;
    if constexpr (VEC == 1) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(r.v[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            const unsigned long long w = (unsigned long long)__float_as_uint(r.v[k]) | ((unsigned long long)__float_as_uint(r.v[k + 1]) << 32);
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(p + k), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <int VEC>
__device__ __forceinline__ RowVec<VEC> load_row_agent(const float* __restrict__ base, long long row, int D, int lane) {
    RowVec<VEC> r;
    const float* p = base + row * D + lane * VEC;
    if constexpr (VEC == 1) {
        r.v[0] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            const unsigned long long w = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.v[k] = __uint_as_float((unsigned)w); r.v[k + 1] = __uint_as_float((unsigned)(w >> 32));
        }
    }
    return r;
}

// phase A: one wave per 64-entry chunk of the sorted list.  Lane l keeps (position, run index) of entry e0 + l; rows
// are fetched SEG_BATCH at a time whatever runs they belong to (independent loads), then folded in order, flushing at
// every run change: control flow is wave-uniform (run indices come from readlane-style shuffles), no global load
// sits on the per-run critical path.
// CUT_ONLY: only the pieces of runs that cross the chunk's borders are summed (two per chunk at most) and stored with agent scope -- the
// runs that lie inside a chunk belong to the row workers of the same launch (adam.hip grad_tail_opt_kernel); a chunk without a cut run
// returns at once.  The cut pieces' additions are the same, in the same order, either way.
// (CUT_ONLY, seg_off != NULL: only of runs LONGER than a chunk -- the shorter ones, which cross one border at most, are the row workers' too)
template <int VEC, bool CUT_ONLY = false>
__device__ __forceinline__ void segreduce_chunks_block(const float* __restrict__ grad_rows, const int* __restrict__ pos_sorted,
                                                       const int* __restrict__ seg_of, int n, float* __restrict__ uniq_grad,
                                                       float* __restrict__ partial, int block, int chunk = SEG_CHUNK,
                                                       const int* __restrict__ seg_off = nullptr) {
    const int D = VEC * 64;
    const int lane = lane_id();
    const int c = block * 4 + wave_id();
    const int e0 = c * chunk;
    if (e0 >= n) return;
    const int cnt = min(chunk, n - e0);
    const bool valid = lane < cnt;
    const int mypos = valid ? pos_sorted[e0 + lane] : 0;
    const int mysg = valid ? seg_of[e0 + lane] : -1;
    // lane indices below are wave-uniform: v_readlane (a few cycles) instead of a ds_bpermute round trip per row
    const int first_sg = __builtin_amdgcn_readlane(mysg, 0), last_sg = __builtin_amdgcn_readlane(mysg, __builtin_amdgcn_readfirstlane(cnt - 1));
    bool starts_before = (e0 > 0) && (seg_of[e0 - 1] == first_sg);
    bool continues_after = (e0 + cnt < n) && (seg_of[e0 + cnt] == last_sg);
    if constexpr (CUT_ONLY) {
        if (seg_off != nullptr) {
            if (starts_before && seg_off[first_sg + 1] - seg_off[first_sg] <= chunk) starts_before = false;
            if (continues_after && seg_off[last_sg + 1] - seg_off[last_sg] <= chunk) continues_after = false;
        }
        if (!starts_before && !continues_after) return;
    }
    RowVec<VEC> acc;
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc.v[k] = 0.f;
    int cur = first_sg;
    auto flush = [&](int sg) {
        const bool head_cut = (sg == first_sg) && starts_before;
        const bool tail_cut = (sg == last_sg) && continues_after;
        if constexpr (CUT_ONLY) {
            if (head_cut || tail_cut) store_row_agent<VEC>(partial, (long long)c * 2 + (head_cut ? 0 : 1), D, lane, acc);
        } else {
            if (!head_cut && !tail_cut) store_row<VEC>(uniq_grad, sg, D, lane, acc);
            else store_row<VEC>(partial, (long long)c * 2 + (head_cut ? 0 : 1), D, lane, acc);
        }
    };
    for (int i = 0; i < cnt; i += SEG_BATCH) {
        RowVec<VEC> r[SEG_BATCH];
#pragma unroll
        for (int j = 0; j < SEG_BATCH; ++j) {
            const int src = min(i + j, cnt - 1);
            r[j] = load_row<VEC>(grad_rows, __builtin_amdgcn_readlane(mypos, __builtin_amdgcn_readfirstlane(src)), D, lane);
        }
#pragma unroll
        for (int j = 0; j < SEG_BATCH; ++j) {
            if (i + j < cnt) {
                const int sg = __builtin_amdgcn_readlane(mysg, __builtin_amdgcn_readfirstlane(i + j));
                if (sg != cur) {
                    flush(cur);
                    cur = sg;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc.v[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc.v[k] += r[j].v[k];
            }
        }
    }
    flush(cur);
}

// ---- the position rows' gradients of the live-sequence train step ------------------------------------------------------------------------
// The embedding layer's element-wise backward ran on the last strip launch (sasrec_strip.hip StripQkvBwdArgs::emb_tmq), so dP_g[t] = sum
// over the LIVE sequences b of domain g of grad_rows[(g B + b) T + t] is a fixed-order sum over rows that are already final: 256 threads =
// 32 float4 columns x 8 groups of sequences (group p adds live sequences p, p + 8, ... in order, eight loads in flight), the eight group
// sums added in order.  The dead sequences' rows are neither written nor read by anybody.
struct PosSum { const float* rows; const int* live; int B, T; float* dst[2]; int nblk; };

template <class Sink = NoSink>
__device__ __forceinline__ void pos_sum_block(const PosSum& ps, int D, int g, int bx, int nbx, const Sink sink = Sink()) {
    __shared__ float4 pred[8][33];
    const int el = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int n0 = ps.live[ps.B];
    const int s0 = g ? n0 : 0, n = g ? ps.B - n0 : n0;
    const int count = ps.T * D;
    const long long seq = (long long)ps.T * D;
    const float* __restrict__ base = ps.rows + (long long)g * ps.B * seq;
    const int* __restrict__ lv = ps.live + s0;
    for (int e0 = bx * 128; e0 < count; e0 += nbx * 128) {        // block-uniform
        const int e = e0 + 4 * el;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < count) {
            int k = pg;
            for (; k + 56 < n; k += 64) {
                int b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = lv[k + 8 * j];
                float4 r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = ld4(base + b[j] * seq + e);
#pragma unroll
                for (int j = 0; j < 8; ++j) s = f4add(s, r[j]);
            }
            for (; k < n; k += 8) s = f4add(s, ld4(base + lv[k] * seq + e));
        }
        pred[pg][el] = s;
        __syncthreads();
        if (pg == 0 && e < count) {
            float4 t = pred[0][el];
#pragma unroll
            for (int q = 1; q < 8; ++q) t = f4add(t, pred[q][el]);
            st4(ps.dst[g] + e, t);
            sink.quad(ps.dst[g] + e, t);
        }
        __syncthreads();
    }
}

}  // namespace amid
