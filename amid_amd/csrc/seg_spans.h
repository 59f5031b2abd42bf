// The segment reduce's row helpers and its second phase as device functions (segreduce.hip: the standalone launches and the gradient
// tail; adam.hip: the optimizer launch of the live-sequence step, where the runs that cross chunk borders are finished by extra
// workgroups and applied on the spot -- no launch of their own).
#pragma once
#include "common.h"

namespace amid {

constexpr int SEG_CHUNK = 64;        // entries per wave in phase A (the step's own lists: the pad run spans ~360 of these chunks)
constexpr int SEG_CHUNK_SHORT = 16;  // ... for short lists without long runs (the data-parallel merge: <= world duplicates per id): 4x the waves
constexpr int SEG_BATCH = 16;      // rows a wave requests before folding them (8: 8 dependent round trips per chunk, 16: 4)

__device__ __forceinline__ int seg_of_entry(const int* __restrict__ seg_off, int U, int e) {
    // largest u in [0,U) with seg_off[u] <= e
    int lo = 0, hi = U - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg_off[mid] <= e) lo = mid; else hi = mid - 1;
    }
    return lo;
}

template <int VEC> struct RowVec { float v[VEC]; };

template <int VEC>
__device__ __forceinline__ RowVec<VEC> load_row(const float* __restrict__ base, long long row, int D, int lane) {
    RowVec<VEC> r;
    const float* p = base + row * D + lane * VEC;
    if constexpr (VEC == 4) { float4 t = ld4(p); r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; }
    else if constexpr (VEC == 2) { float2 t = *reinterpret_cast<const float2*>(p); r.v[0] = t.x; r.v[1] = t.y; }
    else { r.v[0] = *p; }
    return r;
}
template <int VEC>
__device__ __forceinline__ void store_row(float* __restrict__ base, long long row, int D, int lane, const RowVec<VEC>& r) {
    float* p = base + row * D + lane * VEC;
    if constexpr (VEC == 4) st4(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3]));
    else if constexpr (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
    else *p = r.v[0];
}

// Does chunk c own a border-crossing run (its tail run continues into the next chunk and STARTED in c)?  Block-uniform.  u: the run,
// c_last: the last chunk it reaches.
__device__ __forceinline__ bool spans_owner(int c, const int* __restrict__ seg_off, const int* __restrict__ seg_of, int n, int chunk, int& u,
                                            int& c_last) {
    const int e0 = c * chunk;
    if (e0 >= n) return false;
    const int e_end = min(e0 + chunk, n);
    if (e_end >= n) return false;                     // the last chunk's tail run cannot continue
    u = seg_of[e_end - 1];
    if (seg_of[e_end] != u) return false;             // the tail run ends inside this chunk
    const int s_beg = seg_off[u], s_end = seg_off[u + 1];
    if (s_beg < e0) return false;                     // started in an earlier chunk: that chunk owns the sum
    c_last = (s_end - 1) / chunk;
    return true;
}

// The run's partial rows (phase A left one per chunk: the first chunk's tail slot 1, the later chunks' head slot 0) summed by SIXTEEN
// lanes-of-rows: virtual wave k adds chunks c + k, c + k + 16, ... in order into red[k]; the caller adds red[0 .. 15] in order
// (spans_total).  NW real waves share the sixteen (16: one each; 4: four each, one after the other) -- the same additions in the same
// order either way, hence the same bits whichever launch finishes a run.
// AGENT: the partial rows were written by other workgroups of THIS launch (tail_parts.h store_row_agent): read with the same scope.
template <int VEC> __device__ __forceinline__ RowVec<VEC> load_row_agent(const float* __restrict__ base, long long row, int D, int lane);
template <int VEC, bool AGENT>
__device__ __forceinline__ RowVec<VEC> load_partial_row(const float* __restrict__ base, long long row, int D, int lane) {
    if constexpr (AGENT) return load_row_agent<VEC>(base, row, D, lane); else return load_row<VEC>(base, row, D, lane);
}
template <int VEC, int NW, bool AGENT = false>
__device__ __forceinline__ void spans_partials(float (*red)[VEC * 64], int c, int c_last, const float* __restrict__ partial) {
    static_assert(16 % NW == 0, "virtual waves per real wave");
    constexpr int PER = 16 / NW;                      // this wave's virtual waves w, w + NW, ...: walked SIDE BY SIDE (their loads in flight together)
    const int D = VEC * 64;
    const int lane = lane_id(), w = wave_id();
    RowVec<VEC> acc[PER];
    int cc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        cc[j] = c + w + NW * j;
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[j].v[k] = 0.f;
    }
    // (AGENT -- one workgroup finishes the run behind every other's back, on the launch's critical path: eight rows per virtual wave and round)
    constexpr int RPR = (AGENT && VEC <= 2) ? 8 : 4;
    for (bool any = true; any;) {                     // RPR rows per virtual wave and round, while it has as many left ...
        any = false;
        RowVec<VEC> r[PER][RPR];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (cc[j] + 16 * (RPR - 1) <= c_last) {   // (wave-uniform)
#pragma unroll
                for (int i = 0; i < RPR; ++i) { const int ci = cc[j] + 16 * i; r[j][i] = load_partial_row<VEC, AGENT>(partial, (long long)ci * 2 + (ci == c ? 1 : 0), D, lane); }
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (cc[j] + 16 * (RPR - 1) <= c_last) {
#pragma unroll
                for (int i = 0; i < RPR; ++i)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[j].v[k] += r[j][i].v[k];
                cc[j] += 16 * RPR;
                any = true;
            }
        }
    }
    if constexpr (RPR == 8) {                         // ... then four at a time while it has four left ...
        for (bool any = true; any;) {
            any = false;
            RowVec<VEC> r[PER][4];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (cc[j] + 48 <= c_last) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int ci = cc[j] + 16 * i; r[j][i] = load_partial_row<VEC, AGENT>(partial, (long long)ci * 2 + (ci == c ? 1 : 0), D, lane); }
                }
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (cc[j] + 48 <= c_last) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[j].v[k] += r[j][i].v[k];
                    cc[j] += 64;
                    any = true;
                }
            }
        }
    }
    for (bool any = true; any;) {                     // ... then one by one
        any = false;
        RowVec<VEC> r[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (cc[j] <= c_last) r[j] = load_partial_row<VEC, AGENT>(partial, (long long)cc[j] * 2 + (cc[j] == c ? 1 : 0), D, lane);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (cc[j] <= c_last) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[j].v[k] += r[j].v[k];
                cc[j] += 16;
                any = true;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[w + NW * j][lane * VEC + k] = acc[j].v[k];
}
template <int VEC>
__device__ __forceinline__ float spans_total(const float (*red)[VEC * 64], int d) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][d];
    return s;
}

}  // namespace amid
