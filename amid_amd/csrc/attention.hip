// K2: multi-head attention core, general form (any T whose per-head K / V fit in LDS: ~900 at head dim 16; head dim 8/16/32),
// forward and backward.
// Reference: torch.nn.MultiheadAttention as called at model_seq.py:374 (explicit softmax path: q scaled
// by sqrt(1/hd) BEFORE q.k^T, additive -inf causal mask, dropout p=.5 on the probabilities) and the
// hand-written Attention of BERT4Rec (model_seq.py:149-162: scores / sqrt(d_k), masked_fill(mask == 0, -1e9)
// with ONE key mask taken from seq_d2 > 0 for both domains, dropout p=.1).
//
// One workgroup per (domain, batch row); one wave per head.  A lane owns one query row, so the
// softmax reduction is lane-local (no cross-lane traffic at all); K/V rows are LDS broadcasts
// (every lane reads the same address).  Backward recomputes the probabilities from the saved
// row max / 1/sum and runs twice: lanes = queries for dQ, then lanes = keys for dK/dV, so that no
// gradient needs a cross-lane or cross-wave reduction; the dropout keep bits found in the first
// phase are handed to the second through a 64-bit-per-row LDS mask instead of re-running Philox.
// VALU-bound (hd 16: 4*T^2*hd FLOP per (b, h) forward); no HBM pressure: q/k/v/o tiles are read once.
#include "common.h"
#include "rng.h"

namespace amid {

struct AttnArgs {
    const float* q; const float* k; const float* v;       // [2M, D]
    float* o;                                             // [2M, D]
    float* stats;                                         // [2M, H, 2] row max, 1 / row sum
    const float* d_o; float* dq; float* dk; float* dv;    // backward only
    const unsigned char* key_keep;                        // [B, T] 1 = key visible (BERT4Rec), null = no key mask
    int B, T, D, H;
    int causal;                                           // 1: SASRec causal mask, q pre-scaled; 0: BERT4Rec
    float scale;                                          // sqrt(1/hd) (causal) or sqrt(d_k) divisor (BERT)
    const StepState* st; int train; unsigned thr16; float dscale; int layer;
    int stagger_from, stagger_sleeps;      // set by the MFMA backward launcher only
    const long long* row_domain;           // backward only, optional [B]: sequence (g, b) has a gradient only if (row_domain[b] != 0) == g
    const int* live;                       // matrix-core kernels only, optional [B + 1] (amid_live_list_i32): the launch covers the B
                                           // listed sequences only -- slot j < live[B]: (0, live[j]), else (1, live[j]); nothing else is touched
};

template <int HD>
__device__ __forceinline__ float dot_lds(const float (&a)[HD], const float* __restrict__ b) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
        const float4 t = ld4(b + d);
        s = fmaf(a[d], t.x, s); s = fmaf(a[d + 1], t.y, s); s = fmaf(a[d + 2], t.z, s); s = fmaf(a[d + 3], t.w, s);
    }
    return s;
}

template <int HD>
__device__ __forceinline__ void load_vec(float (&a)[HD], const float* __restrict__ p) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) { const float4 t = ld4(p + d); a[d] = t.x; a[d + 1] = t.y; a[d + 2] = t.z; a[d + 3] = t.w; }
}
template <int HD>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&a)[HD]) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) st4(p + d, make_float4(a[d], a[d + 1], a[d + 2], a[d + 3]));
}

// rows x `cols` window of a [rows, ld] global tile -> dense [rows][cols] LDS image (cols = the columns of the workgroup's heads)
__device__ __forceinline__ void copy_cols(float* __restrict__ dst, const float* __restrict__ src, int rows, int cols, int ld) {
    const int q = cols >> 2;
    for (int i = threadIdx.x; i < rows * q; i += blockDim.x) {
        const int r = i / q, c = i - r * q;
        st4(dst + r * cols + 4 * c, ld4(src + (long long)r * ld + 4 * c));
    }
}

// score of (query i, key j) exactly as the forward computes it
template <int HD>
__device__ __forceinline__ float score(const AttnArgs& a, const float (&qs)[HD], const float* __restrict__ krow, int i, int j, bool key_ok) {
    float s = dot_lds<HD>(qs, krow);
    if (a.causal) { if (j > i) s = -INFINITY; }
    else { s = s / a.scale; if (!key_ok) s = -1e9f; }
    return s;
}

template <int HD>
__global__ __launch_bounds__(512) void attn_fwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D;
    // blockIdx.y = head group: the workgroup's waves take heads hg * HL .. + HL - 1 and only their columns of K / V live in LDS
    // (long sequences: all eight heads' K and V no longer fit at T > ~150, see attn_head_groups)
    const int HL = blockDim.x >> 6, DL = HL * HD, hl = wave_id(), lane = lane_id();
    const int h = blockIdx.y * HL + hl;
    float* Ks = smem;
    float* Vs = smem + T * DL;
    const int seq = blockIdx.x, g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    copy_cols(Ks, a.k + rowbase * D + blockIdx.y * DL, T, DL, D);
    copy_cols(Vs, a.v + rowbase * D + blockIdx.y * DL, T, DL, D);
    __syncthreads();
    const int dbits = spec_bits(a.thr16), per = 128 / dbits;       // decisions per Philox call
    const unsigned dthr = spec_thr(a.thr16);
    const int calls_per_row = (T + per - 1) / per;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    const unsigned site = site_id(g, a.layer, SITE_ATTN);
    const unsigned char* kk = a.key_keep ? a.key_keep + (long long)b * T : nullptr;
    for (int qb = 0; qb * 64 < T; ++qb) {
        const int i = qb * 64 + lane;
        const bool valid = i < T;
        const int ic = valid ? i : T - 1;
        float qs[HD];
        load_vec<HD>(qs, a.q + (rowbase + ic) * D + h * HD);
        if (a.causal) {
#pragma unroll
            for (int d = 0; d < HD; ++d) qs[d] *= a.scale;
        }
        const int jmax = a.causal ? min(T, qb * 64 + 64) : T;
        float m = -INFINITY;
        for (int j = 0; j < jmax; ++j) m = fmaxf(m, score<HD>(a, qs, Ks + j * DL + hl * HD, ic, j, kk ? kk[j] != 0 : true));
        float l = 0.f;
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
        const unsigned long long rowcall = ((unsigned long long)(b * a.H + h) * T + ic) * calls_per_row;
        uint4 r = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int j0 = 0; j0 < jmax; j0 += 8) {
            if (a.train && (j0 % per) == 0) r = rng_call(seed, rowcall + j0 / per, site, step);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                if (j < jmax) {
                    const float s = score<HD>(a, qs, Ks + j * DL + hl * HD, ic, j, kk ? kk[j] != 0 : true);
                    const float p = expf(s - m);
                    l += p;
                    const float pd = (!a.train || rng_field(r, j % per, dbits) >= dthr) ? p * a.dscale : 0.f;
                    const float* vr = Vs + j * DL + hl * HD;
#pragma unroll
                    for (int d = 0; d < HD; d += 4) {
                        const float4 t = ld4(vr + d);
                        acc[d] = fmaf(pd, t.x, acc[d]); acc[d + 1] = fmaf(pd, t.y, acc[d + 1]);
                        acc[d + 2] = fmaf(pd, t.z, acc[d + 2]); acc[d + 3] = fmaf(pd, t.w, acc[d + 3]);
                    }
                }
            }
        }
        const float rl = 1.0f / l;
        if (valid) {
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] *= rl;
            store_vec<HD>(a.o + (rowbase + i) * D + h * HD, acc);
            if (a.stats) {
                float* sp = a.stats + ((rowbase + i) * a.H + h) * 2;
                sp[0] = m; sp[1] = rl;
            }
        }
    }
}

template <int HD>
__global__ __launch_bounds__(512) void attn_bwd_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, D = a.D, H = a.H;
    const int TW = (T + 63) >> 6;                              // 64-bit keep words per query row
    const int HL = blockDim.x >> 6, DL = HL * HD, hl = wave_id(), lane = lane_id();      // head group blockIdx.y, as in the forward
    const int h = blockIdx.y * HL + hl;
    const int col0 = blockIdx.y * DL;
    float* S0 = smem;                                          // K, then Q        [T][DL]
    float* S1 = smem + T * DL;                                 // V, then dO
    float* rstat = smem + 2 * T * DL;                          // [HL][T][3]  m, 1/l, delta
    unsigned long long* keepw = reinterpret_cast<unsigned long long*>(rstat + HL * T * 3 + ((2 * T * DL + HL * T * 3) & 1));   // [HL][T][TW]
    const int seq = blockIdx.x, g = seq / a.B, b = seq - g * a.B;
    const long long rowbase = (long long)seq * T;
    const int dbits = spec_bits(a.thr16), per = 128 / dbits;
    const unsigned dthr = spec_thr(a.thr16);
    const int calls_per_row = (T + per - 1) / per;
    unsigned long long seed = 0; unsigned step = 0;
    if (a.train) { seed = a.st->seed; step = (unsigned)a.st->step; }
    const unsigned site = site_id(g, a.layer, SITE_ATTN);
    const unsigned char* kk = a.key_keep ? a.key_keep + (long long)b * T : nullptr;

    // ---- phase 1: lanes = queries -> dQ; leaves row stats and dropout keep words in LDS --------
    copy_cols(S0, a.k + rowbase * D + col0, T, DL, D);
    copy_cols(S1, a.v + rowbase * D + col0, T, DL, D);
    __syncthreads();
    for (int qb = 0; qb * 64 < T; ++qb) {
        const int i = qb * 64 + lane;
        const bool valid = i < T;
        const int ic = valid ? i : T - 1;
        float qs[HD], dO[HD], acc[HD];
        load_vec<HD>(qs, a.q + (rowbase + ic) * D + h * HD);
        load_vec<HD>(dO, a.d_o + (rowbase + ic) * D + h * HD);
        float delta = 0.f;
        {
            float ov[HD];
            load_vec<HD>(ov, a.o + (rowbase + ic) * D + h * HD);
#pragma unroll
            for (int d = 0; d < HD; ++d) delta = fmaf(dO[d], ov[d], delta);
        }
        if (a.causal) {
#pragma unroll
            for (int d = 0; d < HD; ++d) qs[d] *= a.scale;
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
        const float* sp = a.stats + ((rowbase + ic) * H + h) * 2;
        const float m = sp[0], rl = sp[1];
        if (valid) { float* rs = rstat + (hl * T + i) * 3; rs[0] = m; rs[1] = rl; rs[2] = delta; }
        const int jmax = a.causal ? min(T, qb * 64 + 64) : T;
        const unsigned long long rowcall = ((unsigned long long)(b * H + h) * T + ic) * calls_per_row;
        unsigned long long kw = 0;
        uint4 r = make_uint4(~0u, ~0u, ~0u, ~0u);
        for (int j0 = 0; j0 < jmax; j0 += 8) {                 // phase 2 only needs the bits of pairs (i, j < jmax)
            if ((j0 & 63) == 0) kw = 0;
            if (a.train && (j0 % per) == 0) r = rng_call(seed, rowcall + j0 / per, site, step);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                const bool keep = !a.train || rng_field(r, j % per, dbits) >= dthr;
                if (keep) kw |= 1ull << (j & 63);
                if (j < jmax) {
                    const bool key_ok = kk ? kk[j] != 0 : true;
                    const float s = score<HD>(a, qs, S0 + j * DL + hl * HD, ic, j, key_ok);
                    const float p = expf(s - m) * rl;
                    const float dpd = dot_lds<HD>(dO, S1 + j * DL + hl * HD);
                    const float dp = keep ? dpd * a.dscale : 0.f;
                    // masked_fill cuts the gradient of a masked score (it matters when EVERY key of a row is masked: p = 1/T there)
                    const float ds = key_ok ? p * (dp - delta) : 0.f;
                    const float* kr = S0 + j * DL + hl * HD;
#pragma unroll
                    for (int d = 0; d < HD; d += 4) {
                        const float4 t = ld4(kr + d);
                        acc[d] = fmaf(ds, t.x, acc[d]); acc[d + 1] = fmaf(ds, t.y, acc[d + 1]);
                        acc[d + 2] = fmaf(ds, t.z, acc[d + 2]); acc[d + 3] = fmaf(ds, t.w, acc[d + 3]);
                    }
                }
            }
            if (valid && ((j0 & 63) == 56 || j0 + 8 >= jmax)) keepw[(hl * T + i) * TW + (j0 >> 6)] = kw;
        }
        if (valid) {
            const float sc = a.causal ? a.scale : 1.0f / a.scale;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] *= sc;
            store_vec<HD>(a.dq + (rowbase + i) * D + h * HD, acc);
        }
    }
    __syncthreads();
    // ---- phase 2: lanes = keys -> dK, dV -------------------------------------------------------
    copy_cols(S0, a.q + rowbase * D + col0, T, DL, D);
    copy_cols(S1, a.d_o + rowbase * D + col0, T, DL, D);
    __syncthreads();
    for (int kb = 0; kb * 64 < T; ++kb) {
        const int j = kb * 64 + lane;
        const bool valid = j < T;
        const int jc = valid ? j : T - 1;
        float kv[HD], vv[HD], dk[HD], dv[HD];
        load_vec<HD>(kv, a.k + (rowbase + jc) * D + h * HD);
        load_vec<HD>(vv, a.v + (rowbase + jc) * D + h * HD);
#pragma unroll
        for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        const bool key_ok = kk ? kk[jc] != 0 : true;
        const int i0 = a.causal ? kb * 64 : 0;
        for (int i = i0; i < T; ++i) {
            const float* qr = S0 + i * DL + hl * HD;
            const float* dor = S1 + i * DL + hl * HD;
            const float* rs = rstat + (hl * T + i) * 3;
            float qs[HD];
#pragma unroll
            for (int d = 0; d < HD; d += 4) { const float4 t = ld4(qr + d); qs[d] = t.x; qs[d + 1] = t.y; qs[d + 2] = t.z; qs[d + 3] = t.w; }
            if (a.causal) {
#pragma unroll
                for (int d = 0; d < HD; ++d) qs[d] *= a.scale;
            }
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s = fmaf(qs[d], kv[d], s);
            if (a.causal) { if (jc > i) s = -INFINITY; }
            else { s = s / a.scale; if (!key_ok) s = -1e9f; }
            const float p = expf(s - rs[0]) * rs[1];
            const float dpd = dot_lds<HD>(vv, dor);
            const bool keep = (keepw[(hl * T + i) * TW + (jc >> 6)] >> (jc & 63)) & 1ull;
            const float pd = keep ? p * a.dscale : 0.f;
            const float dp = keep ? dpd * a.dscale : 0.f;
            const float ds = key_ok ? p * (dp - rs[2]) : 0.f;
            const float dsq = a.causal ? ds : ds / a.scale;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 t = ld4(dor + d);
                dv[d] = fmaf(pd, t.x, dv[d]); dv[d + 1] = fmaf(pd, t.y, dv[d + 1]);
                dv[d + 2] = fmaf(pd, t.z, dv[d + 2]); dv[d + 3] = fmaf(pd, t.w, dv[d + 3]);
                dk[d] = fmaf(dsq, qs[d], dk[d]); dk[d + 1] = fmaf(dsq, qs[d + 1], dk[d + 1]);
                dk[d + 2] = fmaf(dsq, qs[d + 2], dk[d + 2]); dk[d + 3] = fmaf(dsq, qs[d + 3], dk[d + 3]);
            }
        }
        if (valid) {
            store_vec<HD>(a.dk + (rowbase + j) * D + h * HD, dk);
            store_vec<HD>(a.dv + (rowbase + j) * D + h * HD, dv);
        }
    }
}

}  // namespace amid

using namespace amid;

// attention_mfma.hip: matrix-core kernels for the causal head-dim-16 T <= 64 case
int amid_attn_mfma_fwd_launch(const void* args, void* stream);
int amid_attn_mfma_bwd_launch(const void* args, void* stream);
// (head dim 8 -- the reference's default --emb_dim 64 with 8 heads -- runs as pairs of heads per 16-column tile)
static bool mfma_shape(const AttnArgs& a) {
    const int hd = a.D / a.H;
    return a.causal && a.key_keep == nullptr && a.T <= 64 && a.D % a.H == 0 && a.H <= 8 && (hd == 16 || (hd == 8 && a.H % 2 == 0));
}
// attention_mfma_long.hip: the same shape at 64 < T <= 256, queries and keys walked in blocks of 64
int amid_attn_long_fwd_launch(const void* args, void* stream);
int amid_attn_long_bwd_launch(const void* args, void* stream);
static bool long_shape(const AttnArgs& a) {
    return a.causal && a.key_keep == nullptr && a.D / a.H == 16 && a.T > 64 && a.T <= 256 && a.H <= 8 && a.H % 4 == 0;
}
// attention_mfma_bert.hip: matrix-core kernels for the bidirectional head-dim-32 T <= 64 case (key mask optional)
int amid_attn_bert_fwd_launch(const void* args, void* stream);
int amid_attn_bert_bwd_launch(const void* args, void* stream);
static bool bert_shape(const AttnArgs& a) { return !a.causal && a.D / a.H == 32 && a.T <= 64 && a.H <= 8; }

// LDS of one workgroup that serves H / hg heads (their D / hg columns of K / V, then of Q / dO)
static size_t attn_fwd_lds(int T, int D, int hg) { return (size_t)2 * T * (D / hg) * sizeof(float); }
static size_t attn_bwd_lds(int T, int D, int H, int hg) {
    size_t f = (size_t)2 * T * (D / hg) + (size_t)(H / hg) * T * 3;
    f += f & 1;
    return f * sizeof(float) + (size_t)(H / hg) * T * ((T + 63) / 64) * 8;
}
// smallest number of head groups (workgroups per sequence) whose LDS image fits: 1 up to T ~ 118 (backward) at D = 128, 2 up to
// ~240, ...; 0 = does not fit even with one head per workgroup
template <typename F>
static int attn_head_groups(int H, F lds_of) {
    for (int hg = 1; hg <= H; hg <<= 1)
        if (H % hg == 0 && lds_of(hg) <= (size_t)160 * 1024) return hg;
    return 0;
}

template <typename KernelT>
static int attn_launch(KernelT kern, const AttnArgs& a, size_t lds, int hg, void* stream) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    kern<<<dim3(2 * a.B, hg), (a.H / hg) * 64, lds, (hipStream_t)stream>>>(a);
    AMID_LAUNCH_CHECK();
    return AMID_OK;
}

static int attn_fill(AttnArgs& a, const float* q, const float* k, const float* v, const unsigned char* key_keep, int B, int T, int D,
                     int H, int causal, int layer, const void* step_state, int train, float p_drop) {
    if (!(q && k && v) || B <= 0 || T <= 0 || H <= 0 || H > 8 || D % H != 0 || (train && !step_state)) return AMID_ERR_ARG;
    a.q = q; a.k = k; a.v = v; a.key_keep = key_keep;
    a.B = B; a.T = T; a.D = D; a.H = H; a.causal = causal; a.layer = layer;
    const int hd = D / H;
    a.scale = causal ? sqrtf(1.0f / (float)hd) : sqrtf((float)hd);
    a.st = (const StepState*)step_state;
    a.train = (train && p_drop > 0.f) ? 1 : 0;
    a.thr16 = keep_thr16(p_drop);
    a.dscale = a.train ? 1.0f / (1.0f - p_drop) : 1.0f;
    a.stagger_from = -1; a.stagger_sleeps = 0;
    return AMID_OK;
}

extern "C" int amid_attn_fwd_f32(const float* q, const float* k, const float* v, const unsigned char* key_keep, int B, int T, int D, int H,
                                 int causal, int layer, const void* step_state, int train, float p_drop, float* o, float* stats,
                                 void* stream) {
    AttnArgs a = {};
    if (int e = attn_fill(a, q, k, v, key_keep, B, T, D, H, causal, layer, step_state, train, p_drop)) return e;
    AMID_CHECK_ARG(o);
    a.o = o; a.stats = stats;
    if (mfma_shape(a)) return amid_attn_mfma_fwd_launch(&a, stream);
    if (long_shape(a)) return amid_attn_long_fwd_launch(&a, stream);
    if (bert_shape(a)) return amid_attn_bert_fwd_launch(&a, stream);
    const int hg = attn_head_groups(H, [&](int g_) { return attn_fwd_lds(T, D, g_); });
    if (hg == 0) return AMID_ERR_UNSUPPORTED;
    const size_t lds = attn_fwd_lds(T, D, hg);
    switch (D / H) {
        case 8: return attn_launch(attn_fwd_kernel<8>, a, lds, hg, stream);
        case 16: return attn_launch(attn_fwd_kernel<16>, a, lds, hg, stream);
        case 32: return attn_launch(attn_fwd_kernel<32>, a, lds, hg, stream);
        default: return AMID_ERR_UNSUPPORTED;
    }
}

static int attn_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                         const unsigned char* key_keep, int B, int T, int D, int H, int causal, int layer, const void* step_state,
                         int train, float p_drop, float* dq, float* dk, float* dv, const long long* row_domain, void* stream);

extern "C" int amid_attn_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                                 const unsigned char* key_keep, int B, int T, int D, int H, int causal, int layer, const void* step_state,
                                 int train, float p_drop, float* dq, float* dk, float* dv, void* stream) {
    return attn_bwd_impl(q, k, v, o, stats, d_o, key_keep, B, T, D, H, causal, layer, step_state, train, p_drop, dq, dk, dv, nullptr, stream);
}

// the matrix-core kernels over the sequences of a live list only (amid_live_list_i32; see AttnArgs::live): forward and backward of
// the fused train step, whose loss never reads the other domain's logits of a sample (train_sr.py:205-211).  Shapes outside the
// matrix-core kernels' range (causal, head dim 16, T <= 64) are refused: the caller then encodes every sequence.
extern "C" int amid_attn_live_supported(int T, int D, int H, int causal) {
    if (!(causal && H > 0 && D % H == 0 && T > 0 && T <= 64)) return 0;
    return (H <= 8 && (D / H == 16 || (D / H == 8 && H % 2 == 0))) ? 1 : 0;
}
extern "C" int amid_attn_fwd_live_f32(const float* q, const float* k, const float* v, int B, int T, int D, int H, int causal, int layer,
                                      const void* step_state, int train, float p_drop, float* o, float* stats, const int* live, void* stream) {
    AttnArgs a = {};
    if (int e = attn_fill(a, q, k, v, nullptr, B, T, D, H, causal, layer, step_state, train, p_drop)) return e;
    AMID_CHECK_ARG(o && live);
    a.o = o; a.stats = stats; a.live = live;
    if (!mfma_shape(a)) return AMID_ERR_UNSUPPORTED;
    return amid_attn_mfma_fwd_launch(&a, stream);
}
extern "C" int amid_attn_bwd_live_f32(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                                      int B, int T, int D, int H, int causal, int layer, const void* step_state, int train, float p_drop,
                                      float* dq, float* dk, float* dv, const int* live, void* stream) {
    AttnArgs a = {};
    if (int e = attn_fill(a, q, k, v, nullptr, B, T, D, H, causal, layer, step_state, train, p_drop)) return e;
    AMID_CHECK_ARG(o && stats && d_o && dq && dk && dv && live);
    a.o = const_cast<float*>(o); a.stats = const_cast<float*>(stats); a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv; a.live = live;
    if (!mfma_shape(a)) return AMID_ERR_UNSUPPORTED;
    return amid_attn_mfma_bwd_launch(&a, stream);
}

// ... and the BERT4Rec shape (bidirectional, key mask from seq_d2 > 0 for both encoders, 4 heads of 32, T <= 64: attention_mfma_bert.hip)
// over a live list: nothing of the other sequences is read or written
extern "C" int amid_attn_bert_live_supported(int T, int D, int H) { return (T > 0 && T <= 64 && H == 4 && D == 128) ? 1 : 0; }
extern "C" int amid_attn_bert_fwd_live_f32(const float* q, const float* k, const float* v, const unsigned char* key_keep, int B, int T, int D,
                                           int H, int layer, const void* step_state, int train, float p_drop, float* o, float* stats,
                                           const int* live, void* stream) {
    AttnArgs a = {};
    if (int e = attn_fill(a, q, k, v, key_keep, B, T, D, H, 0, layer, step_state, train, p_drop)) return e;
    AMID_CHECK_ARG(o && live);
    a.o = o; a.stats = stats; a.live = live;
    if (!amid_attn_bert_live_supported(T, D, H) || !bert_shape(a)) return AMID_ERR_UNSUPPORTED;
    return amid_attn_bert_fwd_launch(&a, stream);
}
extern "C" int amid_attn_bert_bwd_live_f32(const float* q, const float* k, const float* v, const float* o, const float* stats,
                                           const float* d_o, const unsigned char* key_keep, int B, int T, int D, int H, int layer,
                                           const void* step_state, int train, float p_drop, float* dq, float* dk, float* dv,
                                           const int* live, void* stream) {
    AttnArgs a = {};
    if (int e = attn_fill(a, q, k, v, key_keep, B, T, D, H, 0, layer, step_state, train, p_drop)) return e;
    AMID_CHECK_ARG(o && stats && d_o && dq && dk && dv && live);
    a.o = const_cast<float*>(o); a.stats = const_cast<float*>(stats); a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv; a.live = live;
    if (!amid_attn_bert_live_supported(T, D, H) || !bert_shape(a)) return AMID_ERR_UNSUPPORTED;
    return amid_attn_bert_bwd_launch(&a, stream);
}

// the same with the training loss's structure handed in: train_sr.py:205-211 masks row b's BCE of domain 1 - domain_id[b] with
// zero, so the sequence (g, b) with g != domain_id[b] receives an all-zero d_o and its dq / dk / dv are exact zeros -- the
// matrix-core kernels write those zeros without loading or computing anything (half of all sequences); the other kernels ignore
// the hint and compute the same zeros the long way.  Only valid when d_o really is zero there (the caller's loss is that BCE).
extern "C" int amid_attn_bwd_rows_f32(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                                      const unsigned char* key_keep, int B, int T, int D, int H, int causal, int layer,
                                      const void* step_state, int train, float p_drop, float* dq, float* dk, float* dv,
                                      const long long* row_domain, void* stream) {
    return attn_bwd_impl(q, k, v, o, stats, d_o, key_keep, B, T, D, H, causal, layer, step_state, train, p_drop, dq, dk, dv, row_domain, stream);
}

static int attn_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                         const unsigned char* key_keep, int B, int T, int D, int H, int causal, int layer, const void* step_state,
                         int train, float p_drop, float* dq, float* dk, float* dv, const long long* row_domain, void* stream) {
    AttnArgs a = {};
    if (int e = attn_fill(a, q, k, v, key_keep, B, T, D, H, causal, layer, step_state, train, p_drop)) return e;
    AMID_CHECK_ARG(o && stats && d_o && dq && dk && dv);
    a.o = const_cast<float*>(o); a.stats = const_cast<float*>(stats); a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv;
    a.row_domain = row_domain;
    if (mfma_shape(a)) return amid_attn_mfma_bwd_launch(&a, stream);
    if (long_shape(a)) return amid_attn_long_bwd_launch(&a, stream);
    if (bert_shape(a)) return amid_attn_bert_bwd_launch(&a, stream);
    const int hg = attn_head_groups(H, [&](int g_) { return attn_bwd_lds(T, D, H, g_); });
    if (hg == 0) return AMID_ERR_UNSUPPORTED;
    const size_t lds = attn_bwd_lds(T, D, H, hg);
    switch (D / H) {
        case 8: return attn_launch(attn_bwd_kernel<8>, a, lds, hg, stream);
        case 16: return attn_launch(attn_bwd_kernel<16>, a, lds, hg, stream);
        case 32: return attn_launch(attn_bwd_kernel<32>, a, lds, hg, stream);
        default: return AMID_ERR_UNSUPPORTED;
    }
}
