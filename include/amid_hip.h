/* libamid_hip.so -- C ABI of the MI355X (gfx950) kernels behind the AMID training hot path.
 *
 * The reference (WujiangXu/AMID) has no native layer: its "FFI" for this path is PyTorch's
 * ATen operators called from model_seq.py / train_sr.py.  Each entry point below names the
 * reference call site (file:line into the reference checkout) whose arithmetic it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; every pointer is DEVICE memory unless named host_*;
 *  - the caller owns every buffer including workspaces; the library keeps no state (the only
 *    handles it hands out are hipGraphExec / hipEvent wrappers the caller destroys);
 *  - all work is enqueued on the caller's hipStream_t (passed as void*), no internal sync, safe
 *    to capture into a hipGraph; scalars that change per step (step counter, lr, dropout seed,
 *    number of unique rows) live in device memory for that reason;
 *  - return 0 on success, AMID_ERR_* (< 0) for argument errors, a hipError_t (> 0) otherwise;
 *    nothing throws or aborts across the ABI;
 *  - fp32 everywhere; activations are row-major [rows, D]; "2M rows" = domain 0 rows [0, M)
 *    followed by domain 1 rows [M, 2M), M = B*T;
 *  - D in {64, 128} for the fused encoder kernels, D in {64, 128, 256} for the segment reduce,
 *    D % 4 == 0 elsewhere.
 */
#ifndef AMID_HIP_H
#define AMID_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AMID_OK 0
#define AMID_ERR_ARG (-1)
#define AMID_ERR_UNSUPPORTED (-2)
/* bits of a plan's device error word (err_flag arguments): 1 = an item index outside the table (nn.Embedding would raise,
 * model_seq.py:27-29); 2 = a data-parallel step found more unique rows than the caller's bound umax (amid_grad_tail_pack_f32) */
#define AMID_FLAG_INDEX_RANGE (1)
#define AMID_FLAG_UMAX_EXCEEDED (2)

int amid_version(void);
const char* amid_error_string(int code);
int amid_device_sync(void);

/* ---- per-step device state: dropout seed, global step t, Adam hyper-parameters -------------
 * replaces: python-side step counters of torch.optim.Adam (train_sr.py:480) and torch's global
 * RNG feeding nn.Dropout (model_seq.py:335,350,356). */
int amid_step_state_bytes(void);
int amid_step_state_pack(void* host_buf, unsigned long long seed, long long step, double lr, double beta1, double beta2, double eps);
int amid_step_begin(void* step_state, void* stream);                       /* step += 1 */

/* ---- K1 embedding gather ---------------------------------------------------------------------
 * replaces: embItemLayerEnhance.forward = nn.Embedding lookup, model_seq.py:27-29, called at
 * :418-421 (SASRec) / :279-282 (BERT4Rec).  idx may be int64 (LongTensor, train_sr.py:191-199) or int32.
 * err_flag (optional int): bit 0 set when an index is outside [0, n_rows) (the row is read as 0). */
int amid_gather_rows_f32(const float* table, long long n_rows, int D, const void* idx, int idx_is_i64, long long n_idx,
                         float* out, int* err_flag, void* stream);
/* concatenate + narrow the four index tensors of a batch: idx_all = [seq_d1 | seq_d2 | (i_node, neg)[b]] */
/* step_state_to_bump (optional): the same launch performs amid_step_begin (one launch fewer per step) */
int amid_pack_indices(const long long* i_node, const long long* neg, const long long* seq_d1, const long long* seq_d2,
                      int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag, void* step_state_to_bump, void* stream);
/* the same with the epoch's batches resident in HBM (train_sr.py:185-199: the reference moves each batch with .cuda() inside the
 * loop): `pool` holds n_pool packed batch images of `pool_stride` int64 words ([i_node B][neg B n_neg][seq_d1 B T][seq_d2 B T]
 * [tail words]); the launch consumes image (step + phase) mod n_pool by the DEVICE step counter, mirrors its in_words words into
 * `in_pack` and bumps the step (amid_step_begin folded in): a replayed graph walks the pool without a per-step copy. */
int amid_pack_indices_pool(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack,
                           int in_words, int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag,
                           void* step_state, void* stream);
/* fused gather + positional add + embedding dropout + feature-level (==0) mask.
 * replaces: model_seq.py:418-421 + Log2feats.forward :361-366.  pos0/pos1 = sac{1,2}.pos_emb.weight
 * (both NULL: plain gather for all rows, BERT4Rec).  xg: [2M + n_item_rows, D]; tmq: [2M, D/4] bytes. */
int amid_embed_fwd_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                       int n_item_rows, float* xg, unsigned char* tmq, const void* step_state, int train, float p_drop, void* stream);
/* backward of the above on the seq rows, in place on dxg [2M(+items), D]; the batch is cut into nsplit row ranges, each
 * writing its partial d pos_emb.weight: dpos_part [nsplit][2][T][D] (summed by amid_reduce_partials_f32).
 * replaces: autograd of model_seq.py:361-366 (EmbeddingBackward of pos_emb). */
int amid_embed_bwd_f32(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part,
                       const void* step_state, int train, float p_drop, void* stream);

/* ---- index sort / unique (no reference counterpart: enables the sparse gradient path) ------ */
/* workspace: amid_sort_unique_workspace_bytes(n_idx) bytes, ZERO-FILLED once before its first use (the four-launch sort keeps two
 * tickets there and leaves them zero), used by one call at a time. */
long long amid_sort_unique_workspace_bytes(int n_idx);
/* lists of at least n_idx indices (default 65536) and keys below 2^24 are sorted in four launches, shorter ones by 8-bit passes
 * (3 launches per pass + run detection): same outputs.  Returns the previous threshold; a negative argument only queries. */
int amid_sort_set_four_launch_min(int n_idx);
int amid_sort_unique_i32(const int* idx, int n_idx, long long n_rows, void* workspace, int* pos_sorted, int* uniq_ids,
                         int* seg_off /* [n_idx + 1] */, int* seg_of /* [n_idx] run index of each sorted entry */,
                         int* n_uniq /* device scalar */, void* stream);
/* The same sort as a PLAN for riders: nothing is launched; the train-step launches that take (plan, phase) run it as extra workgroups
 * in front of their own, so the step's sort costs no launch, no side stream, no fork and no join.  Five phases, each in a later
 * launch of the same stream than the one before, each launch carrying the phase it is built for: 1 (digit-0 counts)
 * amid_lazy_adam_catchup_positions_sort_f32, 2 (scatter 0) amid_sas_strip_ffn_bwd_sort_f32, 3 (digit-1 counts) / 4 (scatter 1)
 * amid_sas_strip_qkv_bwd_sort_f32 with / without the fused feed-forward backward, 5 (run heads) amid_embed_bwd_sort_f32; any other
 * phase number: AMID_ERR_UNSUPPORTED.  host_buf: amid_sort_plan_bytes() bytes; rows optional (as amid_sort_unique_rows_i32);
 * workspace as above.  Keys below 2^24 (digits of up to 12 bits: the rider workgroups run the 4 096-bin build on the head of their host
 * kernel's dynamic LDS; round 6 -- before: below 2^20); AMID_ERR_UNSUPPORTED beyond, and for lists of more than 256 supertiles. */
int amid_sort_plan_bytes(void);
int amid_sort_plan_pack(void* host_buf, const int* idx, const int* rows, int n_idx, long long n_rows, void* workspace, int* pos_sorted,
                        int* uniq_ids, int* seg_off, int* seg_of, int* n_uniq);
/* amid_sort_unique_i32 with a payload: pos_sorted holds rows[i] instead of i (ties keep the order of the list). */
int amid_sort_unique_rows_i32(const int* idx, const int* rows, int n_idx, long long n_rows, void* workspace, int* pos_sorted, int* uniq_ids,
                              int* seg_off, int* seg_of, int* n_uniq, void* stream);

/* ---- K3 embedding gradient as segment reduce -------------------------------------------------
 * replaces: autograd EmbeddingBackward (dense index_add into zero-filled [n_rows, D]) of the four
 * lookups at model_seq.py:418-421, run by loss.backward() train_sr.py:214. */
long long amid_segreduce_workspace_bytes(int n_idx, int D);
int amid_embgrad_segreduce_f32(const float* grad_rows /* [n_idx, D] */, const int* pos_sorted, const int* seg_off, const int* seg_of,
                               int n_idx, int D, void* workspace, float* uniq_grad /* [n_idx, D] */, void* stream);

/* data-parallel exchange helpers (no reference counterpart: the reference is single-GPU, train_sr.py:473).
 * pad: a rank's segment-reduced (ids, rows) padded to n_out entries with (pad_id, zero row) pairs (pad_id < 0: repeat the first id);
 * n_uniq is a device scalar.
 * merge: `world` (<= 16) lists of `len` non-decreasing keys (ascending unique ids, then `sentinel` > every id as padding), rank r's
 * list at keys + r * key_stride, merged stably in rank order -> the outputs of amid_sort_unique_i32 on the concatenation, except that
 * pos_sorted holds row_base + r * row_stride + i for entry (r, i): the row of its gradient in the caller's gathered buffer (packed
 * exchange: ids and rows of a rank travel in one buffer); the sentinel run is left out of n_uniq.
 * workspace: amid_sort_unique_workspace_bytes(world * len). */
int amid_sparse_pad_f32(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int n_out, int D, int pad_id, int* out_ids,
                        float* out_rows, void* stream);
int amid_merge_sorted_lists_i32(const int* keys, int world, int len, long long key_stride, int row_base, int row_stride, int sentinel,
                                void* workspace, int* pos_sorted, int* uniq_ids, int* seg_off, int* seg_of, int* n_uniq, void* stream);
/* the two helpers with `n_entries` fixed-order sums (tables of amid_reduce_entry_pack, semantics of amid_reduce_partials_f32) riding in
 * their first launch: the data-parallel step copies its flat dense gradient behind the padded rows (pad) and sums the ranks' dense
 * parts in rank order (merge) without a launch of their own -- both are independent of the sparse work they ride with. */
int amid_sparse_pad_sum_f32(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int n_out, int D, int pad_id, int* out_ids,
                            float* out_rows, const void* entries_dev, int n_entries, int max_count, void* stream);
int amid_merge_sorted_lists_sum_i32(const int* keys, int world, int len, long long key_stride, int row_base, int row_stride, int sentinel,
                                    void* workspace, int* pos_sorted, int* uniq_ids, int* seg_off, int* seg_of, int* n_uniq,
                                    const void* entries_dev, int n_entries, int max_count, void* stream);

/* Owner-bucketed sparse exchange (the alternative for large unique-row counts; the reference has no multi-GPU
 * path to mirror: train_sr.py:473).  count: counts[o] = how many of the first n_uniq (device scalar, <= cap) ascending unique ids
 * have id % world == o (o < world <= 16); counts[world] = 0 (the fill's overflow flag).  workspace: amid_owner_workspace_bytes(cap),
 * shared by the two calls.  buckets: the stable split of (ids, rows) into `world` packed chunks at out + o * chunk_floats, each
 * [id_rows rows of D floats holding bmax int32 ids | bmax gradient rows] -- the layout amid_merge_sorted_lists_i32 reads; slots past
 * an owner's count carry `sentinel` and a zero row; an owner with more than bmax entries sets counts[world] = 1 (nothing is written
 * out of range). */
long long amid_owner_workspace_bytes(int cap);
int amid_owner_count_i32(const int* uniq_ids, const int* n_uniq, int cap, int world, void* workspace, int* counts, void* stream);
int amid_owner_buckets_f32(const int* uniq_ids, const float* uniq_rows, const int* n_uniq, int cap, int D, int world, int bmax, int sentinel,
                           const void* workspace, float* out, long long chunk_floats, int id_rows, int* counts, void* stream);

/* amid_embgrad_segreduce_f32 + amid_reduce_partials_f32 with their first phases in ONE launch (the two independent ends of backward
 * side by side without a second stream) */
int amid_grad_tail_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                       void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count, const int* blk_off,
                       int total_blocks, void* stream);
/* blk_off (optional, device, [n_entries + 1] ascending from 0; total_blocks = its last value): entry e's sums run on the blocks
 * [blk_off[e], blk_off[e + 1]) -- sized by the caller to the entry (1024 elements per block when it has at most 32 aligned
 * partials, 128 otherwise) -- instead of max_count / 128 blocks for every entry. */
/* amid_grad_tail_f32 that also packs this rank's chunk of the data-parallel exchange in its second launch: uniq_grad points INTO the
 * chunk (the segment reduce writes the rows in place), out_ids[0, n_out) <- the first n_uniq unique ids then pad_id, and
 * dense_dst[0, dense_n) <- dense_src (the flat dense gradient the first launch has just reduced); 16-byte aligned dense pointers;
 * dense_n = 0: no dense part (the caller all-reduces the dense gradient itself).  The caller's bound n_out must cover the step's
 * unique rows: when *n_uniq > n_out the rows have overrun the chunk's row part and err_flag (optional) gets AMID_FLAG_UMAX_EXCEEDED. */
int amid_grad_tail_pack_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                            void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count, const int* uniq_ids,
                            const int* n_uniq, int n_out, int pad_id, int* out_ids, const float* dense_src, float* dense_dst,
                            long long dense_n, int* err_flag, const int* blk_off, int total_blocks, void* stream);

/* ---- K4 optimizer ----------------------------------------------------------------------------
 * replaces: torch.optim.Adam(model.parameters(), lr).step(), train_sr.py:480, :215 (dense over the table).
 * lazy rows: m, v [n_rows, D], last [n_rows] int32 (0 = never touched). */
int amid_lazy_adam_catchup_f32(float* table, float* m, float* v, int* last, const int* uniq_ids, const int* n_uniq, int n_uniq_max,
                               int D, const void* step_state, void* stream);
/* catch-up driven by the raw (non-unique) index list: needs no sort, so the sort can overlap the forward pass */
int amid_lazy_adam_catchup_positions_f32(float* table, float* m, float* v, int* last, const int* idx, int n_idx, int D,
                                         const void* step_state, void* stream);
/* ... carrying phase 1 of a sort plan (amid_sort_plan_pack) as extra workgroups in front of its own -- or, sort_phase = 6, ALL FIVE phases
 * chained in this one launch (the riders meet at a barrier of their own between the phases): for steps without five launches to ride in, whose
 * sort ran as a dozen small launches on a side stream.  The chain takes plans with keys below 2^20 over at most
 * amid_sort_chain_max_indices() indices; AMID_ERR_UNSUPPORTED otherwise. */
int amid_sort_chain_max_indices(void);
int amid_lazy_adam_catchup_positions_sort_f32(float* table, float* m, float* v, int* last, const int* idx, int n_idx, int D,
                                              const void* step_state, const void* sort_plan, int sort_phase, void* stream);
int amid_lazy_adam_apply_f32(float* table, float* m, float* v, int* last, const int* uniq_ids, const int* n_uniq, int n_uniq_max,
                             const float* uniq_grad, float grad_scale, int D, const void* step_state, void* stream);
int amid_lazy_adam_flush_f32(float* table, float* m, float* v, int* last, long long n_rows, int D, const void* step_state, void* stream);
int amid_adam_dense_f32(float* p, float* m, float* v, const float* g, long long n, float grad_scale, const void* step_state, void* stream);
/* amid_adam_dense_f32 + amid_lazy_adam_apply_f32 as ONE launch */
int amid_optimizer_step_f32(float* p, float* m, float* v, const float* g, long long n, float* table, float* m_tab, float* v_tab,
                            int* last, const int* uniq_ids, const int* n_uniq, int n_uniq_max, const float* uniq_grad, int D,
                            float grad_scale, const void* step_state, void* stream);
/* The data-parallel optimizer as ONE launch over the world's gathered chunks (the reference has no multi-GPU path: train_sr.py:473).
 * Chunk r at gathered + r * chunk_floats = [id_rows rows of D floats holding umax ascending unique int32 ids, padded with `sentinel`
 * (> every id) | umax gradient rows | at dense_off: the rank's flat dense gradient, n floats].  g <- the ranks' dense parts summed in
 * rank order, then dense Adam; every table row whose id occurs in some chunk gets the lazy row Adam with the rows of equal ids
 * summed in rank order.  world <= 16; dense_off and chunk_floats multiples of 4.  dense_off < 0: the chunks carry no dense part and
 * g already holds the world's summed dense gradient (the caller's RCCL all-reduce). */
int amid_optimizer_step_gathered_f32(float* p, float* m, float* v, float* g, long long n, float* table, float* m_tab, float* v_tab,
                                     int* last, const float* gathered, int world, int umax, long long chunk_floats, int id_rows,
                                     long long dense_off, int D, int sentinel, float grad_scale, const void* step_state, void* stream);

/* ---- SASRec encoder layer, forward ------------------------------------------------------------
 * Pointer-array arguments are HOST arrays of 2 device pointers (domain 0, domain 1).
 * mma_bf16 (D = 128 only): 0 = exact fp32 matrix products (v_mfma_f32_16x16x4_f32); 1 = the products' operands rounded to bf16 with
 * fp32 accumulation (v_mfma_f32_16x16x32_bf16), everything else fp32 -- BASELINE.json configs[2], checked against the fp32 oracle
 * at 2e-2 relative. */
int amid_rows_per_tile(int M);
/* replaces: attention_layernorms[i] + the packed in-projection of nn.MultiheadAttention, model_seq.py:373-374 */
int amid_sas_qkv_fwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* const* w_in,
                         const float* const* b_in, float ln_eps, int M, int D, int rows_per_tile, float* qn, float* q, float* k, float* v,
                         int mma_bf16, void* stream);
/* replaces: out_proj of nn.MultiheadAttention + "seqs = Q + mha_outputs" + forward_layernorms[i], model_seq.py:374-381 */
int amid_sas_oproj_fwd_f32(const float* o, const float* const* w_o, const float* const* b_o, const float* qn, const float* const* ln_w,
                           const float* const* ln_b, float ln_eps, int M, int D, int rows_per_tile, float* r, float* y, int mma_bf16, void* stream);
/* replaces: PointWiseFeedForward.forward model_seq.py:322-326 + "seqs *= ~timeline_mask" :383 */
int amid_sas_ffn_fwd_f32(const float* y, const float* const* w1, const float* const* b1, const float* const* w2, const float* const* b2,
                         const unsigned char* tmq, int M, int D, int rows_per_tile, int layer, const void* step_state, int train,
                         float p_drop, float* h, float* xo, int mma_bf16, void* stream);

/* amid_sas_oproj_fwd_f32 + amid_sas_ffn_fwd_f32 as ONE launch (the LN2 output stays on chip between the out-projection and the
 * feed-forward): replaces model_seq.py:374-383 after the attention core */
int amid_sas_oproj_ffn_fwd_f32(const float* o, const float* qn, const float* const* w_o, const float* const* b_o, const float* const* ln_w,
                               const float* const* ln_b, const float* const* w1, const float* const* b1, const float* const* w2,
                               const float* const* b2, const unsigned char* tmq, float ln_eps, int M, int D, int rows_per_tile, int layer,
                               const void* step_state, int train, float p_drop, float* r, float* y, float* h, float* xo, int mma_bf16,
                               void* stream);

/* amid_sas_oproj_ffn_fwd_f32 of layer l followed by amid_sas_qkv_fwd_f32 of layer l + 1 (n* arguments) on the same row tile, ONE launch */
int amid_sas_oproj_ffn_qkv_fwd_f32(const float* o, const float* qn, const float* const* w_o, const float* const* b_o,
                                   const float* const* ln_w, const float* const* ln_b, const float* const* w1, const float* const* b1,
                                   const float* const* w2, const float* const* b2, const unsigned char* tmq, float ln_eps, int M, int D,
                                   int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* r, float* y, float* h,
                                   float* xo, const float* const* nln_w, const float* const* nln_b, const float* const* nw_in,
                                   const float* const* nb_in, float* nqn, float* nq, float* nk, float* nv, int mma_bf16, void* stream);

/* ---- K2 attention core ------------------------------------------------------------------------
 * replaces: softmax(q k^T + mask) dropout v inside nn.MultiheadAttention (model_seq.py:374, causal=1) and
 * Attention.forward of BERT4Rec (model_seq.py:149-162, causal=0 with key_keep [B,T] from seq_d2 > 0, :288).
 * stats: [2M, H, 2] (row max, 1/row sum) saved for backward.  Causal, T <= 64, H <= 8, head dim 16 or 8 (D = 64 with the reference's 8
 * heads, train_sr.py:364: a 16-column tile is then a PAIR of heads) run on the matrix cores (csrc/attention_mfma.h); 64 < T <= 256 at head
 * dim 16 their blocked form; BERT4Rec's shape its own matrix-core kernels; everything else the general VALU kernels. */
int amid_attn_fwd_f32(const float* q, const float* k, const float* v, const unsigned char* key_keep, int B, int T, int D, int H, int causal,
                      int layer, const void* step_state, int train, float p_drop, float* o, float* stats, void* stream);
int amid_attn_bwd_f32(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                      const unsigned char* key_keep, int B, int T, int D, int H, int causal, int layer, const void* step_state, int train,
                      float p_drop, float* dq, float* dk, float* dv, void* stream);
/* backward with the loss's structure as a hint: row_domain [B] (= the batch's domain_id, train_sr.py:184) says that sequence (g, b)
 * has a non-zero d_o only if (row_domain[b] != 0) == g -- the other domain's BCE is multiplied by zero (train_sr.py:205-211) -- so
 * the matrix-core kernels store exact zeros for the other half without loading anything.  Same results as amid_attn_bwd_f32. */
int amid_attn_bwd_rows_f32(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o,
                           const unsigned char* key_keep, int B, int T, int D, int H, int causal, int layer, const void* step_state,
                           int train, float p_drop, float* dq, float* dk, float* dv, const long long* row_domain, void* stream);

/* ---- SASRec encoder layer, backward (autograd of model_seq.py:371-383 under loss.backward(), train_sr.py:214) */
int amid_transpose_weights_f32(const float* const* src, float* const* dst, int n /* <= 32 */, int D, void* stream);
int amid_sas_ffn_bwd_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                         const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int M, int D,
                         int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* dpre2, float* dpre1,
                         float* dr, float* d_o, float* ln_part /* [tiles][2][D] */, int mma_bf16, void* stream);
int amid_sas_qkv_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w,
                         const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int M, int D,
                         int rows_per_tile, float* dx, float* ln_part, int mma_bf16, void* stream);
/* amid_sas_qkv_bwd_f32 of layer l + 1 followed by amid_sas_ffn_bwd_f32 of layer l (f* arguments; its dxo is the dx just produced) on the
 * same row tile, ONE launch */
int amid_sas_qkv_ffn_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w,
                             const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int M, int D,
                             int rows_per_tile, float* dx, float* ln_part, const unsigned char* tmq, const float* fh, const float* fr,
                             const float* const* fln_w, const float* const* fw1T, const float* const* fw2T, const float* const* fwoT,
                             int flayer, const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr,
                             float* fd_o, float* fln_part, int mma_bf16, void* stream);
/* the six weight + bias gradients of n_layers (1 or 2) layers as split partials, ONE launch (two workgroups per CU): dy / x are host
 * arrays of 6 * n_layers device pointers, per layer in the order in_proj q, k, v, out_proj, conv1, conv2; w_part / b_part are host
 * arrays of n_layers device pointers to [2][6][splits][D*D] and [2][6][splits][D] */
/* mma_bf16 (D 128 only beyond 0): 0 = fp32 matrix instructions; 1 (compute = "bf16") = the products' operands rounded to bf16 on the way
 * into LDS, v_mfma_f32_16x16x32_bf16 with fp32 accumulation; 2 / 3 (compute = "fp32") = every fp32 operand as three bf16 pieces whose sum
 * is the operand exactly, nine / six piece pairs on the same instruction -- fp32 accuracy (3.8e-7 of the largest entry against the
 * fp64 product, 4.4e-7 for mode 0) at 46 us instead of 69.  Partial sums and the bias sums (taken from the unrounded rows) stay fp32 */
int amid_sas_wgrad_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits, float* const* w_part,
                       float* const* b_part, int mma_bf16, void* stream);
/* with the loss structure as a hint (row_domain [B] = the batch's domain_id, M = B * T): of domain g only the sequences b with
 * (row_domain[b] != 0) == g have non-zero dY rows (train_sr.py:205-211 multiplies the other domain's BCE by zero), and only those
 * rows are read -- the K dimension of every weight-gradient product halves.  Same results as amid_sas_wgrad_f32 (zeros left out). */
int amid_sas_wgrad_rows_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits, float* const* w_part,
                            float* const* b_part, const long long* row_domain, int B, int T, int mma_bf16, void* stream);
/* fixed-order reduction of partial buffers; entries are packed on the host then copied to the device by the caller */
int amid_reduce_entry_bytes(void);
int amid_reduce_entry_pack(void* host_buf, int index, const float* src, float* dst, long long stride, int n_part, int count);
int amid_reduce_partials_f32(const void* entries_dev, int n_entries, int max_count, void* stream);

/* ---- head: last LayerNorm + mean over time, scorer, masked BCE ------------------------------------
 * replaces: last_layernorm model_seq.py:385 + torch.mean(seq, 1) :432-434 (w0 == NULL: mean only, BERT4Rec :299-300) */
int amid_lnmean_fwd_f32(const float* x, const float* w0, const float* b0, const float* w1, const float* b1, int B, int T, int D, float eps,
                        float* u /* [2, B, D] */, void* stream);
int amid_lnmean_bwd_f32(const float* x, const float* du, const float* w0, const float* w1, int B, int T, int D, float eps, float* dx,
                        float* part /* [2B][2][D] */, void* stream);
/* replaces: predictModule.forward model_seq.py:40-54; with labels != NULL also the masked BCE mean of
 * train_sr.py:203-212 (per-row loss partials + dLoss/dp). */
int amid_scorer_fwd_f32(const float* u, const float* items, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* labels, const long long* domain_id, int B, int NI, int D, int hid, float* p1, float* p2, float* dp1,
                        float* dp2, float* loss_part, void* stream);
long long amid_scorer_part_floats(int D, int hid);
int amid_scorer_bwd_f32(const float* u, const float* items, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* p1, const float* p2, const float* dp1, const float* dp2, int B, int NI, int D, int hid, float* du,
                        float* ditems, float* part /* [B][amid_scorer_part_floats] */, int accumulate /* 1: du, ditems += */, void* stream);
/* doubly-robust objectives (next-4 of SURVEY.md 8(f)); replaces: train_sr_dr.py:216-221 (mode 0: loss_cls + dr_e_w * loss_dr_e) and
 * :392-394 (mode 1: loss_dr_r, needs ob_label [B]) and their autograd down to the three heads' outputs.  p / ips / g and the gradient
 * outputs are HOST arrays of 2 device pointers ([B, NI] each: domain-1 head, domain-2 head); loss_part [B][3] = per-row partials of
 * (loss_cls, loss_dr_e, loss_dr_r), already divided by B * NI */
int amid_dr_loss_f32(const float* const* p, const float* const* ips, const float* const* g, const float* labels,
                     const long long* domain_id, const long long* ob_label, int mode, float dr_e_w, int B, int NI, float* const* dp,
                     float* const* dips, float* const* dg, float* loss_part, void* stream);
int amid_sum_vector_f32(const float* v, int n, float* out, void* stream);
/* fused head, one workgroup per batch row: last LayerNorm + mean over T (model_seq.py:385, :432-434) -> predictModule
 * (model_seq.py:40-54) -> masked BCE partials + dLoss/dp (train_sr.py:203-212); ln_w / ln_b: host arrays of 2 device pointers
 * (NULL arrays: plain mean, BERT4Rec model_seq.py:299-300) */
int amid_head_fwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                      const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id, int B, int T,
                      int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2, float* loss_part,
                      void* stream);
/* backward of the head (scorer -> mean -> last LayerNorm); extra workgroups transpose n_tr (<= 32) square D x D weight
 * matrices (tr_src / tr_dst: host arrays of device pointers) for the backward GEMMs of the encoder */
int amid_head_bwd_f32(const float* x, const float* const* ln_w, const float* u, const float* items, const float* w1, const float* b1,
                      const float* w2, const float* b2, const float* p1, const float* p2, const float* dp1, const float* dp2, int B, int T,
                      int NI, int D, int hid, float eps, float* dx, float* ditems, float* ln_part, float* sc_part,
                      const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream);
/* amid_head_fwd_f32 (with labels) immediately followed by amid_head_bwd_f32 in ONE launch: the head of a training step (the backward half
 * reuses the forward half's LDS state) */
int amid_head_fwd_bwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                          const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id, int B, int T,
                          int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2, float* loss_part,
                          float* dx, float* ditems, float* ln_part, float* sc_part, const float* const* tr_src, float* const* tr_dst,
                          int n_tr, void* stream);
/* evaluation (next-2 of SURVEY.md 8(f)): rank (0 = best) of the positive, column 0, among the NI scores of a row, read from the
 * head of the row's own domain.  replaces: choose_predict utils.py:21-40 + the double argsort of get_sample_scores utils.py:296-297
 * after "pred[:, 0] -= fix_value" (train_sr.py:114-115: a tie counts against the positive; fix_value = 0: a tie favours it, as a
 * stable sort would) */
int amid_positive_rank_f32(const float* p1, const float* p2, const long long* domain_id, int B, int NI, float fix_value, int* rank,
                           void* stream);
/* replaces: torch.nn.LayerNorm(D, eps) applied row-wise (last_layernorm of a standalone Log2feats, model_seq.py:385) */
int amid_layernorm_rows_f32(const float* x, const float* w, const float* b, long long rows, int D, float eps, float* y, void* stream);

/* ---- BERT4Rec encoder block (hidden 128, 4 heads, FFN 512, dropout 0.1: hard-coded by the reference, model_seq.py:264-267)
 * replaces: TransformerBlock.forward model_seq.py:242-245 (SublayerConnection :140-142, LayerNorm :124-127 with unbiased std and
 * eps added to the std, MultiHeadedAttention :183-196, PositionwiseFeedForward :216-217, GELU :204) and its autograd.
 * Pointer-array parameters are HOST arrays of device pointers: [domain] or, for the three q/k/v projections, [which * 2 + domain].
 * The attention core in between is amid_attn_{fwd,bwd}_f32 with causal = 0 and key_keep from seq_d2 > 0 (model_seq.py:288). */
int amid_bert_qkv_fwd_f32(const float* x, const float* const* ln_a, const float* const* ln_b, const float* const* w3x2,
                          const float* const* b3x2, int M, int rows_per_tile, float* y, float* q, float* k, float* v, void* stream);
int amid_bert_oproj_fwd_f32(const float* o, const float* x, const float* const* w, const float* const* b, int M, int rows_per_tile, int layer,
                            const void* step_state, int train, float p_drop, float* x1, void* stream);
int amid_bert_ffn1_fwd_f32(const float* x1, const float* const* ln_a, const float* const* ln_b, const float* const* w1, const float* const* b1,
                           int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* y2, float* pre,
                           float* h, void* stream);
int amid_bert_ffn2_fwd_f32(const float* h, const float* x1, const float* const* w2, const float* const* b2, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* x2, void* stream);
int amid_bert_ffn2_bwd_f32(const float* dx2, const float* pre, const float* const* w2T, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* dz, float* dpre, void* stream);
int amid_bert_ffn1_bwd_f32(const float* dpre, const float* dx2, const float* x1, const float* const* ln_a, const float* const* w1T,
                           const float* const* woT, int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop,
                           float* dx1, float* dt, float* d_o, float* ln_part, void* stream);
int amid_bert_qkv_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x, const float* const* ln_a,
                          const float* const* wT3x2, int M, int rows_per_tile, float* dx, float* ln_part, void* stream);
/* the three backward kernels over the live sequences only (the step's own loss sends no gradient into the other domain's encoder
 * of a sample, train_sr.py:205-211): live = amid_live_list_i32's list of the batch, M = B * T, rows_per_tile a multiple of T -- a
 * tile holds rows_per_tile / T whole live sequences of one domain, gathered from wherever they sit in the batch (half the tiles;
 * these kernels are bound by the weights every tile streams).  qkv also writes the exact-zero dx rows of its sequences'
 * counterparts in the other domain; LayerNorm partials: 2 * ceil(B / (rows_per_tile / T)) slots. */
/* n_ent (<= 12) weight-gradient tiles of 128 x 128 as split partials: w_part [2][n_ent][splits][128*128], b_part [2][n_ent][splits][128];
 * tile e lands in w_part[domain][out_group[e]][split] at column out_col[e] with row stride out_ld[e] (standalone tile: 128, e, 0; the
 * out_ld/128 tiles of one [128, out_ld] matrix share out_group so its partials are one contiguous [splits][128*out_ld] block) */
int amid_bert_wgrad_f32(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                        const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part, float* b_part, void* stream);
/* key mask keep[b][t] = seq[b][t] > 0, replaces: mask = (seq_d2 > 0)... model_seq.py:288 */
int amid_key_keep_u8(const long long* seq, long long n, unsigned char* keep, void* stream);
int amid_transpose_rect_f32(const float* const* src, float* const* dst, const int* rows, const int* cols, int n, void* stream);

/* ---- data pipeline on the device (next-3 of SURVEY.md 8(f)) ------------------------------------------------------------------
 * replaces: random.sample(item_pool_d - set(own sequence), k) per sample in DualDomainSeqDataset.__getitem__ (dataset_seq.py:188,
 * :198, :206, :215): k distinct negatives per row, uniform over the row's domain pool minus its own items.  pool_d*: sorted unique
 * ids; own_items / own_off [N + 1]: the rows' own item ids, concatenated; out [N, k] (out[r][0] = -1: pool exhausted for row r). */
int amid_sample_negatives_i64(const long long* pool_d1, int n_pool_d1, const long long* pool_d2, int n_pool_d2,
                              const long long* own_items, const int* own_off, const long long* domain_id, int N, int k,
                              unsigned long long seed, unsigned epoch, long long* out, void* stream);

/* ---- InterComp on the SASRec path (isItC; next-1 of SURVEY.md 8(f)) -------------------------------------------------------
 * replaces: InterComp.forward model_seq.py:483-497 as used at model_seq.py:426-434 (both directions) and its autograd.  Since
 * SASRec only means over the time axis afterwards, the module collapses onto per-row means (derivation: csrc/intercomp.hip):
 *   pairmax : s[j] = max_{a,c} LN_last(x[0,j,a]) . LN_last(x[1,j,c])                                   (:488-489)
 *   mix_fwd : gate = softmax_batch(s) > threshold (:490-491); z_g = sum_j w_bs_g[j] gate_j u_raw[1-g][j];
 *             u_mix[g][b] = 0.5 u_raw[g][b] + 0.5 (W_nn_g z_g + b_nn_g sum_j w_bs_g[j] + b_bs_g)        (:492-495, :432-434)
 *   mix_bwd : gradients of the above into du_raw and the eight InterComp parameters (written whole, no partials).
 * Host pointer arrays hold 2 device pointers: index 0 = itc_d1 / sac1, 1 = itc_d2 / sac2.  u_raw: amid_lnmean_fwd_f32. */
int amid_itc_pairmax_f32(const float* x, const float* const* ln_w, const float* const* ln_b, int B, int T, int D, float eps, float* s,
                         float* u_raw /* optional [2, B, D]: also emit mean_t LN_last(x), saving the amid_lnmean_fwd_f32 launch */,
                         void* stream);
int amid_itc_mix_fwd_f32(const float* u_raw, const float* s, const float* const* w_nn, const float* const* b_nn,
                         const float* const* w_bs, const float* const* b_bs, float threshold, int B, int D, float* gate, float* z,
                         float* sw, float* u_mix, void* stream);
int amid_itc_mix_bwd_f32(const float* du_mix, const float* u_raw, const float* gate, const float* z, const float* sw,
                         const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B, int D, float* du_raw,
                         float* const* dw_nn, float* const* db_nn, float* const* dw_bs, float* const* db_bs, void* stream);

/* Up to three scorers forward + the row's loss terms + backward in ONE launch on given user vectors u [2, B, D]: the head of the
 * isItC / isDR train step (replaces predictModule / predict_ips / predict_gfunc forward model_seq.py:436-440, the objectives of
 * train_sr.py:203-212 or train_sr_dr.py:216-221 / :392-394, and their autograd).  Host arrays of n_heads device pointers: w1, b1,
 * w2, b2, p1, p2 (outputs), dp1, dp2 (gradients of the outputs, written here), sc_part (weight-gradient partials as
 * amid_scorer_bwd_f32).  n_heads 1: masked BCE (loss_part [B]); n_heads 3: objective `mode` of amid_dr_loss_f32 (dr_loss_part
 * [B][3]).  du [2, B, D] and ditems [B, NI, D] are sums over the heads.  tr_*: as amid_head_bwd_f32. */
int amid_scorer_multi_fwd_bwd_f32(const float* u, const float* items, const float* const* w1, const float* const* b1,
                                  const float* const* w2, const float* const* b2, int n_heads, const float* labels,
                                  const long long* domain_id, const long long* ob_label, int mode, float dr_e_w, int B, int NI,
                                  int D, int hid, float* const* p1, float* const* p2, float* const* dp1, float* const* dp2,
                                  float* loss_part, float* dr_loss_part, float* du, float* ditems, float* const* sc_part,
                                  const float* const* tr_src, float* const* tr_dst, int n_tr, void* stream);

/* ---- InnerComp on the SASRec path (isInC; next-1 of SURVEY.md 8(f)) -------------------------------------------------------
 * replaces: InnerComp.forward model_seq.py:459-472 as used at model_seq.py:422-424 (on the gathered rows, before the encoders,
 * which then run over 2T tokens with 2T-row pos_emb tables, :398-401) and its autograd.  Compute-once form (csrc/innercomp.hip):
 *   score     : s[g][j] = max_{a,c} e[g,j,a] . e[g,j,c]                                              (:463-464)
 *   embed_fwd : gate = softmax_batch(s[g]) > threshold (:465-466); S[g][t] = sum_j w_bs_g[j] gate_j e[g,j,t];
 *               Z[g][t] = W_nn_g S[g][t] + b_nn_g sum_j w_bs_g[j] + b_bs_g (:467-469); then the encoder input
 *               x0[g,b,0..2T) = [e[g,b,:] | Z[g]] + P_g[0..2T), dropout, (==0) mask (:470, :361-366); tmq as amid_embed_fwd_f32
 *   bwd       : after amid_embed_bwd_f32(dx0, tmq, B, 2T, ...) -- whose pos_emb partials of rows T..2T-1 are dZ's partials --:
 *               InnerComp parameter gradients (written whole) and dxg[g,j,t] = dx0[g,j,t] + w_bs_g[j] gate_j (dZ[g][t] W_nn_g).
 * xg: [2, B, T, D] plain gathered rows (amid_gather_rows_f32); host pointer arrays hold 2 device pointers (inc_d1, inc_d2). */
int amid_inc_score_f32(const float* xg, int B, int T, int D, float* s, void* stream);
int amid_inc_embed_fwd_f32(const float* xg, const float* s, const float* const* w_nn, const float* const* b_nn,
                           const float* const* w_bs, const float* const* b_bs, float threshold, const float* pos0,
                           const float* pos1, int B, int T, int D, float* gate, float* S, float* Z, float* sw, float* x0,
                           unsigned char* tmq, const void* step_state, int train, float p_drop, void* stream);
int amid_inc_bwd_f32(const float* dpos_part, int nsplit, const float* xg, const float* dx0, const float* gate, const float* S,
                     const float* sw, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B, int T,
                     int D, float* dZ, float* dS, float* rows, float* const* dw_nn, float* const* db_nn, float* const* dw_bs,
                     float* const* db_bs, float* dxg, void* stream);
/* a data-parallel shard of InnerComp's batch (B rows = samples j0 .. j0 + B - 1 of the Bg = len(trans_bs.weight) rows the module spans,
 * model_seq.py:457): s_all [2, Bg] = the ranks' amid_inc_score_f32 outputs all-gathered per domain.  phase 1: gates, this shard's partial S
 * [2, T, D], sw; the caller all-reduces S; phase 2: Z and the encoder input.  Backward: phase 1 = this shard's dZ [2, T, D]; the caller
 * all-reduces it; phase 2 = dS, the parameter gradients (W_nn, b_nn, b_bs times gscale = 1 / world: every rank computes them alike and
 * the dense exchange sums the ranks; w_bs: this shard's slice, zeros elsewhere) and the table-row gradients */
int amid_inc_embed_fwd_shard_f32(const float* xg, const float* s_all, const float* const* w_nn, const float* const* b_nn,
                                 const float* const* w_bs, const float* const* b_bs, float threshold, const float* pos0, const float* pos1,
                                 int B, int T, int D, int Bg, int j0, int phase, float* gate, float* S, float* Z, float* sw, float* x0,
                                 unsigned char* tmq, const void* step_state, int train, float p_drop, void* stream);
int amid_inc_bwd_shard_f32(const float* dpos_part, int nsplit, const float* xg, const float* dx0, const float* gate, const float* S,
                           const float* sw, const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int B, int T, int D,
                           int Bg, int j0, int phase, float gscale, float* dZ, float* dS, float* rows, float* const* dw_nn,
                           float* const* db_nn, float* const* dw_bs, float* const* db_bs, float* dxg, void* stream);

/* ---- the comp modules in front of BERT4Rec's encoders (isInC model_seq.py:283-286, isItC :289-294) -------------------------
 * replaces: InnerComp.forward (:459-472) / InterComp.forward (:483-497) as BERT4Rec calls them -- on the gathered rows, before
 * the transformer blocks, which then see 2T tokens -- and their autograd.  Same compute-once token group as above without
 * positional rows or input dropout; cross = 1 (InterComp): module g scores row j by max_{a,c} e[g,j,a] . e[1-g,j,c] and mixes
 * the OTHER domain's rows, S[g][t] = sum_j w_bs_g[j] gate_g[j] e[1-g,j,t]; cross = 0 (InnerComp): its own.
 *   score : s [2, B];   fwd : gate, S, Z, sw as amid_inc_embed_fwd_f32 and x0[g,b,0..2T) = [e[g,b,:] | Z[g]]
 *   bwd   : dZ[g][t] = sum_b dx0[g,b,T+t] (fixed order), the modules' parameter gradients (written whole) and
 *           dxg[sd,j,t] = dx0[sd,j,t] + w_bs_g[j] gate_g[j] (dZ[g][t] W_nn_g),  sd = cross ? 1-g : g
 * amid_key_keep_tiled_u8: the key mask of :286 / :294, keep[b][r*T + t] = seq[b][t] > 0 for r < reps. */
int amid_bert_comp_score_f32(const float* xg, int B, int T, int D, int cross, float* s, void* stream);
int amid_bert_comp_fwd_f32(const float* xg, const float* s, const float* const* w_nn, const float* const* b_nn,
                           const float* const* w_bs, const float* const* b_bs, float threshold, int cross, int B, int T, int D,
                           float* gate, float* S, float* Z, float* sw, float* x0, void* stream);
int amid_bert_comp_bwd_f32(const float* xg, const float* dx0, const float* gate, const float* S, const float* sw,
                           const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int cross, int B, int T,
                           int D, float* dZ, float* dS, float* rows, float* const* dw_nn, float* const* db_nn, float* const* dw_bs,
                           float* const* db_bs, float* dxg, void* stream);
int amid_key_keep_tiled_u8(const long long* seq, int B, int T, int reps, unsigned char* keep, void* stream);

/* ---- the BERT4Rec row-tile entry points built with shorter tiles (3 / 4 / 5 MFMA row tiles per workgroup instead of 7) ----------------
 * Identical signatures and semantics; callers use *_rt3 when rows_per_tile <= 48 (seq_len 20 at batch 256: the mybank shape of
 * BASELINE.json configs[3]), *_rt4 when <= 64 and *_rt5 when <= 80, where the 112-row build would spend much of its matrix work on zero
 * rows (csrc/tile_gemm.h).  (The SASRec row-tile kernels exist in the 112-row build only: every SASRec shape of BASELINE.json runs the
 * strip kernels; the row-tile kernels remain for bf16 operands beyond 64 tokens and for activations beyond 2 GiB.) */
int amid_bert_qkv_fwd_f32_rt3(const float* x, const float* const* ln_a, const float* const* ln_b, const float* const* w3x2,
                          const float* const* b3x2, int M, int rows_per_tile, float* y, float* q, float* k, float* v, void* stream);
int amid_bert_qkv_fwd_f32_rt5(const float* x, const float* const* ln_a, const float* const* ln_b, const float* const* w3x2,
                          const float* const* b3x2, int M, int rows_per_tile, float* y, float* q, float* k, float* v, void* stream);
int amid_bert_oproj_fwd_f32_rt3(const float* o, const float* x, const float* const* w, const float* const* b, int M, int rows_per_tile, int layer,
                            const void* step_state, int train, float p_drop, float* x1, void* stream);
int amid_bert_oproj_fwd_f32_rt5(const float* o, const float* x, const float* const* w, const float* const* b, int M, int rows_per_tile, int layer,
                            const void* step_state, int train, float p_drop, float* x1, void* stream);
int amid_bert_ffn1_fwd_f32_rt3(const float* x1, const float* const* ln_a, const float* const* ln_b, const float* const* w1, const float* const* b1,
                           int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* y2, float* pre,
                           float* h, void* stream);
int amid_bert_ffn1_fwd_f32_rt5(const float* x1, const float* const* ln_a, const float* const* ln_b, const float* const* w1, const float* const* b1,
                           int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* y2, float* pre,
                           float* h, void* stream);
int amid_bert_ffn2_fwd_f32_rt3(const float* h, const float* x1, const float* const* w2, const float* const* b2, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* x2, void* stream);
int amid_bert_ffn2_fwd_f32_rt5(const float* h, const float* x1, const float* const* w2, const float* const* b2, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* x2, void* stream);
int amid_bert_ffn2_bwd_f32_rt3(const float* dx2, const float* pre, const float* const* w2T, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* dz, float* dpre, void* stream);
int amid_bert_ffn2_bwd_f32_rt5(const float* dx2, const float* pre, const float* const* w2T, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* dz, float* dpre, void* stream);
int amid_bert_ffn1_bwd_f32_rt3(const float* dpre, const float* dx2, const float* x1, const float* const* ln_a, const float* const* w1T,
                           const float* const* woT, int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop,
                           float* dx1, float* dt, float* d_o, float* ln_part, void* stream);
int amid_bert_ffn1_bwd_f32_rt5(const float* dpre, const float* dx2, const float* x1, const float* const* ln_a, const float* const* w1T,
                           const float* const* woT, int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop,
                           float* dx1, float* dt, float* d_o, float* ln_part, void* stream);
int amid_bert_qkv_bwd_f32_rt3(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x, const float* const* ln_a,
                          const float* const* wT3x2, int M, int rows_per_tile, float* dx, float* ln_part, void* stream);
int amid_bert_qkv_bwd_f32_rt5(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x, const float* const* ln_a,
                          const float* const* wT3x2, int M, int rows_per_tile, float* dx, float* ln_part, void* stream);

/* ---- hipGraph capture / replay of a whole step; HIP events on the caller's stream ---------------- */
int amid_graph_capture_begin(void* stream);
int amid_graph_capture_end(void* stream, void** graph_exec_out);
int amid_graph_launch(void* graph_exec, void* stream);
int amid_graph_destroy(void* graph_exec);
int amid_event_create(void** ev_out);
int amid_event_record(void* ev, void* stream);
int amid_event_sync(void* ev);
int amid_event_elapsed_ms(void* start, void* stop, float* ms_out);
int amid_event_destroy(void* ev);


/* ---- the BERT4Rec block as register-resident STRIP kernels (csrc/bert_strip.hip; machinery: csrc/strip_gemm.h, strip_chain.h) -------------
 * replace: TransformerBlock.forward model_seq.py:242-245 (SublayerConnection :140-142, the reference LayerNorm :124-127,
 * MultiHeadedAttention's projections :183-196, PositionwiseFeedForward :216-217 with the tanh GELU :204) and its autograd
 * (loss.backward(), train_sr.py:214), like the amid_bert_*_f32 row-tile entry points, same operations, dropout counters and saved
 * tensors; hidden 128 / feed-forward 512 as hard-coded by the reference (:264-267).  Activations [2 * B * T, 128] and [2 * B * T, 512]
 * (domain 0's rows, then domain 1's), the wide ones at most 2 GiB (amid_bert_strip_supported).  live: as the amid_sas_strip_* entry
 * points (NULL = every sequence; else amid_live_list_i32: only the listed sequences are read / written).  Per-domain parameters: host
 * arrays of two device pointers; w3 / b3 / wT3: six, ordered [q, k, v][domain].  ln_part: [2 * ceil(B * T /
 * amid_sas_strip_tile_rows())][2][128], domain g owns the slots [g * tpg, (g + 1) * tpg), unused ones are zeroed. */
int amid_bert_strip_supported(int B, int T, int D);
int amid_bert_strip_qkv_fwd_f32(const float* x, const float* const* la, const float* const* lb, const float* const* w3,
                                const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k, float* v,
                                void* stream);
/* the same launch with the step's prologue riding as extra workgroups: key_keep[i] = seq_d2[i] > 0, i < n_keys (the ONE key mask of
 * both encoders, model_seq.py:288; seq_d2 == NULL: none) and tr_dst[m][c][r] = tr_src[m][r][c] for n_tr <= 24 matrices of
 * tr_rows[m] x tr_cols[m] floats (multiples of 64): what amid_key_keep_u8 and amid_transpose_rect_f32 do in launches of their own */
int amid_bert_strip_qkv_fwd_pro_f32(const float* x, const float* const* la, const float* const* lb, const float* const* w3,
                                    const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k,
                                    float* v, const long long* seq_d2, int n_keys, unsigned char* key_keep,
                                    const float* const* tr_src, float* const* tr_dst, const int* tr_rows, const int* tr_cols,
                                    int n_tr, void* stream);
/* nla != NULL: block l + 1's amid_bert_strip_qkv_fwd_f32 (n* arguments) continues on the block output x2 in registers */
int amid_bert_strip_oproj_ffn_fwd_f32(const float* o, const float* x, const float* const* wo, const float* const* bo,
                                      const float* const* la, const float* const* lb, const float* const* w1,
                                      const float* const* b1, const float* const* w2, const float* const* b2, int B, int T,
                                      const int* live, int layer, const void* step_state, int train, float p_drop, float* x1,
                                      float* y2, float* pre, float* h, float* x2, const float* const* nla,
                                      const float* const* nlb, const float* const* nw3, const float* const* nb3, float* ny,
                                      float* nq, float* nk, float* nv, void* stream);
/* w2T [512, 128], w1T [128, 512], woT [128, 128]: the transposed weights (amid_transpose_rect_f32) */
int amid_bert_strip_ffn_bwd_f32(const float* dx2, const float* pre, const float* x1, const float* const* la,
                                const float* const* w2T, const float* const* w1T, const float* const* woT, int B, int T,
                                const int* live, int layer, const void* step_state, int train, float p_drop, float* dz,
                                float* dpre, float* dx1, float* dt, float* d_o, float* ln_part, void* stream);
/* fpre != NULL: the block below's amid_bert_strip_ffn_bwd_f32 (f* arguments) continues on d x in registers; dx is then not written.
 * zero_dead (with a live list and dx): the rows of dx of the sequences NOT on the list are zero-filled (their gradient is exactly
 * zero, train_sr.py:205-211; the segment reduce of the table-row gradients reads every row) */
int amid_bert_strip_qkv_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x,
                                const float* const* la, const float* const* wT3, int B, int T, const int* live, float* dx,
                                int zero_dead, float* ln_part, const float* fpre, const float* fx1, const float* const* fla,
                                const float* const* fw2T, const float* const* fw1T, const float* const* fwoT, int flayer,
                                const void* step_state, int train, float p_drop, float* fdz, float* fdpre, float* fdx1,
                                float* fdt, float* fd_o, float* fln_part, void* stream);
/* the BERT4Rec attention core (bidirectional, ONE key mask from seq_d2 > 0 for both encoders model_seq.py:288, 4 heads of 32, T <= 64:
 * csrc/attention_mfma_bert.hip) over the sequences of a live list only */
int amid_attn_bert_live_supported(int T, int D, int H);
int amid_attn_bert_fwd_live_f32(const float* q, const float* k, const float* v, const unsigned char* key_keep, int B, int T, int D,
                                int H, int layer, const void* step_state, int train, float p_drop, float* o, float* stats,
                                const int* live, void* stream);
int amid_attn_bert_bwd_live_f32(const float* q, const float* k, const float* v, const float* o, const float* stats,
                                const float* d_o, const unsigned char* key_keep, int B, int T, int D, int H, int layer,
                                const void* step_state, int train, float p_drop, float* dq, float* dk, float* dv,
                                const int* live, void* stream);

/* ---- the three SASRec backward row-tile kernels over the LIVE sequences only ------------------------------------------------
 * The train step's own loss multiplies the other domain's BCE of every sample by zero (train_sr.py:205-211): of encoder g only the
 * sequences b with (row_domain[b] != 0) == g receive a gradient, everything else in its backward is exact zeros.  These entry points
 * take the batch's domain ids and tile those sequences' rows only (M = B * T; rows_per_tile counts live rows; ln_part needs
 * 2 * ceil(M / rows_per_tile) slots, the unused ones are zeroed); rows of the dead sequences are neither read nor written.
 * Same arguments as the plain entry points otherwise. */
int amid_sas_ffn_bwd_rows_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w, const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int M, int D, int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* dpre2, float* dpre1, float* dr, float* d_o, float* ln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream);
int amid_sas_qkv_bwd_rows_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream);
int amid_sas_qkv_ffn_bwd_rows_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w, const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int M, int D, int rows_per_tile, float* dx, float* ln_part, const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w, const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer, const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr, float* fd_o, float* fln_part, int mma_bf16, const long long* row_domain, int B, int T, void* stream);

/* ---- the whole encoder forward of a sequence in ONE launch (csrc/sasrec_seq.hip) ------------------------------------------------------
 * replaces: Log2feats.forward model_seq.py:371-383 for n_layers layers, the attention core of nn.MultiheadAttention (:374) included --
 * amid_sas_strip_qkv_fwd_f32 + amid_attn_fwd_f32 + amid_sas_strip_oproj_ffn_fwd_f32 per layer, with q / k / v / o never re-read.
 * Shapes: amid_sas_seq_supported (8 heads, T <= 64, D 128 -- or D 64, head dim 8: csrc/sasrec_seqn.hip's N-split builds only, fp32 products,
 * p_drop 0.5 or eval mode).  Per-domain parameter families: 2 * n_layers pointers ordered
 * [layer][domain]; saved-tensor families: n_layers pointers; x_in[l] = layer l's input rows (x_in[0] read, the others written),
 * xout = the last layer's output; stats [2 B T, H, 2]; live as for the strip kernels. */
int amid_sas_seq_supported(int B, int T, int D, int H);
int amid_sas_seq_fwd_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                         const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                         const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                         const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                         float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                         const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live, const void* step_state,
                         int train, float p_drop, void* stream);
/* Which build of the fused forward runs (diagnostics, tests, A/B measurements): 0 = auto (per shape), 1 = whole-row waves
 * (csrc/sasrec_seq.hip), 2 = the N-split build's default split (csrc/sasrec_seqn.hip: NS waves share a 16-row strip, each owning
 * D / NS columns), 42 / 22 / 24 / 14 / 18 = (strips per sequence, column parts) spelled out.  v < 0 only queries.  Returns the previous
 * value.  Process-wide host state, not a kernel launch. */
int amid_sas_seq_fwd_variant(int v);
/* The same forward with the twelve projections' matrix products on v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, fp32 accumulation;
 * LayerNorm, the attention core, residuals, dropout and everything stored stay fp32) -- BASELINE.json configs[2], arithmetic
 * model_seq.py:371-383.  w16 = amid_sas_weights_bf16 images of THIS step's weights, [layer][domain][q, k, v, o, conv1, conv2][D][D] bf16.
 * The fp32 weight arguments are still required (biases, and the fp32 builds' operands). */
int amid_sas_seq_fwd_bf16w_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                               const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                               const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                               const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                               float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                               const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                               const void* step_state, int train, float p_drop, const void* w16, void* stream);
/* bf16 fragment images of n (<= 48) square [D][D] fp32 matrices, row-major [out][in] (transposed != 0: of their transposes -- the
 * backward data-gradient products): dst16 [n][D][D] bf16 with the in-features of a row permuted into the order the bf16 MFMA's lanes
 * consume them (csrc/sasrec_seq.hip weights_bf16_kernel).  Once per step, before the forward. */
int amid_sas_weights_bf16(const float* const* src, int n, int D, int transposed, void* dst16, void* stream);
/* planes = 3: three images per matrix, dst16 [n][3][D][D] bf16 -- every element as hi + mid + lo, three bf16 pieces (each rounded to
 * nearest even) whose sum is the fp32 element exactly; planes = 1: amid_sas_weights_bf16 */
int amid_sas_weights_bf16_planes(const float* const* src, int n, int D, int transposed, int planes, void* dst16, void* stream);
/* amid_sas_seq_fwd_f32 (Log2feats.forward, model_seq.py:371-383) with the twelve projections' products on the bf16 matrix cores at FP32
 * ACCURACY: every operand element as three bf16 pieces, six piece pairs per product on v_mfma_f32_16x16x32_bf16 (96 x 16 cycles per wave
 * and product instead of 128 x 32); LayerNorm, the attention core, residuals, dropout and everything stored as in the fp32 build.
 * w16x3 = amid_sas_weights_bf16_planes(..., planes = 3, ...) images of THIS step's weights, [layer][domain][q, k, v, o, conv1, conv2][3][D][D] */
int amid_sas_seq_fwd_split_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                               const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                               const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                               const float* const* w2, const float* const* b2, float* const* qn, float* const* q, float* const* k,
                               float* const* v, float* const* o, float* const* stats, float* const* r, float* const* y, float* const* h,
                               const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                               const void* step_state, int train, float p_drop, const void* w16x3, void* stream);

/* ---- the encoder's data gradients of the live sequences in ONE launch (csrc/sasrec_strip.hip: seq_bwd_kernel) -------------------------
 * replaces: autograd of Log2feats.forward model_seq.py:371-383 under loss.backward() (train_sr.py:214) -- per layer, top down,
 * amid_sas_strip_ffn_bwd_f32 + amid_attn_bwd_live_f32 + amid_sas_strip_qkv_bwd_f32 as one workgroup-long chain per live sequence.
 * Shapes: amid_sas_seq_bwd_supported (8 heads, D 128 or 64, T <= 64; T <= 32 and D 64 run the N-split build of csrc/sasrec_seqn_bwd.hip -- two
 * strips x four column parts, or one x eight -- which needs p_drop = 0.5 or eval mode: AMID_ERR_UNSUPPORTED otherwise).  Saved tensors / gradient outputs / LayerNorm partials:
 * n_layers pointers each; parameters and transposed weights: 2 * n_layers pointers ordered [layer][domain].  dxo: gradient of the last
 * layer's output; dx: gradient of layer 0's input (rows of the live sequences; the others are not touched); d_o: scratch [2 B T, D].
 * ln1_part / ln2_part[l]: [2 B][2][D] -- domain g's slots are [g B, (g + 1) B): its live sequences' partial sums first, then zeros.
 * live: amid_live_list_i32 (required).  The weight gradients (amid_sas_wgrad_rows_f32) read dq / dk / dv / dr / dpre1 / dpre2 as before. */
int amid_sas_seq_bwd_supported(int B, int T, int D, int H);
/* Which build of the one-launch backward runs (diagnostics, tests, A/B measurements): 0 = auto (= 1: the two measure the same), 1 = a wave per 16-row strip
 * (csrc/sasrec_strip.hip seq_bwd_kernel), 2 = the N-split build (csrc/sasrec_seqn_bwd.hip: eight waves per sequence, two per strip with
 * half the columns each, a wave per head in the attention core; same bits -- at T > 32, D 128 that build is compiled into the diagnostic
 * library only, amid_diag_variants() = 1, and the product answers 2 with build 1).  v < 0 only queries.  Returns the previous value.  Host state. */
int amid_sas_seq_bwd_variant(int v);
int amid_diag_variants(void);      /* 1: a diagnostic build of the library (profiles/tools/build_diag.sh, -DAMID_DIAG_VARIANTS); the product: 0 */
int amid_sas_seq_bwd_f32(int n_layers, const float* dxo, const unsigned char* tmq, const float* const* h, const float* const* r,
                         const float* const* x, const float* const* q, const float* const* k, const float* const* v,
                         const float* const* o, const float* const* stats, const float* const* ln1_w, const float* const* ln2_w,
                         const float* const* wqT, const float* const* wkT, const float* const* wvT, const float* const* woT,
                         const float* const* w1T, const float* const* w2T, float ln_eps, int B, int T, int D, int H, const int* live,
                         const void* step_state, int train, float p_drop, float* const* dpre2, float* const* dpre1, float* const* dr,
                         float* d_o, float* const* dq, float* const* dk, float* const* dv, float* dx, float* const* ln1_part,
                         float* const* ln2_part, int mma_bf16, void* stream);

/* ---- the fused train step over the LIVE sequences -------------------------------------------------------------------------------
 * train_sr.py:205-211 multiplies the BCE terms of domain 1 - domain_id[b] of every sample b by zero: of the 2 B sequences a step
 * encodes only the B "live" ones (domain_id[b], b) reach the loss.  amid_live_list_i32 lists them ([B + 1] ints: batch rows of
 * domain 0's live sequences ascending, then domain 1's, then n0); the *_live / *_own entry points walk that list only and leave
 * every row of the other sequences untouched.  model.forward (which returns both domains' logits) never uses them. */
int amid_live_list_i32(const long long* domain, int B, int* live, void* stream);
/* amid_pack_indices / amid_pack_indices_pool with the live list written by the same launch (the pool images carry the batch's
 * domain ids in the B words behind the index words: SasrecEngine.pack_batch) */
int amid_pack_indices_live(const long long* i_node, const long long* neg, const long long* seq_d1, const long long* seq_d2, int B, int T,
                           int n_neg, long long n_rows, int* idx_all, int* err_flag, void* step_state_to_bump, const long long* domain,
                           int* live, void* stream);
int amid_pack_indices_pool_live(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack,
                                int in_words, int B, int T, int n_neg, long long n_rows, int* idx_all, int* err_flag, void* step_state,
                                int* live, void* stream);
/* amid_embed_bwd_f32 / amid_embed_bwd_rows_f32 (row_domain optional) carrying phase 5 of a sort plan as extra workgroups */
int amid_embed_bwd_sort_f32(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part, const void* rng_state,
                            int train, float p_drop, const long long* row_domain, const void* sort_plan, int sort_phase, void* stream);
int amid_embed_fwd_live_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                            int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop,
                            const int* live, void* stream);
/* amid_embed_fwd_live_f32 that also writes the step's compact index list over the live sequences + the items (B T + n_item_rows
 * entries): idx_c[i] = the id at walk position i, row_c[i] = that position's row in the full [2 B T + items] layout (where its
 * gradient will stand).  Sort (amid_sort_unique_rows_i32), segment reduce and row Adam of the step then run on half the entries:
 * the dead sequences' gradient rows are exact zeros (train_sr.py:205-211). */
int amid_embed_fwd_live_compact_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D,
                                    int n_item_rows, float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop,
                                    const int* live, int* idx_c, int* row_c, void* stream);
/* K1 with the lazy-Adam catch-up folded in (replaces amid_lazy_adam_catchup_positions_f32 + amid_embed_fwd*_f32 of a train step;
 * reference: the dense torch.optim.Adam of train_sr.py:480 moves every row every step, so a row idle since step `last` owes the
 * zero-gradient steps last + 1 .. t - 1 before model_seq.py:418-421 reads it): the owed steps are replayed IN REGISTERS for the value
 * written to xg; table / m / v / last are only read -- amid_optimizer_step_f32 of the same step replays them again (bit-identically) in
 * front of the real step.  sort_plan != NULL: phase 1 of the step's index sort rides as extra workgroups.  live, idx_c, row_c: NULL or
 * as amid_embed_fwd_live_f32 / amid_embed_fwd_live_compact_f32. */
int amid_embed_fwd_replay_f32(const float* table, const float* m_tab, const float* v_tab, const int* last, const int* idx_all,
                              const float* pos0, const float* pos1, int B, int T, int D, int n_item_rows, float* xg,
                              unsigned char* tmq, const void* rng_state, int train, float p_drop, const int* live, int* idx_c,
                              int* row_c, const void* adam_state, const void* sort_plan, int sort_phase, void* stream);
/* K1 (the gather: live / idx_c / row_c as amid_embed_fwd_live(_compact)_f32, NULL = every sequence / no compact list) with this step's bf16
 * fragment images of n_w (<= 24) square [D][D] weights written by extra workgroups of the same launch (what amid_sas_weights_bf16_planes
 * does as a launch of its own): w16_dst [n_w][planes][D][D] bf16, planes = 1 or 3; w16t_dst: NULL, or the same for the weights' TRANSPOSES
 * (the operands of the backward strips, amid_sas_strip_*_bwd_* with mma_bf16 = 3).  D = 128. */
int amid_embed_fwd_w16_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D, int n_item_rows,
                           float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop, const int* live, int* idx_c,
                           int* row_c, const float* const* w_src, int n_w, int w_planes, void* w16_dst, void* w16t_dst, void* stream);
/* 1 when the matrix-core attention kernels cover the shape (causal, T <= 64, H <= 8, head dim 16 or 8): the live-list entries below */
int amid_attn_live_supported(int T, int D, int H, int causal);
int amid_attn_fwd_live_f32(const float* q, const float* k, const float* v, int B, int T, int D, int H, int causal, int layer,
                           const void* step_state, int train, float p_drop, float* o, float* stats, const int* live, void* stream);
int amid_attn_bwd_live_f32(const float* q, const float* k, const float* v, const float* o, const float* stats, const float* d_o, int B,
                           int T, int D, int H, int causal, int layer, const void* step_state, int train, float p_drop, float* dq,
                           float* dk, float* dv, const int* live, void* stream);
int amid_head_fwd_bwd_own_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id, int B,
                              int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                              float* loss_part, float* dx, float* ditems, float* ln_part, float* sc_part, const float* const* tr_src,
                              float* const* tr_dst, int n_tr, void* stream);

/* ---- the same encoder-layer GEMM chains as register-resident STRIP kernels (csrc/strip_gemm.h, csrc/sasrec_strip.hip) -------------
 * replace: Log2feats.forward model_seq.py:371-383 (LayerNorms, nn.MultiheadAttention's in/out projections as called at :374,
 * PointWiseFeedForward :322-326) and its autograd (loss.backward(), train_sr.py:214), like the row-tile entry points above, fp32 only.
 * A wave keeps a [16, D] strip of activations in registers through a whole chain (accumulators of one GEMM are the operand of the
 * next); weights stream through LDS by DMA.  Activations are [2 * B * T, D] (domain 0's rows, then domain 1's), at most 2 GiB each.
 * live: NULL = every sequence; else int[B + 1] = the batch rows b with domain_id[b] == 0 (ascending), then those with
 * domain_id[b] != 0, then n0 = the number of the former (amid_live_list_i32): only the sequences (0, b) of the first group and
 * (1, b) of the second are read / written (the step's own loss masks the other domain of every sample, train_sr.py:205-211).
 * ln_part of the backward entries: [2 * ceil(B * T / amid_sas_strip_tile_rows())][2][D]; domain g owns the slots
 * [g * tpg, (g + 1) * tpg), unused ones are zeroed. */
int amid_sas_strip_tile_rows(void);
int amid_sas_strip_qkv_fwd_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* const* w_in,
                               const float* const* b_in, float ln_eps, int B, int T, int D, const int* live, float* qn, float* q, float* k,
                               float* v, void* stream);
/* nln_w != NULL: layer l + 1's amid_sas_strip_qkv_fwd_f32 (n* arguments) continues on the layer output in registers */
int amid_sas_strip_oproj_ffn_fwd_f32(const float* o, const float* qn, const float* const* w_o, const float* const* b_o,
                                     const float* const* ln_w, const float* const* ln_b, const float* const* w1, const float* const* b1,
                                     const float* const* w2, const float* const* b2, const unsigned char* tmq, float ln_eps, int B, int T,
                                     int D, const int* live, int layer, const void* step_state, int train, float p_drop, float* r, float* y,
                                     float* h, float* xo, const float* const* nln_w, const float* const* nln_b, const float* const* nw_in,
                                     const float* const* nb_in, float* nqn, float* nq, float* nk, float* nv, void* stream);
/* The backward strip entry points (and amid_sas_seq_bwd_f32) take mma_bf16: 1 = the data-gradient products on
 * v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, fp32 accumulation, everything else fp32; D = 128) -- the w*T pointer arguments then
 * hold the bf16 fragment images of the TRANSPOSED weights (amid_sas_weights_bf16(..., transposed = 1, ...), 2 D D bytes each) instead of
 * fp32 transposes; 0 = exact fp32 products. */
int amid_sas_strip_ffn_bwd_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                               const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int B, int T, int D,
                               const int* live, int layer, const void* step_state, int train, float p_drop, float* dpre2, float* dpre1,
                               float* dr, float* d_o, float* ln_part, int mma_bf16, void* stream);
/* fh != NULL: the layer below's amid_sas_strip_ffn_bwd_f32 (f* arguments) continues on d x in registers; dx is then not written */
int amid_sas_strip_qkv_bwd_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w,
                               const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int B, int T, int D,
                               const int* live, float* dx, float* ln_part, const unsigned char* tmq, const float* fh, const float* fr,
                               const float* const* fln_w, const float* const* fw1T, const float* const* fw2T, const float* const* fwoT,
                               int flayer, const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr,
                               float* fd_o, float* fln_part, int mma_bf16, void* stream);
/* the backward strip launches carrying a phase of a sort plan (amid_sort_plan_pack; ffn: 2, qkv with / without the fused
 * feed-forward backward: 3 / 4) as extra workgroups in front of the tiles' (the live tiles leave CUs free) */
int amid_sas_strip_ffn_bwd_sort_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                               const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int B, int T, int D,
                               const int* live, int layer, const void* step_state, int train, float p_drop, float* dpre2, float* dpre1,
                               float* dr, float* d_o, float* ln_part, const void* sort_plan, int sort_phase, int mma_bf16, void* stream);
int amid_sas_strip_qkv_bwd_sort_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w,
                               const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int B, int T, int D,
                               const int* live, float* dx, float* ln_part, const unsigned char* tmq, const float* fh, const float* fr,
                               const float* const* fln_w, const float* const* fw1T, const float* const* fw2T, const float* const* fwoT,
                               int flayer, const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr,
                               float* fd_o, float* fln_part, const void* sort_plan, int sort_phase, int mma_bf16, void* stream);

int amid_embed_bwd_rows_f32(float* dxg, const unsigned char* tmq, int B, int T, int D, int nsplit, float* dpos_part, const void* rng_state,
                            int train, float p_drop, const long long* row_domain, void* stream);   /* amid_embed_bwd_f32 behind the *_rows kernels: the dead sequences' rows are zero-filled, not read */

/* the 64-row (_rt4) build exists for the backward row-tile kernels only (the live-row backward of the headline shape: 50 rows per tile) */

/* amid_bert_wgrad_f32 over the live sequences only (see amid_sas_wgrad_rows_f32) */
int amid_bert_wgrad_rows_f32(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                             const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part, float* b_part,
                             const long long* row_domain, int B, int T, void* stream);


/* amid_bert_wgrad(_rows)_f32 with the products' mode: 0 = fp32 matrix instructions; 2 / 3 = every fp32 operand as three bf16 pieces whose
 * sum is the operand exactly, nine / six piece pairs on v_mfma_f32_16x16x32_bf16 (csrc/wgrad_split.h; fp32 accuracy, see amid_sas_wgrad_f32).
 * row_domain may be NULL (every row is walked) */
int amid_bert_wgrad_mode_f32(const float* const* dy, const float* const* x, const int* ldy, const int* ldx, const int* out_ld,
                             const int* out_group, const int* out_col, int n_ent, int M, int splits, float* w_part, float* b_part,
                             const long long* row_domain, int B, int T, int mode, void* stream);

/* the 64-row build of the BERT4Rec row-tile entry points (_rt4) and the *_rows entry points in every build */
int amid_bert_qkv_fwd_f32_rt4(const float* x, const float* const* ln_a, const float* const* ln_b, const float* const* w3x2,
                          const float* const* b3x2, int M, int rows_per_tile, float* y, float* q, float* k, float* v, void* stream);
int amid_bert_oproj_fwd_f32_rt4(const float* o, const float* x, const float* const* w, const float* const* b, int M, int rows_per_tile, int layer,
                            const void* step_state, int train, float p_drop, float* x1, void* stream);
int amid_bert_ffn1_fwd_f32_rt4(const float* x1, const float* const* ln_a, const float* const* ln_b, const float* const* w1, const float* const* b1,
                           int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop, float* y2, float* pre,
                           float* h, void* stream);
int amid_bert_ffn2_fwd_f32_rt4(const float* h, const float* x1, const float* const* w2, const float* const* b2, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* x2, void* stream);
int amid_bert_ffn2_bwd_f32_rt4(const float* dx2, const float* pre, const float* const* w2T, int M, int rows_per_tile, int layer,
                           const void* step_state, int train, float p_drop, float* dz, float* dpre, void* stream);
int amid_bert_ffn1_bwd_f32_rt4(const float* dpre, const float* dx2, const float* x1, const float* const* ln_a, const float* const* w1T,
                           const float* const* woT, int M, int rows_per_tile, int layer, const void* step_state, int train, float p_drop,
                           float* dx1, float* dt, float* d_o, float* ln_part, void* stream);
int amid_bert_qkv_bwd_f32_rt4(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x, const float* const* ln_a,
                          const float* const* wT3x2, int M, int rows_per_tile, float* dx, float* ln_part, void* stream);

/* ---- round 5: the live-sequence train step on an input pool in twelve launches (was fifteen) ------------------------------------------
 * replaces, inside the loop body of train() (train_sr.py:190-217): the batch marshal (:191-200) + the catch-up of the lazy dense-equivalent
 * Adam (torch.optim.Adam over the whole table, :480) as ONE launch; the embedding layer's backward (autograd of model_seq.py:361-366) on
 * the last strip launch; the second phase of the embedding-gradient reduction (autograd of :27-29) inside the optimizer launch.
 *
 * amid_step_head_f32: the batch (step_done + phase) % n_pool of the pool (rows = SasrecEngine.pack_batch images) mirrored into in_pack, the
 * full index list idx_all [2 B T + B (1 + n_neg)], the live list (amid_live_list_i32), the COMPACT list idx_c / row_c [B T + B (1 + n_neg)]
 * (every sample's own-domain sequence then the items; row_c = the entry's row in the full layout), the lazy-Adam catch-up of the compact
 * list's rows, phase 1 of sort_plan (a plan packed on (idx_c, row_c); NULL: no rider) and the step counter's bump.  The caller's next
 * launch on the stream must be one of amid_embed_fwd*_f32 (it re-joins the step state's two counters).
 * replaces: amid_pack_indices_pool_live + amid_lazy_adam_catchup_positions_sort_f32. */
int amid_step_head_f32(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack, int in_words, int B, int T,
                       int n_neg, long long n_rows, int* idx_all, int* idx_c, int* row_c, int* live, int* err_flag, float* table, float* m,
                       float* v, int* last, int D, void* step_state, const void* sort_plan, void* stream);
/* amid_sas_strip_qkv_bwd(_sort)_f32 without the fused feed-forward backward, with the embedding layer's element-wise backward applied to dx
 * before it is stored: the input dropout's keep bits (site SITE_EMB, p = emb_p_drop) redrawn from step_state and the "== 0" bits of
 * emb_tmq (model_seq.py:361-366) -- what amid_embed_bwd_f32 does in a pass of its own.  sort_plan NULL or a plan whose phase 4 rides. */
int amid_sas_strip_qkv_bwd_emb_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x, const float* const* ln_w,
                                   const float* const* wqT, const float* const* wkT, const float* const* wvT, float ln_eps, int B, int T, int D,
                                   const int* live, float* dx, float* ln_part, const unsigned char* emb_tmq, const void* step_state, int train,
                                   float emb_p_drop, const void* sort_plan, int sort_phase, int mma_bf16, void* stream);
/* amid_sas_strip_ffn_bwd_f32 (mma_bf16 = 3) as the N-split build on producer-side pieces (csrc/sasrec_strip_px.hip, round 6): eight waves per
 * 64-row tile -- four strips x two column parts --, the chain's operands cross the parts through LDS as bf16 pieces, the three-plane images of
 * the TRANSPOSED weights stream through three plane slots (the forward's machinery, amid_sas_seq_fwd_split_f32).  w1T / w2T / woT: per domain,
 * the images ([3][D][D] bf16).  ln_stat: the forward's row statistics of the layer ([2 B T][4], LN2's mean and rstd at +2) or NULL.
 * sort_plan: optional rider (phase 2).  D = 128, p_drop = 0.5 or eval.  Agrees with the strip build to rounding (the LayerNorm backward's row
 * sums are added part by part). */
int amid_sas_strip_ffn_bwd_px_f32(const float* dxo, const unsigned char* tmq, const float* h, const float* r, const float* const* ln_w,
                                  const float* const* w1T, const float* const* w2T, const float* const* woT, float ln_eps, int B, int T, int D,
                                  const int* live, int layer, const void* step_state, int train, float p_drop, float* dpre2, float* dpre1,
                                  float* dr, float* d_o, float* ln_part, const float* ln_stat, const void* sort_plan, int sort_phase,
                                  void* stream);
/* amid_sas_wgrad_rows_f32 (mma_bf16 = 3, D = 128) carrying the LAST phase (5: run heads) of a sort plan as extra workgroups.
 * (amid_sas_wgrad_rows_sort_ln_f32 also takes mma_bf16 = 4: the same launch staging and multiplying ONE bf16 piece per operand -- bf16
 * products with fp32 accumulation, the folded bf16 step's weight gradients; round 6) */
int amid_sas_wgrad_rows_sort_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits, float* const* w_part,
                                 float* const* b_part, const long long* row_domain, int B, int T, int mma_bf16, const void* sort_plan,
                                 void* stream);
/* The one-launch forward on bf16 pieces saving SEVEN tensors per layer instead of nine: qn = LN1(x) and y = LN2(r) (model_seq.py:373, :381)
 * are not stored, ln_stat[l] [2 B T][4] receives every row's (mean, rstd) of the two LayerNorms instead.  The backward strips rebuild the
 * normalised rows from x / r anyway; amid_sas_wgrad_rows_sort_ln_f32 (x[6 l + 0] = the layer's x, x[6 l + 4] = its r; ln1_* / ln2_*: 2 n_layers
 * pointers [layer][domain]) rebuilds the two operands as (row - mean) rstd gamma + beta while it stages them.  D = 128. */
int amid_sas_seq_fwd_split_lnstat_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                      const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                      const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                      const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                      float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h,
                                      const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                      const void* step_state, int train, float p_drop, const void* w16x3, void* stream);
/* amid_sas_seq_fwd_split_lnstat_f32 + amid_head_fwd_bwd_own_vec_f32 in ONE launch: a live sequence is a sample, so the workgroup that encoded
 * the sequence of sample b finishes with that sample's head -- LN_last + mean over T (model_seq.py:385, :432-434), predictModule.forward
 * (model_seq.py:40-54), the masked BCE term and dLoss/dp (train_sr.py:203-212), the scorer's backward, LN_last' -- on the rows it still holds.
 * The last layer's output is stored only if xout != NULL (nothing of the step reads it).  Arguments: the forward's, then the head's (without x and the transposes; last_ln_w /
 * last_ln_b: 2 pointers, one per domain).  Same additions in the same order as the two launches.  T 33 ... 64, D 128, live != NULL, hid <= 32,
 * NI <= 64; AMID_ERR_UNSUPPORTED (nothing enqueued) otherwise. */
int amid_sas_seq_fwd_split_lnstat_head_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                           const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                           const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                           const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q,
                                           float* const* k, float* const* v, float* const* o, float* const* stats, float* const* r,
                                           float* const* h, const unsigned char* tmq, float ln_eps, int B, int T, int D, int H, const int* live,
                                           const void* step_state, int train, float p_drop, const void* w16x3,
                                           const float* const* last_ln_w, const float* const* last_ln_b, const float* items, const float* sw1,
                                           const float* sb1, const float* sw2, const float* sb2, const float* labels, const long long* domain_id,
                                           int NI, int hid, float* u, float* p1, float* p2, float* dp1, float* dp2, float* loss_part, float* dx,
                                           float* ditems, float* ln_part, float* hidg, void* stream);
int amid_sas_wgrad_rows_sort_ln_f32(const float* const* dy, const float* const* x, int n_layers, int M, int D, int splits, float* const* w_part,
                                    float* const* b_part, const long long* row_domain, int B, int T, int mma_bf16, const void* sort_plan,
                                    const float* const* ln_stat, const float* const* ln1_w, const float* const* ln1_b,
                                    const float* const* ln2_w, const float* const* ln2_b, void* stream);
/* amid_grad_tail_f32's first launch over a compact sorted list (pos_sorted = rows of grad_rows) + the position rows' gradients
 * dpos0 / dpos1 [T, D] = the sum over the live sequences of a domain (live: amid_live_list_i32) of their rows of grad_rows, in list order.
 * Phase B of the segment reduce is left to amid_optimizer_step_spans_f32. */
int amid_grad_tail_live_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                            void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off, int total_blocks,
                            const int* live, int B, int T, float* dpos0, float* dpos1, const float* hidg, const float* u,
                            const float* items, int NI, int hid, float* dW1, float* db1, float* dW2, float* db2, void* stream);
/* amid_grad_tail_live_f32 (hidg = NULL) for a caller that SHIPS the row gradients -- the data-parallel step (no reference analogue:
 * train_sr.py:473 has DataParallel commented out): phase B of the segment reduce follows as a second launch instead of
 * riding in this rank's optimizer.  n_out = 0: that launch alone (uniq_grad complete).  n_out > 0: amid_grad_tail_pack_f32's second launch
 * (uniq_grad points into the exchange chunk, out_ids <- ids padded with pad_id to n_out, dense_dst <- dense_src; err_flag gets
 * AMID_FLAG_UMAX_EXCEEDED when *n_uniq > n_out).  The same additions in the same order as the single-GPU folded step's. */
int amid_grad_tail_live_dp_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                               void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off, int total_blocks,
                               const int* live, int B, int T, float* dpos0, float* dpos1, const int* uniq_ids, const int* n_uniq, int n_out,
                               int pad_id, int* out_ids, const float* dense_src, float* dense_dst, long long dense_n, int* err_flag,
                               void* stream);
/* hidg (optional; with it u [2, B, D], items [B, NI, D] and the four gradient outputs): the scorer's weight gradients (autograd of
 * predictModule.forward, model_seq.py:40-54) summed over the batch from the per-sample hidden gradients of amid_head_fwd_bwd_own_vec_f32
 * -- hidg [B][amid_scorer_vec_floats(NI, hid)] = da [2][hid] | dc [NI][hid] | dW2's [hid] | db2's: dW1 = sum_b da_b (x) u_b + dc_b (x) items_b.
 * amid_head_fwd_bwd_own_vec_f32 = amid_head_fwd_bwd_own_f32 writing hidg instead of the 32 KB-per-sample partials sc_part. */
long long amid_scorer_vec_floats(int NI, int hid);
/* amid_sas_strip_qkv_bwd_sort_f32 with the fused feed-forward backward (phase 3 of the sort plan) that ALSO carries the scorer sums above as
 * extra workgroups (the launch's live tiles leave CUs idle at the headline shape, and the sums depend on the head launch only): then
 * amid_grad_tail_live_f32 is called with hidg = NULL.  D = 128, mma_bf16 = 3. */
int amid_sas_strip_qkv_bwd_sort_scorer_f32(const float* dq, const float* dk, const float* dv, const float* dr, const float* x,
                                           const float* const* ln_w, const float* const* wqT, const float* const* wkT,
                                           const float* const* wvT, float ln_eps, int B, int T, int D, const int* live, float* ln_part,
                                           const unsigned char* tmq, const float* fh, const float* fr, const float* const* fln_w,
                                           const float* const* fw1T, const float* const* fw2T, const float* const* fwoT, int flayer,
                                           const void* step_state, int train, float p_drop, float* fdpre2, float* fdpre1, float* fdr,
                                           float* fd_o, float* fln_part, const void* sort_plan, int sort_phase, int mma_bf16,
                                           const float* hidg, const float* u, const float* items, int NI, int hid, float* dW1, float* db1,
                                           float* dW2, float* db2, void* stream);
int amid_head_fwd_bwd_own_vec_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* items, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id, int B,
                                  int T, int NI, int D, int hid, float eps, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                  float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg, const float* const* tr_src,
                                  float* const* tr_dst, int n_tr, void* stream);
/* amid_grad_tail_f32 without its second launch (phase B of the segment reduce): for a caller whose NEXT launch is
 * amid_optimizer_step_spans_f32 on the same (seg_off, seg_of, n_idx, workspace) */
int amid_grad_tail_nospans_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                               void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, int max_count, const int* blk_off,
                               int total_blocks, void* stream);
/* amid_optimizer_step_f32 behind amid_grad_tail_live_f32 / amid_grad_tail_nospans_f32: the runs of the sorted list that cross 64-entry chunks are summed from the tail's
 * partial rows (workspace) by extra workgroups -- the additions of amid_embgrad_segreduce_f32's second launch in the same order --, written
 * to uniq_grad and applied on the spot.  D = 64 / 128 / 256. */
int amid_optimizer_step_spans_f32(float* p, float* m, float* v, const float* g, long long n, float* table, float* m_tab, float* v_tab, int* last,
                                  const int* uniq_ids, const int* n_uniq, int n_uniq_max, float* uniq_grad, int D, float grad_scale,
                                  const void* step_state, const int* seg_off, const int* seg_of, int n_sorted, const void* workspace,
                                  void* stream);

/* ---- the gather K1 as the prologue of the forward's workgroups (round 6) -----------------------------------------------------------------------
 * replaces: amid_embed_fwd_w16_f32 (embItemLayerEnhance.forward model_seq.py:27-29 + Log2feats' position add / dropout / == 0 mask :361-366) as a
 * launch of its own in the folded step and in the evaluation batch.  The workgroup that encodes a live sequence builds layer 0's input rows
 * from table[idx_all[row]] + pos (K1's arithmetic: the same bits), stores them to x_in[0] and the mask bytes to tmq for the backward (the
 * inference forward stores neither), gathers the sample's NI item rows into `items` [B, NI, D] and the launch re-joins StepState::step_done.
 * idx_all: the step's full index list [seq_d1 B T | seq_d2 B T | items B NI]; the weight images w16x3 must be current (amid_step_head_w16_f32
 * writes them with extra workgroups; evaluation: amid_sas_weights_bf16_planes once).  D = 128, 32 < T <= 64 with the head, T <= 64 without,
 * p_drop = 0.5 or eval; otherwise AMID_ERR_UNSUPPORTED and nothing is enqueued.  The other arguments: as amid_sas_seq_fwd_split_lnstat(_head)_f32 /
 * amid_sas_seq_fwd_split_infer_f32. */
int amid_sas_seq_fwd_gather_head_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                     const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                     const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                     const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                     float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h, unsigned char* tmq,
                                     float ln_eps, int B, int T, int D, int H, const int* live, void* step_state, int train, float p_drop,
                                     const void* w16x3, const float* const* last_ln_w, const float* const* last_ln_b, float* items,
                                     const float* sw1, const float* sb1, const float* sw2, const float* sb2, const float* labels,
                                     const long long* domain_id, int NI, int hid, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                     float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg, const float* table,
                                     const int* idx_all, const float* pos0, const float* pos1, void* stream);
int amid_sas_seq_fwd_gather_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h, unsigned char* tmq,
                                float ln_eps, int B, int T, int D, int H, const int* live, void* step_state, int train, float p_drop,
                                const void* w16x3, float* items, int NI, const float* table, const int* idx_all, const float* pos0,
                                const float* pos1, void* stream);
/* ... with the twelve projection products on ONE bf16 piece per operand (compute = "bf16" on the folded step, round 6): the same launches on
 * the same three-plane images, reading only the hi planes and multiplying only the operands' hi pieces -- bf16 products, fp32 accumulation. */
int amid_sas_seq_fwd_gather_head_p1_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                     const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                     const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                     const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                     float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h, unsigned char* tmq,
                                     float ln_eps, int B, int T, int D, int H, const int* live, void* step_state, int train, float p_drop,
                                     const void* w16x3, const float* const* last_ln_w, const float* const* last_ln_b, float* items,
                                     const float* sw1, const float* sb1, const float* sw2, const float* sb2, const float* labels,
                                     const long long* domain_id, int NI, int hid, float* u, float* p1, float* p2, float* dp1, float* dp2,
                                     float* loss_part, float* dx, float* ditems, float* ln_part, float* hidg, const float* table,
                                     const int* idx_all, const float* pos0, const float* pos1, void* stream);
int amid_sas_seq_fwd_gather_p1_f32(int n_layers, const float* const* x_in, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                const float* const* w2, const float* const* b2, float* const* ln_stat, float* const* q, float* const* k,
                                float* const* v, float* const* o, float* const* stats, float* const* r, float* const* h, unsigned char* tmq,
                                float ln_eps, int B, int T, int D, int H, const int* live, void* step_state, int train, float p_drop,
                                const void* w16x3, float* items, int NI, const float* table, const int* idx_all, const float* pos0,
                                const float* pos1, void* stream);

int amid_sas_seq_fwd_gather_infer_f32(int n_layers, float* xout, const float* const* ln1_w, const float* const* ln1_b, const float* const* w_in,
                                      const float* const* b_in, const float* const* w_o, const float* const* b_o, const float* const* ln2_w,
                                      const float* const* ln2_b, const float* const* w1, const float* const* b1, const float* const* w2,
                                      const float* const* b2, float ln_eps, int B, int T, int D, int H, const int* live, const void* w16x3,
                                      const float* table, const int* idx_all, const float* pos0, const float* pos1, void* stream);
/* amid_step_head_f32 + the step's weight images by extra workgroups (amid_embed_fwd_w16_f32's riders): w_src = n_w square [D][D] fp32 weights,
 * w16_dst [n_w][w_planes][D][D] bf16, w16t_dst (optional) the images of their transposes.  D = 128, n_w (x 2 with transposes) <= 96. */
int amid_step_head_w16_f32(const long long* pool, long long pool_stride, int n_pool, long long phase, long long* in_pack, int in_words, int B, int T,
                           int n_neg, long long n_rows, int* idx_all, int* idx_c, int* row_c, int* live, int* err_flag, float* table, float* m,
                           float* v, int* last, int D, void* step_state, const void* sort_plan, const float* const* w_src, int n_w, int w_planes,
                           void* w16_dst, void* w16t_dst, void* stream);

/* amid_grad_tail_live_f32 (hidg = NULL) and amid_optimizer_step_spans_f32 as ONE launch (round 6): every producer of a gradient slice applies
 * Adam to it on the spot -- the partial sums and the position rows their dense slices, a half-wave per unique row the rows whose runs lie
 * inside a 64-entry chunk, and the chunk block that takes the LAST ticket the runs that cross chunks (their pieces travel with agent scope;
 * nobody waits).  Replaces: optimizer.zero_grad(); loss.backward(); optimizer.step() 's tail, train_sr.py:213-215.  The same additions in the
 * same order and the same Adam arithmetic as the two launches: the same bits.  ticket: one int32, zero before the first call (left zero).
 * [left_lo, left_hi): floats of the flat dense buffer whose gradients are final before the launch (the scorer's, summed by the strip riders). */
int amid_grad_tail_opt_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                           void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off, int total_blocks,
                           const int* live, int B, int T, float* dpos0, float* dpos1, float* p, float* m, float* v, float* g, long long n,
                           long long left_lo, long long left_hi, float* table, float* m_tab, float* v_tab, int* last, const int* uniq_ids,
                           const int* n_uniq, int n_uniq_max, float grad_scale, const void* step_state, int* ticket, void* stream);

/* amid_grad_tail_live_dp_f32 as ONE launch (amid_grad_tail_opt_f32's workgroups, shipping instead of applying; no reference analogue:
 * train_sr.py:473): uniq_grad complete at the end of the launch; n_out > 0: out_ids [n_out] <- the unique ids padded with pad_id, dense_dst
 * (optional, 16-byte aligned, laid out like g) <- every dense gradient the launch finishes + the floats [left_lo, left_hi) of g; err_flag gets
 * AMID_FLAG_UMAX_EXCEEDED when *n_uniq > n_out.  ticket: one int32, zero before the first call (left zero).  The same bits as the two launches. */
int amid_grad_tail_live_dp1_f32(const float* grad_rows, const int* pos_sorted, const int* seg_off, const int* seg_of, int n_idx, int D,
                                void* workspace, float* uniq_grad, const void* entries_dev, int n_entries, const int* blk_off, int total_blocks,
                                const int* live, int B, int T, float* dpos0, float* dpos1, float* g, long long n, long long left_lo,
                                long long left_hi, const int* uniq_ids, const int* n_uniq, int n_uniq_max, int n_out, int pad_id, int* out_ids,
                                float* dense_dst, int* err_flag, int* ticket, void* stream);

/* ---- evaluation (round 6): test(), train_sr.py:31-128 ------------------------------------------------------------------------------------------
 * replaces, for the plain SASRec model: model(u, i, neg, seq_d1, seq_d2, ..., False) under no_grad (train_sr.py:55-56) + the masked BCE
 * (:63-64) + choose_predict / get_sample_scores' rank of column 0 (utils.py:21-40, :296-297; fix_value: train_sr.py:42, :114-115).
 * test() reads of every sample only its OWN domain's logits and the loss masks the other domain's, so the forward below encodes the live
 * sequences only (live: amid_live_list_i32) and the head forms only those NI logits.
 * amid_sas_seq_fwd_split_infer_f32 = amid_sas_seq_fwd_split_f32 in eval mode that stores nothing but xout (bit-identical rows; x0 = layer 0's
 * input rows [2 B T, D]); D = 128, T <= 64.
 * amid_eval_head_f32: a workgroup per sample b -- u = mean_t LN_last(x[own, b]) (ln_w / ln_b: host arrays of 2 device pointers, or both NULL),
 * p[n] = predictModule(u, table[ids[b][n]]) for the NI candidates (ids [B, NI] int32, column 0 the positive: the item part of the step's index
 * list, range-checked by amid_pack_indices*; the rows are gathered from the table inside the launch), and from them rank[b] = #{n >= 1 :
 * p[n] > p[0] - fix_value}, rank_raw[b] (fix_value = 0), loss_part[b] = sum_n BCE(p[n], labels[b][n]) / (B NI) (labels optional, with
 * loss_part); u [B, D], p [B, NI], rank, rank_raw optional outputs.  Same operations in the same order as amid_head_fwd_f32 +
 * amid_positive_rank_f32 on a forward over both domains: the same bits.  D <= 128 (% 32), hid <= 64 (% 4). */
int amid_sas_seq_fwd_split_infer_f32(int n_layers, const float* x0, float* xout, const float* const* ln1_w, const float* const* ln1_b,
                                     const float* const* w_in, const float* const* b_in, const float* const* w_o, const float* const* b_o,
                                     const float* const* ln2_w, const float* const* ln2_b, const float* const* w1, const float* const* b1,
                                     const float* const* w2, const float* const* b2, const unsigned char* tmq, float ln_eps, int B, int T,
                                     int D, int H, const int* live, const void* w16x3, void* stream);
int amid_eval_head_f32(const float* x, const float* const* ln_w, const float* const* ln_b, const float* table, const int* ids,
                       const float* w1, const float* b1, const float* w2, const float* b2, const float* labels, const long long* domain_id,
                       int B, int T, int NI, int D, int hid, float eps, float fix_value, float* u, float* p, int* rank, int* rank_raw,
                       float* loss_part, void* stream);

/* ---- BERT4Rec strips on bf16 pieces (round 5; csrc/bert_strip.hip MODE 3) ------------------------------------------------------------------
 * The strip launches of a TransformerBlock (model_seq.py:242-245 and its autograd) with every product as six bf16 piece pairs at fp32
 * accuracy -- what SASRec's strips run on since round 4.  Each 128 x 128 weight TILE a chain multiplies with is a three-plane fragment image
 * (hi + mid + lo = the fp32 element exactly) written once per step by amid_bert_weight_images_f32: tile i = src[i][r * ld[i] + c] (tr[i] = 0)
 * or its transpose (tr[i] != 0), r, c < 128, n <= 96 tiles, dst16 [n][3][128][128] bf16.  The *_p3 entry points take the argument lists of
 * their fp32 twins with every weight pointer replaced by a pointer to tile images (as float*): w3 / wo / wT3 / woT -- one image each; w1, w2,
 * w1T, w2T -- the FIRST of the matrix's four tiles' images, the others 3 * 128 * 128 bf16 further each (w_1 [512][128] and w_2^T: row blocks;
 * w_2 [128][512] and w_1^T: column blocks). */
int amid_bert_weight_images_f32(const float* const* src, const int* ld, const int* tr, int n, void* dst16, void* stream);
/* BERT4Rec(isInC / isItC) under data parallel (round 5): amid_bert_comp_fwd_f32 / _bwd_f32 (model_seq.py:283-294 and its autograd) on a
 * shard of the global batch, cut at the all-reduces of the token sums S (forward) and of their gradient dZ (backward) exactly as
 * amid_inc_embed_fwd_shard_f32 / amid_inc_bwd_shard_f32: B rows = samples j0 .. j0 + B - 1 of Bg, s_all [2, Bg] the all-gathered scores. */
int amid_bert_comp_fwd_shard_f32(const float* xg, const float* s_all, const float* const* w_nn, const float* const* b_nn,
                                 const float* const* w_bs, const float* const* b_bs, float threshold, int cross, int B, int T, int D, int Bg,
                                 int j0, int phase, float* gate, float* S, float* Z, float* sw, float* x0, void* stream);
int amid_bert_comp_bwd_shard_f32(const float* xg, const float* dx0, const float* gate, const float* S, const float* sw,
                                 const float* const* w_nn, const float* const* b_nn, const float* const* w_bs, int cross, int B, int T, int D,
                                 int Bg, int j0, int phase, float gscale, float* dZ, float* dS, float* rows, float* const* dw_nn,
                                 float* const* db_nn, float* const* dw_bs, float* const* db_bs, float* dxg, void* stream);
/* K1 (amid_embed_fwd_live_f32 / amid_embed_fwd_f32 by `live`) carrying amid_bert_weight_images_f32's tiles as extra workgroups of the gather */
int amid_embed_fwd_tiles_f32(const float* table, const int* idx_all, const float* pos0, const float* pos1, int B, int T, int D, int n_item_rows,
                             float* xg, unsigned char* tmq, const void* rng_state, int train, float p_drop, const int* live,
                             const float* const* w_src, const int* w_ld, const int* w_tr, int n_w, int w_planes, void* w16_dst, void* stream);
int amid_bert_strip_qkv_fwd_pro_p3_f32(const float* x, const float* const* la, const float* const* lb, const float* const* w3_img,
                                       const float* const* b3, int B, int T, const int* live, float* y, float* q, float* k, float* v,
                                       const long long* seq_d2, int n_keys, unsigned char* key_keep, const float* const* tr_src,
                                       float* const* tr_dst, const int* tr_rows, const int* tr_cols, int n_tr, void* stream);
int amid_bert_strip_oproj_ffn_fwd_p3_f32(const float* o, const float* x, const float* const* wo_img, const float* const* bo,
                                         const float* const* la, const float* const* lb, const float* const* w1_img, const float* const* b1,
                                         const float* const* w2_img, const float* const* b2, int B, int T, const int* live, int layer,
                                         const void* step_state, int train, float p_drop, float* x1, float* y2, float* pre, float* h,
                                         float* x2, const float* const* nla, const float* const* nlb, const float* const* nw3_img,
                                         const float* const* nb3, float* ny, float* nq, float* nk, float* nv, void* stream);
int amid_bert_strip_ffn_bwd_p3_f32(const float* dx2, const float* pre, const float* x1, const float* const* la, const float* const* w2T_img,
                                   const float* const* w1T_img, const float* const* woT_img, int B, int T, const int* live, int layer,
                                   const void* step_state, int train, float p_drop, float* dz, float* dpre, float* dx1, float* dt,
                                   float* d_o, float* ln_part, void* stream);
int amid_bert_strip_qkv_bwd_p3_f32(const float* dq, const float* dk, const float* dv, const float* dx1, const float* x,
                                   const float* const* la, const float* const* wT3_img, int B, int T, const int* live, float* dx,
                                   int zero_dead, float* ln_part, const float* fpre, const float* fx1, const float* const* fla,
                                   const float* const* fw2T_img, const float* const* fw1T_img, const float* const* fwoT_img, int flayer,
                                   const void* step_state, int train, float p_drop, float* fdz, float* fdpre, float* fdx1, float* fdt,
                                   float* fd_o, float* fln_part, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMID_HIP_H */
