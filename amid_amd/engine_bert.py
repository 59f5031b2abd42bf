"""Host-side driver of the BERT4Rec variant of the hot path (reference: BERT4Rec model_seq.py:248-309; training loop
train_sr.py:190-217).  Same engine as SASRec (table, lazy Adam, index sort, segment reduce, fused head, hipGraph);
only the encoder launches, the saved activations and the dense parameter list differ.

Reference quirks kept: hidden size 128 / 4 heads / FFN 512 / dropout 0.1 are hard-coded whatever ``--emb_dim`` says
(model_seq.py:264-267), so emb_dim must be 128; there is NO positional embedding and no final LayerNorm; ONE key mask,
taken from ``seq_d2 > 0``, is applied to BOTH encoders (model_seq.py:288, :295-298)."""
from __future__ import annotations

import ctypes
import os
from typing import List, Tuple

import torch

from ._lib import lib, ptr_array
from .engine import SasrecEngine, SasrecPlan

BERT_HEADS, BERT_FF, BERT_P_DROP, BERT_HIDDEN = 4, 512, 0.1, 128


def bert4rec_dense_names(hid: int, dr: bool = False, comp: str = "", comp_bs: int = 0) -> List[Tuple[str, Tuple[int, ...]]]:
    """Non-table parameters in the reference's state_dict order; dr: with the predict_ips / predict_gfunc heads of isDR=True
    (model_seq.py:268-271); comp = "inc" / "itc": with the InnerComp / InterComp modules of isInC / isItC at bs = comp_bs
    (:257-263)."""
    D, F = BERT_HIDDEN, BERT_FF
    out: List[Tuple[str, Tuple[int, ...]]] = []
    if comp:
        for d in (1, 2):
            out.append((f"{comp}_d{d}.trans_nn.weight", (D, D)))
            out.append((f"{comp}_d{d}.trans_nn.bias", (D,)))
            out.append((f"{comp}_d{d}.trans_bs.weight", (1, comp_bs)))
            out.append((f"{comp}_d{d}.trans_bs.bias", (1,)))
    for d in (1, 2):
        for l in (0, 1):
            pre = f"transform{d}.{l}"
            for j in range(3):
                out.append((f"{pre}.attention.linear_layers.{j}.weight", (D, D)))
                out.append((f"{pre}.attention.linear_layers.{j}.bias", (D,)))
            out.append((f"{pre}.attention.output_linear.weight", (D, D)))
            out.append((f"{pre}.attention.output_linear.bias", (D,)))
            out.append((f"{pre}.feed_forward.w_1.weight", (F, D)))
            out.append((f"{pre}.feed_forward.w_1.bias", (F,)))
            out.append((f"{pre}.feed_forward.w_2.weight", (D, F)))
            out.append((f"{pre}.feed_forward.w_2.bias", (D,)))
            out.append((f"{pre}.input_sublayer.norm.a_2", (D,)))
            out.append((f"{pre}.input_sublayer.norm.b_2", (D,)))
            out.append((f"{pre}.output_sublayer.norm.a_2", (D,)))
            out.append((f"{pre}.output_sublayer.norm.b_2", (D,)))
    for head in ("predictModule",) + (("predict_ips", "predict_gfunc") if dr else ()):
        out.append((f"{head}.fc.0.weight", (hid, 2 * D)))
        out.append((f"{head}.fc.0.bias", (hid,)))
        out.append((f"{head}.fc.2.weight", (1, hid)))
        out.append((f"{head}.fc.2.bias", (1,)))
    return out


N_ENT = 12      # weight-gradient tiles of 128 x 128 per block: q, k, v, o, 4 x w_1, 4 x w_2


class BertPlan(SasrecPlan):
    LIVE_ROWS_BWD = False

    def _alloc_model_fwd(self, eng, f) -> None:
        M, D, F = self.shape.M, eng.D, BERT_FF
        # the block's GEMM chains as register-resident strip kernels (csrc/bert_strip.hip) wherever the [2M, 512] tensors fit a buffer
        # descriptor; the row-tile kernels of csrc/bert.hip beyond
        self.strip = bool(self.strip and lib().value("amid_bert_strip_supported", self.shape.B, self.shape.Tenc, D))
        self.key_keep = torch.zeros(self.shape.B, self.shape.Tenc, dtype=torch.uint8, device=eng.device)
        self.y = [f(2 * M, D) for _ in range(2)]        # LNb_in(x)
        self.x1 = [f(2 * M, D) for _ in range(2)]
        self.y2 = [f(2 * M, D) for _ in range(2)]       # LNb_out(x1)
        self.pre = [f(2 * M, F) for _ in range(2)]
        self.h = [f(2 * M, F) for _ in range(2)]

    def _alloc_model_bwd(self, eng, f) -> None:
        M, D, F = self.shape.M, eng.D, BERT_FF
        self.rpt_b, self.rt_suffix_b = self.rpt, self.rt_suffix
        self.tpg_b = self.stpg if self.strip else (M + self.rpt_b - 1) // self.rpt_b          # strip kernels: 64-row tiles
        self.ln1_part = [f(2 * self.tpg_b, 2, D) for _ in range(2)]       # one slot per backward tile
        self.ln2_part = [f(2 * self.tpg_b, 2, D) for _ in range(2)]
        self.dz, self.dt, self.dx1 = f(2 * M, D), f(2 * M, D), f(2 * M, D)
        self.dpre = f(2 * M, F)
        # 2 domains x 12 tiles x splits workgroups: 21 splits = 504, two per CU, for the bf16-piece products (72 KB of LDS each); 10 = 240,
        # one per CU, for the fp32 matrix instructions (AMID_WGRAD_SPLIT=0).  Measured step 0.6239 (21) / 0.6252 (10) / 0.6398 (14) ms
        default_splits = 21 if eng.WGRAD_SPLIT in ("6", "9") else 10
        self.splits = max(1, min(int(self.WGRAD_SPLITS or default_splits), M // 128))
        self.w_part = [f(2, N_ENT, self.splits, D * D) for _ in range(2)]
        self.b_part = [f(2, N_ENT, self.splits, D) for _ in range(2)]

    def _model_reduce_entries(self, eng, add) -> None:
        D, F = eng.D, BERT_FF
        fp, G = eng.dense, eng.dense.grad
        S = self.splits
        for l in (0, 1):
            for g in (0, 1):
                pre = f"transform{g + 1}.{l}"
                wbase = lambda e: ((g * N_ENT + e) * S) * D * D      # noqa: E731
                bbase = lambda e: ((g * N_ENT + e) * S) * D          # noqa: E731
                for j in range(3):
                    add(self.w_part[l], wbase(j), fp.ptr(f"{pre}.attention.linear_layers.{j}.weight", G), D * D, S, D * D)
                    add(self.b_part[l], bbase(j), fp.ptr(f"{pre}.attention.linear_layers.{j}.bias", G), D, S, D)
                add(self.w_part[l], wbase(3), fp.ptr(f"{pre}.attention.output_linear.weight", G), D * D, S, D * D)
                add(self.b_part[l], bbase(3), fp.ptr(f"{pre}.attention.output_linear.bias", G), D, S, D)
                for c in range(4):
                    # w_1 [512,128]: rows c*128 .. are one contiguous 128 x 128 tile
                    add(self.w_part[l], wbase(4 + c), fp.ptr(f"{pre}.feed_forward.w_1.weight", G, c * D * D), D * D, S, D * D)
                    add(self.b_part[l], bbase(4 + c), fp.ptr(f"{pre}.feed_forward.w_1.bias", G, c * D), D, S, D)
                # w_2 [128,512]: its four column tiles share output group 8 -> partials are [S][128*512] from entry 8 on
                add(self.w_part[l], wbase(8), fp.ptr(f"{pre}.feed_forward.w_2.weight", G), D * F, S, D * F)
                add(self.b_part[l], bbase(8), fp.ptr(f"{pre}.feed_forward.w_2.bias", G), D, S, D)
                tb = g * self.tpg_b * 2 * D
                add(self.ln1_part[l], tb, fp.ptr(f"{pre}.input_sublayer.norm.a_2", G), 2 * D, self.tpg_b, D)
                add(self.ln1_part[l], tb + D, fp.ptr(f"{pre}.input_sublayer.norm.b_2", G), 2 * D, self.tpg_b, D)
                add(self.ln2_part[l], tb, fp.ptr(f"{pre}.output_sublayer.norm.a_2", G), 2 * D, self.tpg_b, D)
                add(self.ln2_part[l], tb + D, fp.ptr(f"{pre}.output_sublayer.norm.b_2", G), 2 * D, self.tpg_b, D)


class Bert4recEngine(SasrecEngine):
    HEADS = BERT_HEADS
    PLAN_CLS = BertPlan
    EMB_DIMS = (BERT_HIDDEN,)
    SHORT_TILE_BUILDS = True
    STRIP_KERNELS = True         # the block's GEMM chains on csrc/bert_strip.hip (BertPlan.strip); bert.hip's row-tile kernels beyond 2 GiB
    SORT_RIDERS = False          # (the riders' host launches are the SASRec strip backward's)
    FUSED_TAIL = False           # (the one-launch step head and the folded tail are the SASRec step's)
    EVAL_FUSED = False           # (the four-launch evaluation batch is the SASRec model's: engine.enqueue_eval)
    STRIP_P3 = True              # the strips' products on bf16 pieces at fp32 accuracy (csrc/bert_strip.hip MODE 3; False: fp32 matrix instructions)

    def live_forward_ok(self, pl) -> bool:
        """Whether this engine's train step on `pl` encodes the live sequences only (engine._enqueue_fwd_bwd): the plain head, no
        comp module in front of the encoders, the matrix-core attention kernel's shape."""
        shp = pl.shape
        return bool(getattr(pl, "strip", False) and not self.comp and not self.dr and self.FUSED_HEAD and self.LIVE_FORWARD
                    and lib().value("amid_attn_bert_live_supported", shp.Tenc, self.D, self.H))

    def __init__(self, *args, comp: str = "", comp_bs: int = 0, comp_threshold: float = 0.5, **kw):
        """comp = "inc" / "itc": BERT4Rec(isInC=True) / (isItC=True) with bs = comp_bs and threshold1 / threshold2 = comp_threshold:
        the comp module runs on the gathered rows in FRONT of the encoders (model_seq.py:283-294; csrc/innercomp.hip), which then
        see 2T tokens per row under the T-token key mask tiled twice; every batch must hold exactly comp_bs rows.  The reference
        itself fails with both flags (2T mask keys for 4T tokens, :294), so there is no combined mode."""
        if comp not in ("", "inc", "itc"):
            raise ValueError(f"comp must be '', 'inc' or 'itc', got {comp!r}")
        if comp and comp_bs <= 0:
            raise ValueError("comp needs comp_bs = the reference's bs argument")
        self.comp, self.comp_cross = comp, 1 if comp == "itc" else 0
        # the base plan's isInC workspace (separate 2T-token encoder input, token-group buffers) is exactly what both modes need
        super().__init__(*args, inc_bs=comp_bs if comp else 0, inc_threshold=comp_threshold, **kw)

    def _dense_names(self):
        return bert4rec_dense_names(self.hid, self.dr, self.comp, self.inc_bs)

    # BERT4Rec has no last LayerNorm: the user vectors are the plain means over time (model_seq.py:299-300)
    def _enqueue_user_vectors(self, pl) -> None:
        shp = pl.shape
        lib().call("amid_lnmean_fwd_f32", pl.x[2].data_ptr(), None, None, None, None, shp.B, shp.Tenc, self.D, 0.0, pl.u.data_ptr(), self.s)

    def _enqueue_user_vectors_bwd(self, pl) -> None:
        shp = pl.shape
        lib().call("amid_lnmean_bwd_f32", pl.x[2].data_ptr(), pl.du.data_ptr(), None, None, shp.B, shp.Tenc, self.D, 0.0, pl.dxbuf.data_ptr(), None,
                   self.s)

    def _alloc_model_buffers(self) -> None:
        D, F = self.D, BERT_FF
        # transposed weights per [layer][domain]: q, k, v, o (128x128), w_1^T [128,512], w_2^T [512,128]
        self.wT_sq = torch.zeros(2, 2, 4, D * D, dtype=torch.float32, device=self.device)
        self.w1T = torch.zeros(2, 2, D * F, dtype=torch.float32, device=self.device)
        self.w2T = torch.zeros(2, 2, F * D, dtype=torch.float32, device=self.device)

    def enqueue_forward(self, pl: BertPlan, train: bool, with_loss: bool, sum_loss: bool = True) -> None:
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T, NI, M = shp.B, shp.Tenc, shp.NI, shp.M
        st = self.step_state.data_ptr()
        tr = 1 if train else 0
        fp = self.dense
        # the train step's own loss reads only the sequence (domain_id[b], b) of every sample (engine._enqueue_fwd_bwd): with live_fwd the
        # forward encodes nothing else
        lv = self._live_list(pl)
        live_fwd = lv is not None and getattr(self, "_live_fwd", False)
        lf = lv if live_fwd else None
        if lv is not None and not getattr(pl, "live_packed", False):
            L.call("amid_live_list_i32", pl.domain.data_ptr(), B, pl.live.data_ptr(), s)
        # model_seq.py:288: ONE mask, from domain 2, for both encoders
        if self.comp:      # :286 / :294 the T-token mask tiled twice over the 2T keys; the comp module's token group behind each row
            c = self.comp
            L.call("amid_key_keep_tiled_u8", pl.in_seq_d2.data_ptr(), B, shp.T, 2, pl.key_keep.data_ptr(), s)
            L.call("amid_gather_rows_f32", self.table.data_ptr(), self.n_rows, D, pl.idx_all.data_ptr(), 0, shp.n_idx, pl.xg.data_ptr(), None, s)
            L.call("amid_bert_comp_score_f32", pl.xg.data_ptr(), B, shp.T, D, self.comp_cross, pl.inc_s.data_ptr(), s)
            wts = (self._pp(c + "_d{d}.trans_nn.weight"), self._pp(c + "_d{d}.trans_nn.bias"), self._pp(c + "_d{d}.trans_bs.weight"),
                   self._pp(c + "_d{d}.trans_bs.bias"))
            out = (pl.inc_gate.data_ptr(), pl.inc_S.data_ptr(), pl.inc_Z.data_ptr(), pl.inc_sw.data_ptr(), pl.x[0].data_ptr(), s)
            if getattr(pl, "inc_world", 1) > 1:
                # data parallel (as SasrecEngine's isInC branch): the softmax over the batch and Linear(bs, 1) span the GLOBAL batch -- the
                # ranks all-gather their scores, form their rows' gates and partial token sums, all-reduce the sums and finish Z alike
                ex = self._inc_exchange(pl)
                self._coll(lambda: [ex.all_gather_packed(pl.inc_s[g], pl.inc_s_g[g]) for g in (0, 1)])
                shard = (B, shp.T, D, self.inc_bs, ex.rank * B)
                L.call("amid_bert_comp_fwd_shard_f32", pl.xg.data_ptr(), pl.inc_s_g.data_ptr(), *wts, self.inc_threshold, self.comp_cross, *shard, 1, *out)
                self._coll(lambda: ex.all_reduce_dense(pl.inc_S))
                L.call("amid_bert_comp_fwd_shard_f32", pl.xg.data_ptr(), pl.inc_s_g.data_ptr(), *wts, self.inc_threshold, self.comp_cross, *shard, 2, *out)
            else:
                L.call("amid_bert_comp_fwd_f32", pl.xg.data_ptr(), pl.inc_s.data_ptr(), *wts, self.inc_threshold, self.comp_cross, B, shp.T, D, *out)
        else:
            if not pl.strip:      # (strip path: the key mask rides in the first strip launch, _enqueue_blocks_strip)
                L.call("amid_key_keep_u8", pl.in_seq_d2.data_ptr(), B * T, pl.key_keep.data_ptr(), s)
            self._enqueue_k1(pl, None, None, None, 0, 0.0, lf)      # plain gather: no positional table, no embedding dropout, no "== 0" mask
        if pl.strip:
            self._enqueue_blocks_strip(pl, lf, st, tr)
        else:
            self._enqueue_blocks_rowtile(pl, st, tr)
        items = pl.xg.data_ptr() + 4 * 2 * shp.Mi * D
        if self.dr:                                  # three heads (model_seq.py:301-305) on the plain means
            if getattr(self, "_fuse_scorers", False) and with_loss and not sum_loss:
                self._enqueue_user_vectors(pl)       # the scorers run as ONE forward + loss + backward launch in enqueue_backward
            else:
                self._enqueue_head_dr_fwd(pl, items, with_loss)
            return
        if getattr(self, "_fuse_head", False) and with_loss and not sum_loss:
            return                                   # train step: the head runs as ONE forward + backward launch in enqueue_backward
        L.call("amid_head_fwd_f32", pl.x[2].data_ptr(), None, None, items, fp.ptr("predictModule.fc.0.weight"), fp.ptr("predictModule.fc.0.bias"),
               fp.ptr("predictModule.fc.2.weight"), fp.ptr("predictModule.fc.2.bias"), pl.labels.data_ptr() if with_loss else None,
               pl.domain.data_ptr() if with_loss else None, B, T, NI, D, self.hid, 0.0, pl.u.data_ptr(), pl.p1.data_ptr(), pl.p2.data_ptr(),
               pl.dp1.data_ptr() if with_loss else None, pl.dp2.data_ptr() if with_loss else None,
               pl.loss_part.data_ptr() if with_loss else None, s)
        if with_loss and sum_loss:
            L.call("amid_sum_vector_f32", pl.loss_part.data_ptr(), B, pl.loss.data_ptr(), s)

    def _block_ptrs(self, l: int):
        """Host pointer arrays (domain 0, domain 1) of block l's forward parameters (cached)."""
        key = ("bert_blk", l)
        c = self._ptr_cache.get(key)
        if c is None:
            fp = self.dense
            pre = f"transform{{d}}.{l}"
            c = dict(la1=self._pp(pre + ".input_sublayer.norm.a_2"), lb1=self._pp(pre + ".input_sublayer.norm.b_2"),
                     w3=ptr_array([fp.ptr(f"transform{d}.{l}.attention.linear_layers.{j}.weight") for j in range(3) for d in (1, 2)]),
                     b3=ptr_array([fp.ptr(f"transform{d}.{l}.attention.linear_layers.{j}.bias") for j in range(3) for d in (1, 2)]),
                     wo=self._pp(pre + ".attention.output_linear.weight"), bo=self._pp(pre + ".attention.output_linear.bias"),
                     la2=self._pp(pre + ".output_sublayer.norm.a_2"), lb2=self._pp(pre + ".output_sublayer.norm.b_2"),
                     w1=self._pp(pre + ".feed_forward.w_1.weight"), b1=self._pp(pre + ".feed_forward.w_1.bias"),
                     w2=self._pp(pre + ".feed_forward.w_2.weight"), b2=self._pp(pre + ".feed_forward.w_2.bias"))
            self._ptr_cache[key] = c
        return c

    def _transpose_lists(self):
        """(src pointers, dst pointers, rows, cols) of the 24 weight transposes of a step (cached host arrays)."""
        c = self._ptr_cache.get("bert_tr")
        if c is None:
            fp, D, F = self.dense, self.D, BERT_FF
            src, dst, rows, cols = [], [], [], []
            for l in (0, 1):
                for g in (0, 1):
                    pre = f"transform{g + 1}.{l}"
                    for j in range(3):
                        src.append(fp.ptr(f"{pre}.attention.linear_layers.{j}.weight")); dst.append(self.wT_sq[l, g, j].data_ptr()); rows.append(D); cols.append(D)
                    src.append(fp.ptr(f"{pre}.attention.output_linear.weight")); dst.append(self.wT_sq[l, g, 3].data_ptr()); rows.append(D); cols.append(D)
                    src.append(fp.ptr(f"{pre}.feed_forward.w_1.weight")); dst.append(self.w1T[l, g].data_ptr()); rows.append(F); cols.append(D)
                    src.append(fp.ptr(f"{pre}.feed_forward.w_2.weight")); dst.append(self.w2T[l, g].data_ptr()); rows.append(D); cols.append(F)
            c = (ptr_array(src), ptr_array(dst), (ctypes.c_int * len(rows))(*rows), (ctypes.c_int * len(cols))(*cols))
            self._ptr_cache["bert_tr"] = c
        return c

    N_TILE_IMG = 24              # tile images per (block, domain): q k v o | w_1's four | w_2's four | the same twelve transposed

    def _p3(self, pl) -> bool:
        return bool(self.STRIP_P3 and pl.strip and self.compute == "f32")

    def _tile_images(self):
        """(host spec arrays of the step's 96 weight-tile images, the image buffer [2 blocks][2 domains][24][3][D D] bf16): tile order per
        (block, domain) = Wq Wk Wv Wo, W1's four row blocks, W2's four column blocks, then the transposes of the same twelve
        (include/amid_hip.h amid_bert_weight_images_f32)."""
        c = self._ptr_cache.get("bert_img")
        if c is None:
            fp, D, F = self.dense, self.D, BERT_FF
            buf = torch.empty(2, 2, self.N_TILE_IMG, 3, D * D, dtype=torch.bfloat16, device=self.device)
            src, ld, trn = [], [], []
            for l in (0, 1):
                for g in (0, 1):
                    pre = f"transform{g + 1}.{l}"
                    w1, w2 = fp.ptr(f"{pre}.feed_forward.w_1.weight"), fp.ptr(f"{pre}.feed_forward.w_2.weight")
                    for t in (0, 1):
                        for j in range(3):
                            src.append(fp.ptr(f"{pre}.attention.linear_layers.{j}.weight")); ld.append(D); trn.append(t)
                        src.append(fp.ptr(f"{pre}.attention.output_linear.weight")); ld.append(D); trn.append(t)
                        for k in range(4):
                            src.append(w1 + 4 * k * D * D); ld.append(D); trn.append(t)          # w_1 [512][128]: rows 128 k ...
                        for k in range(4):
                            src.append(w2 + 4 * k * D); ld.append(F); trn.append(t)              # w_2 [128][512]: columns 128 k ...
            n = len(src)
            c = (ptr_array(src), (ctypes.c_int * n)(*ld), (ctypes.c_int * n)(*trn), n, buf)
            self._ptr_cache["bert_img"] = c
        return c

    def _img(self, l: int, g: int, i: int) -> int:
        """Device address of tile image i of (block l, domain g)."""
        return self._tile_images()[4][l, g, i].data_ptr()

    def _enqueue_k1(self, pl, pos0, pos1, tmq, tr: int, p_drop: float, lf) -> None:
        """The plain gather; with the strips on bf16 pieces its extra workgroups write the step's 96 tile images (amid_embed_fwd_tiles_f32:
        what amid_bert_weight_images_f32 does in a launch of its own, 11 - 13 us)."""
        if not (self._p3(pl) and pos0 is None) or (lf is not None and getattr(pl, "compact", False)) or getattr(pl, "fold_catchup", False):
            pl.tiles_written = False            # (long lists: K1 also writes the compact index list -- the tiles take their own launch)
            return super()._enqueue_k1(pl, pos0, pos1, tmq, tr, p_drop, lf)
        shp = pl.shape
        src, ld, trn, n, buf = self._tile_images()
        lib().call("amid_embed_fwd_tiles_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), None, None, shp.B, shp.Tenc, self.D, shp.B * shp.NI,
                   pl.xg.data_ptr(), None, self.step_state.data_ptr(), 0, 0.0, lf, src, ld, trn, n, 3, buf.data_ptr(), self.s)
        pl.tiles_written = True
        if getattr(self, "_sort_owed", False):      # (engine.py _enqueue_k1: a deferred side-stream sort starts behind K1)
            self.enqueue_sort(pl)

    def _enqueue_tile_images(self) -> None:
        src, ld, trn, n, buf = self._tile_images()
        lib().call("amid_bert_weight_images_f32", src, ld, trn, n, buf.data_ptr(), self.s)

    def _block_img_ptrs(self, l: int):
        key = ("bert_blk_img", l)
        c = self._ptr_cache.get(key)
        if c is None:
            I = self._img
            c = dict(w3=ptr_array([I(l, g, j) for j in range(3) for g in (0, 1)]), wo=ptr_array([I(l, g, 3) for g in (0, 1)]),
                     w1=ptr_array([I(l, g, 4) for g in (0, 1)]), w2=ptr_array([I(l, g, 8) for g in (0, 1)]),
                     wT3=ptr_array([I(l, g, 12 + j) for j in range(3) for g in (0, 1)]), woT=ptr_array([I(l, g, 15) for g in (0, 1)]),
                     w1T=ptr_array([I(l, g, 16) for g in (0, 1)]), w2T=ptr_array([I(l, g, 20) for g in (0, 1)]))
            self._ptr_cache[key] = c
        return c

    def _enqueue_blocks_strip(self, pl: BertPlan, lf, st, tr) -> None:
        """Both blocks on the strip kernels (csrc/bert_strip.hip): q / k / v of block 0, then per block the attention core and ONE launch
        for the out-projection, the feed-forward and -- block 0 -- the next block's LayerNorm + q / k / v.  lf: the live list (a train
        step over the own-domain sequences) or None."""
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T = shp.B, shp.Tenc
        live_attn = lf is not None
        p0, p1 = self._block_ptrs(0), self._block_ptrs(1)
        # the step's prologue rides in this launch: the key mask (not with a comp module: its tiled mask has a launch of its own) and,
        # when a backward will follow, the transposed weights
        p3 = self._p3(pl)
        pl.p3_fwd = p3
        if p3:                       # this step's weight tiles as three-plane images (forward tiles and, for the backward, their transposes)
            if not getattr(pl, "tiles_written", False):      # (the gather K1 of this forward wrote them with extra workgroups)
                self._enqueue_tile_images()
            pl.tiles_written = False
            i0, i1 = self._block_img_ptrs(0), self._block_img_ptrs(1)
        sfx = "_p3_f32" if p3 else "_f32"
        trl = self._transpose_lists() if pl.need_grad and not p3 else None
        pl.transposed_in_forward = trl is not None or p3
        n_tr = len(trl[2]) if trl else 0
        L.call("amid_bert_strip_qkv_fwd_pro" + sfx, pl.x[0].data_ptr(), p0["la1"], p0["lb1"], i0["w3"] if p3 else p0["w3"], p0["b3"], B, T, lf, pl.y[0].data_ptr(),
               pl.q[0].data_ptr(), pl.k[0].data_ptr(), pl.v[0].data_ptr(), None if self.comp else pl.in_seq_d2.data_ptr(), B * T,
               pl.key_keep.data_ptr(), trl[0] if trl else None, trl[1] if trl else None, trl[2] if trl else None, trl[3] if trl else None, n_tr, s)
        for l, p in ((0, p0), (1, p1)):
            if live_attn:
                L.call("amid_attn_bert_fwd_live_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.key_keep.data_ptr(), B, T, D,
                       self.H, l, st, tr, BERT_P_DROP, pl.o[l].data_ptr(), pl.stats[l].data_ptr(), lf, s)
            else:
                L.call("amid_attn_fwd_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.key_keep.data_ptr(), B, T, D, self.H,
                       0, l, st, tr, BERT_P_DROP, pl.o[l].data_ptr(), pl.stats[l].data_ptr(), s)
            nxt = ((p1["la1"], p1["lb1"], i1["w3"] if p3 else p1["w3"], p1["b3"], pl.y[1].data_ptr(), pl.q[1].data_ptr(), pl.k[1].data_ptr(), pl.v[1].data_ptr())
                   if l == 0 else (None,) * 8)
            w = (i0, i1)[l] if p3 else p
            L.call("amid_bert_strip_oproj_ffn_fwd" + sfx, pl.o[l].data_ptr(), pl.x[l].data_ptr(), w["wo"], p["bo"], p["la2"], p["lb2"], w["w1"],
                   p["b1"], w["w2"], p["b2"], B, T, lf, l, st, tr, BERT_P_DROP, pl.x1[l].data_ptr(), pl.y2[l].data_ptr(), pl.pre[l].data_ptr(),
                   pl.h[l].data_ptr(), pl.x[l + 1].data_ptr(), *nxt, s)

    def _enqueue_blocks_rowtile(self, pl: BertPlan, st, tr) -> None:
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T, M = shp.B, shp.Tenc, shp.M
        fp = self.dense
        for l in (0, 1):
            pre = f"transform{{d}}.{l}"
            w3 = ptr_array([fp.ptr(f"transform{d}.{l}.attention.linear_layers.{j}.weight") for j in range(3) for d in (1, 2)])
            b3 = ptr_array([fp.ptr(f"transform{d}.{l}.attention.linear_layers.{j}.bias") for j in range(3) for d in (1, 2)])
            L.call("amid_bert_qkv_fwd_f32" + pl.rt_suffix, pl.x[l].data_ptr(), self._pp(pre + ".input_sublayer.norm.a_2"), self._pp(pre + ".input_sublayer.norm.b_2"),
                   w3, b3, M, pl.rpt, pl.y[l].data_ptr(), pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), s)
            L.call("amid_attn_fwd_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.key_keep.data_ptr(), B, T, D, self.H, 0, l,
                   st, tr, BERT_P_DROP, pl.o[l].data_ptr(), pl.stats[l].data_ptr(), s)
            L.call("amid_bert_oproj_fwd_f32" + pl.rt_suffix, pl.o[l].data_ptr(), pl.x[l].data_ptr(), self._pp(pre + ".attention.output_linear.weight"),
                   self._pp(pre + ".attention.output_linear.bias"), M, pl.rpt, l, st, tr, BERT_P_DROP, pl.x1[l].data_ptr(), s)
            L.call("amid_bert_ffn1_fwd_f32" + pl.rt_suffix, pl.x1[l].data_ptr(), self._pp(pre + ".output_sublayer.norm.a_2"), self._pp(pre + ".output_sublayer.norm.b_2"),
                   self._pp(pre + ".feed_forward.w_1.weight"), self._pp(pre + ".feed_forward.w_1.bias"), M, pl.rpt, l, st, tr, BERT_P_DROP,
                   pl.y2[l].data_ptr(), pl.pre[l].data_ptr(), pl.h[l].data_ptr(), s)
            L.call("amid_bert_ffn2_fwd_f32" + pl.rt_suffix, pl.h[l].data_ptr(), pl.x1[l].data_ptr(), self._pp(pre + ".feed_forward.w_2.weight"),
                   self._pp(pre + ".feed_forward.w_2.bias"), M, pl.rpt, l, st, tr, BERT_P_DROP, pl.x[l + 1].data_ptr(), s)

    def enqueue_backward(self, pl: BertPlan, train: bool) -> None:
        L, s, shp, D, F = lib(), self.s, pl.shape, self.D, BERT_FF
        B, T, NI, M = shp.B, shp.Tenc, shp.NI, shp.M
        st = self.step_state.data_ptr()
        tr = 1 if train else 0
        fp = self.dense
        # transposed weights (rectangular ones included) -- unless this step's forward already made them (strip path: riders of its first launch)
        if not getattr(pl, "transposed_in_forward", False):
            tsrc, tdst, trows, tcols = self._transpose_lists()
            L.call("amid_transpose_rect_f32", tsrc, tdst, trows, tcols, len(trows), s)
        pl.transposed_in_forward = False
        items = pl.xg.data_ptr() + 4 * 2 * shp.Mi * D
        ditems = pl.dxg.data_ptr() + 4 * 2 * shp.Mi * D
        if self.dr and getattr(self, "_fuse_scorers", False):
            self._enqueue_scorers_fused(pl, items, ditems, None, None, 0)
            self._enqueue_user_vectors_bwd(pl)
        elif self.dr:
            self._enqueue_head_dr_bwd(pl, items, ditems)
        elif getattr(self, "_fuse_head", False):
            own = getattr(self, "_live_fwd", False) and self._live_list(pl) is not None      # only the own-domain sequences were encoded
            L.call("amid_head_fwd_bwd_own_f32" if own else "amid_head_fwd_bwd_f32", pl.x[2].data_ptr(), None, None, items, fp.ptr("predictModule.fc.0.weight"),
                   fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"), fp.ptr("predictModule.fc.2.bias"),
                   pl.labels.data_ptr(), pl.domain.data_ptr(), B, T, NI, D, self.hid, 0.0, pl.u.data_ptr(), pl.p1.data_ptr(), pl.p2.data_ptr(),
                   pl.dp1.data_ptr(), pl.dp2.data_ptr(), pl.loss_part.data_ptr(), pl.dxbuf.data_ptr(), ditems, None, pl.sc_part.data_ptr(),
                   None, None, 0, s)
        else:
            L.call("amid_head_bwd_f32", pl.x[2].data_ptr(), None, pl.u.data_ptr(), items, fp.ptr("predictModule.fc.0.weight"),
               fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"), fp.ptr("predictModule.fc.2.bias"), pl.p1.data_ptr(),
               pl.p2.data_ptr(), pl.dp1.data_ptr(), pl.dp2.data_ptr(), B, T, NI, D, self.hid, 0.0, pl.dxbuf.data_ptr(), ditems, None,
               pl.sc_part.data_ptr(), None, None, 0, s)
        # the train step's own backward walks the LIVE sequences only (the loss sends no gradient into the other domain's encoder of a
        # sample); a comp module in front of the encoders reads the gradient of EVERY encoder-input row: all rows then
        lv = None if self.comp else self._live_list(pl)
        live_attn = lv is not None and bool(L.value("amid_attn_bert_live_supported", T, D, self.H))

        def attn_bwd(l):
            if live_attn:
                L.call("amid_attn_bert_bwd_live_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.o[l].data_ptr(),
                       pl.stats[l].data_ptr(), pl.d_o.data_ptr(), pl.key_keep.data_ptr(), B, T, D, self.H, l, st, tr, BERT_P_DROP,
                       pl.dq.data_ptr(), pl.dk.data_ptr(), pl.dv.data_ptr(), lv, s)
            else:
                L.call("amid_attn_bwd_rows_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.o[l].data_ptr(), pl.stats[l].data_ptr(),
                       pl.d_o.data_ptr(), pl.key_keep.data_ptr(), B, T, D, self.H, 0, l, st, tr, BERT_P_DROP, pl.dq.data_ptr(), pl.dk.data_ptr(),
                       pl.dv.data_ptr(), self._own_rows(pl), s)

        def wgrad(l):
            # weight-gradient tiles need dx-independent operands only: launched before the q / k / v backward overwrites its buffers
            dy = [pl.dq.data_ptr(), pl.dk.data_ptr(), pl.dv.data_ptr(), pl.dt.data_ptr()]
            xx = [pl.y[l].data_ptr()] * 3 + [pl.o[l].data_ptr()]
            ldy, ldx = [D] * 4, [D] * 4
            old, ogr, oco = [D] * 8 + [F] * 4, list(range(8)) + [8] * 4, [0] * 8 + [c * D for c in range(4)]
            for c in range(4):                                   # w_1 tile c: dY = dpre[:, c*128:], X = y2
                dy.append(pl.dpre.data_ptr() + 4 * c * D); xx.append(pl.y2[l].data_ptr()); ldy.append(F); ldx.append(D)
            for c in range(4):                                   # w_2 tile c: dY = dz, X = h[:, c*128:]
                dy.append(pl.dz.data_ptr()); xx.append(pl.h[l].data_ptr() + 4 * c * D); ldy.append(D); ldx.append(F)
            L.call("amid_bert_wgrad_mode_f32", ptr_array(dy), ptr_array(xx), (ctypes.c_int * N_ENT)(*ldy), (ctypes.c_int * N_ENT)(*ldx),
                   (ctypes.c_int * N_ENT)(*old), (ctypes.c_int * N_ENT)(*ogr), (ctypes.c_int * N_ENT)(*oco), N_ENT, M,
                   pl.splits, pl.w_part[l].data_ptr(), pl.b_part[l].data_ptr(), self._own_rows(pl), B, T,
                   {"9": 2, "6": 3}.get(self.WGRAD_SPLIT, 0), s)

        if pl.strip:
            # csrc/bert_strip.hip: block 1's feed-forward / out-projection chain, its attention core and weight gradients, then ONE launch
            # for block 1's q / k / v + LayerNorm backward and block 0's feed-forward / out-projection chain, ..., block 0's q / k / v
            p3 = bool(getattr(pl, "p3_fwd", False) and self._p3(pl))     # (this step's forward wrote the tile images, transposes included)
            sfx = "_p3_f32" if p3 else "_f32"

            def ffn_args(l):
                if p3:
                    im = self._block_img_ptrs(l)
                    return (pl.pre[l].data_ptr(), pl.x1[l].data_ptr(), self._pp(f"transform{{d}}.{l}.output_sublayer.norm.a_2"),
                            im["w2T"], im["w1T"], im["woT"])
                return (pl.pre[l].data_ptr(), pl.x1[l].data_ptr(), self._pp(f"transform{{d}}.{l}.output_sublayer.norm.a_2"),
                        ptr_array([self.w2T[l, g].data_ptr() for g in (0, 1)]), ptr_array([self.w1T[l, g].data_ptr() for g in (0, 1)]),
                        ptr_array([self.wT_sq[l, g, 3].data_ptr() for g in (0, 1)]))
            ffn_out = (pl.dz.data_ptr(), pl.dpre.data_ptr(), pl.dx1.data_ptr(), pl.dt.data_ptr(), pl.d_o.data_ptr())
            pre1, x11, la21, w2T1, w1T1, woT1 = ffn_args(1)
            L.call("amid_bert_strip_ffn_bwd" + sfx, pl.dxbuf.data_ptr(), pre1, x11, la21, w2T1, w1T1, woT1, B, T, lv, 1, st, tr, BERT_P_DROP, *ffn_out,
                   pl.ln2_part[1].data_ptr(), s)
            for l in (1, 0):
                attn_bwd(l)
                wgrad(l)
                wT3 = (self._block_img_ptrs(l)["wT3"] if p3 else
                       ptr_array([self.wT_sq[l, g, j].data_ptr() for j in range(3) for g in (0, 1)]))
                la1 = self._pp(f"transform{{d}}.{l}.input_sublayer.norm.a_2")
                if l == 1:
                    L.call("amid_bert_strip_qkv_bwd" + sfx, pl.dq.data_ptr(), pl.dk.data_ptr(), pl.dv.data_ptr(), pl.dx1.data_ptr(), pl.x[1].data_ptr(),
                           la1, wT3, B, T, lv, None, 0, pl.ln1_part[1].data_ptr(), *ffn_args(0), 0, st, tr, BERT_P_DROP, *ffn_out,
                           pl.ln2_part[0].data_ptr(), s)
                else:
                    dx_out = pl.dx0 if self.comp else pl.dxg
                    L.call("amid_bert_strip_qkv_bwd" + sfx, pl.dq.data_ptr(), pl.dk.data_ptr(), pl.dv.data_ptr(), pl.dx1.data_ptr(), pl.x[0].data_ptr(),
                           la1, wT3, B, T, lv, dx_out.data_ptr(), 1 if lv is not None else 0, pl.ln1_part[0].data_ptr(), None, None, None, None,
                           None, None, 0, None, 0, 0.0, None, None, None, None, None, None, s)
            pl.p3_fwd = False
        for l in ((1, 0) if not pl.strip else ()):
            pre = f"transform{{d}}.{l}"
            wsq = lambda j: ptr_array([self.wT_sq[l, g, j].data_ptr() for g in (0, 1)])      # noqa: E731
            L.call("amid_bert_ffn2_bwd_f32" + pl.rt_suffix_b, pl.dxbuf.data_ptr(), pl.pre[l].data_ptr(),
                   ptr_array([self.w2T[l, g].data_ptr() for g in (0, 1)]), M, pl.rpt_b, l, st, tr, BERT_P_DROP, pl.dz.data_ptr(), pl.dpre.data_ptr(),
                   s)
            L.call("amid_bert_ffn1_bwd_f32" + pl.rt_suffix_b, pl.dpre.data_ptr(), pl.dxbuf.data_ptr(), pl.x1[l].data_ptr(),
                   self._pp(pre + ".output_sublayer.norm.a_2"), ptr_array([self.w1T[l, g].data_ptr() for g in (0, 1)]), wsq(3), M, pl.rpt_b, l, st, tr,
                   BERT_P_DROP, pl.dx1.data_ptr(), pl.dt.data_ptr(), pl.d_o.data_ptr(), pl.ln2_part[l].data_ptr(), s)
            attn_bwd(l)
            dx_out = (pl.dx0 if self.comp else pl.dxg) if l == 0 else pl.dxbuf
            wT3 = ptr_array([self.wT_sq[l, g, j].data_ptr() for j in range(3) for g in (0, 1)])
            wgrad(l)
            L.call("amid_bert_qkv_bwd_f32" + pl.rt_suffix_b, pl.dq.data_ptr(), pl.dk.data_ptr(), pl.dv.data_ptr(), pl.dx1.data_ptr(),
                   pl.x[l].data_ptr(), self._pp(pre + ".input_sublayer.norm.a_2"), wT3, M, pl.rpt_b, dx_out.data_ptr(), pl.ln1_part[l].data_ptr(),
                   s)
        if self.comp:      # the comp modules' parameter gradients; the rows' own halves + their share of the token group -> dxg
            c, G = self.comp, self.dense.grad
            head = (pl.xg.data_ptr(), pl.dx0.data_ptr(), pl.inc_gate.data_ptr(), pl.inc_S.data_ptr(), pl.inc_sw.data_ptr(),
                    self._pp(c + "_d{d}.trans_nn.weight"), self._pp(c + "_d{d}.trans_nn.bias"), self._pp(c + "_d{d}.trans_bs.weight"),
                    self.comp_cross, B, shp.T, D)
            outs = (pl.inc_dZ.data_ptr(), pl.inc_dS.data_ptr(), pl.inc_rows.data_ptr(), self._pp(c + "_d{d}.trans_nn.weight", G),
                    self._pp(c + "_d{d}.trans_nn.bias", G), self._pp(c + "_d{d}.trans_bs.weight", G), self._pp(c + "_d{d}.trans_bs.bias", G),
                    pl.dxg.data_ptr(), s)
            if getattr(pl, "inc_world", 1) > 1:      # data parallel: the group's gradient dZ is the sum over every rank's rows (engine.py, isInC)
                ex = self._inc_exchange(pl)
                shard = (self.inc_bs, ex.rank * B)
                L.call("amid_bert_comp_bwd_shard_f32", *head, *shard, 1, 1.0 / ex.world, *outs)
                self._coll(lambda: ex.all_reduce_dense(pl.inc_dZ))
                L.call("amid_bert_comp_bwd_shard_f32", *head, *shard, 2, 1.0 / ex.world, *outs)
            else:
                L.call("amid_bert_comp_bwd_f32", *head, *outs)
        self._enqueue_grad_tail(pl)
