"""Per-shape workspace of the SASRec engine (split out of engine.py): the flat parameter buffer, the batch shape and the plan --
saved activations, gradient scratch, index workspaces and the reduce tables of the gradient tail -- allocated once per (B, T, n_items).
Mirrors no reference code: the reference lets autograd allocate (train_sr.py:190-217)."""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import lib, ptr_array

SASREC_HEADS = 8          # model_seq.py:348-350
SASREC_P_DROP = 0.5       # model_seq.py:335,350,356
SASREC_LN_EPS = 1e-8      # model_seq.py:342-353


def sasrec_dense_names(T: int, D: int, hid: int, itc_bs: int = 0, dr: bool = False, inc_bs: int = 0) -> List[Tuple[str, Tuple[int, ...]]]:
    """Non-table parameters in the reference's state_dict order; itc_bs > 0: with the InterComp
    modules of SASRec(isItC=True, bs=itc_bs) (model_seq.py:403-405, :478-480); inc_bs > 0: with the InnerComp modules of
    isInC=True (:398-401; T is then the doubled pos_emb length)."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    if inc_bs:
        for d in (1, 2):
            out.append((f"inc_d{d}.trans_nn.weight", (D, D)))
            out.append((f"inc_d{d}.trans_nn.bias", (D,)))
            out.append((f"inc_d{d}.trans_bs.weight", (1, inc_bs)))
            out.append((f"inc_d{d}.trans_bs.bias", (1,)))
    if itc_bs:
        for d in (1, 2):
            out.append((f"itc_d{d}.trans_nn.weight", (D, D)))
            out.append((f"itc_d{d}.trans_nn.bias", (D,)))
            out.append((f"itc_d{d}.trans_bs.weight", (1, itc_bs)))
            out.append((f"itc_d{d}.trans_bs.bias", (1,)))
    for d in (1, 2):
        pre = f"sac{d}"
        out.append((f"{pre}.pos_emb.weight", (T, D)))
        out.append((f"{pre}.attention_layernorms.0.weight", (D,)))
        out.append((f"{pre}.attention_layernorms.0.bias", (D,)))
        out.append((f"{pre}.attention_layernorms.1.weight", (D,)))
        out.append((f"{pre}.attention_layernorms.1.bias", (D,)))
        for l in (0, 1):
            out.append((f"{pre}.attention_layers.{l}.in_proj_weight", (3 * D, D)))
            out.append((f"{pre}.attention_layers.{l}.in_proj_bias", (3 * D,)))
            out.append((f"{pre}.attention_layers.{l}.out_proj.weight", (D, D)))
            out.append((f"{pre}.attention_layers.{l}.out_proj.bias", (D,)))
        out.append((f"{pre}.forward_layernorms.0.weight", (D,)))
        out.append((f"{pre}.forward_layernorms.0.bias", (D,)))
        out.append((f"{pre}.forward_layernorms.1.weight", (D,)))
        out.append((f"{pre}.forward_layernorms.1.bias", (D,)))
        for l in (0, 1):
            out.append((f"{pre}.forward_layers.{l}.conv1.weight", (D, D, 1)))
            out.append((f"{pre}.forward_layers.{l}.conv1.bias", (D,)))
            out.append((f"{pre}.forward_layers.{l}.conv2.weight", (D, D, 1)))
            out.append((f"{pre}.forward_layers.{l}.conv2.bias", (D,)))
        out.append((f"{pre}.last_layernorm.weight", (D,)))
        out.append((f"{pre}.last_layernorm.bias", (D,)))
    for head in ("predictModule",) + (("predict_ips", "predict_gfunc") if dr else ()):      # isDR heads: model_seq.py:411-414
        out.append((f"{head}.fc.0.weight", (hid, 2 * D)))
        out.append((f"{head}.fc.0.bias", (hid,)))
        out.append((f"{head}.fc.2.weight", (1, hid)))
        out.append((f"{head}.fc.2.bias", (1,)))
    return out


DR_HEADS = ("predictModule", "predict_ips", "predict_gfunc")


class FlatParams:
    """One flat fp32 buffer with named views (each slot padded to 4 floats = 16 B)."""

    def __init__(self, names: List[Tuple[str, Tuple[int, ...]]], device):
        self.slots: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for n, shp in names:
            numel = 1
            for s in shp:
                numel *= s
            self.slots[n] = (off, shp)
            off += (numel + 3) & ~3
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.data)
        self.m = torch.zeros_like(self.data)
        self.v = torch.zeros_like(self.data)

    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        off, shp = self.slots[name]
        numel = 1
        for s in shp:
            numel *= s
        return (self.data if buf is None else buf)[off: off + numel].view(*shp)

    def ptr(self, name: str, buf: Optional[torch.Tensor] = None, extra: int = 0) -> int:
        off, _ = self.slots[name]
        return (self.data if buf is None else buf).data_ptr() + 4 * (off + extra)


@dataclass
class Shape:
    B: int
    T: int
    NI: int          # items scored per row: 1 positive + negatives
    Te: int = 0      # tokens per sequence inside the encoder when it differs from T (isInC appends T tokens: 2T); 0 = T

    @property
    def Tenc(self) -> int:
        return self.Te or self.T

    @property
    def M(self) -> int:
        """Encoder rows per domain."""
        return self.B * self.Tenc

    @property
    def Mi(self) -> int:
        """Gathered sequence rows per domain (index layout)."""
        return self.B * self.T

    @property
    def n_idx(self) -> int:
        return 2 * self.B * self.T + self.B * self.NI


class SasrecPlan:
    """Workspace for one batch shape: saved activations, gradient scratch, index workspaces."""

    def __init__(self, eng: "SasrecEngine", shp: Shape, need_grad: bool):
        L = lib()
        self.shape = shp
        self.need_grad = need_grad
        dev, D, H, hid = eng.device, eng.D, eng.H, eng.hid
        B, T, NI, M, N = shp.B, shp.T, shp.NI, shp.M, shp.n_idx
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)   # noqa: E731
        self.rpt = L.value("amid_rows_per_tile", M)
        self.tpg = (M + self.rpt - 1) // self.rpt
        # the encoder's GEMM chains as register-resident strip kernels (csrc/sasrec_strip.hip): fp32, activations up to 2 GiB each;
        # one tile geometry (64-row tiles) for every sequence and for the live sequences of a train step alike
        # compute = "bf16": the strip path when the one-launch forward covers the shape -- its twelve projection products then run on the
        # bf16 matrix cores (amid_sas_seq_fwd_bf16w_f32); other shapes keep the row-tile kernels' bf16 mode
        self.strip = bool(eng.STRIP_KERNELS and 2 * M * D * 4 <= 0x7FFFFFF0 and
                          (eng.compute == "f32" or (eng.BF16_STRIP and not getattr(eng, "inc_bs", 0)
                                                    and L.value("amid_sas_seq_supported", B, shp.Tenc, D, H))))
        if self.strip:
            self.stpg = -(-M // L.value("amid_sas_strip_tile_rows"))
        self.live = torch.zeros(B + 1, dtype=torch.int32, device=dev)       # amid_live_list_i32: the step's live sequences
        # the train step's compact index list over the live sequences + items (amid_lazy_adam_catchup_live_f32): ids and, for every
        # entry, the row of its gradient in the full [2 B T + items] layout
        self.n_compact = B * self.shape.T + B * self.shape.NI
        self.idx_c = torch.zeros(self.n_compact, dtype=torch.int32, device=dev)
        self.row_c = torch.zeros(self.n_compact, dtype=torch.int32, device=dev)
        self.compact = False             # set per step by enqueue_prepare
        # short tiles (seq_len 20 at batch 256: 40 rows per CU) run the 48- / 80-row builds of the row-tile kernels (csrc/tile_gemm.h)
        self.rt_suffix = ("_rt3" if self.rpt <= 48 else "_rt5" if self.rpt <= 80 else "") if eng.SHORT_TILE_BUILDS else ""
        # static inputs (graph-replay safe): ONE int64 buffer so a batch arrives with a single copy
        #   [i_node B | neg B*(NI-1) | seq_d1 B*T | seq_d2 B*T | domain B | labels B*NI fp32 (packed two per word)]
        n_lab_words = (B * NI + 1) // 2
        dr = bool(getattr(eng, "dr", False))
        self.in_words = B + B * (NI - 1) + 2 * B * T + B + n_lab_words + (B if dr else 0)       # DR: + ob_label [B] at the end
        self.in_pack = torch.zeros(self.in_words, dtype=torch.int64, device=dev)
        o = 0
        self.in_i_node = self.in_pack[o:o + B]; o += B
        self.in_neg = self.in_pack[o:o + B * (NI - 1)].view(B, NI - 1); o += B * (NI - 1)
        self.in_seq_d1 = self.in_pack[o:o + B * T].view(B, T); o += B * T
        self.in_seq_d2 = self.in_pack[o:o + B * T].view(B, T); o += B * T
        self.domain = self.in_pack[o:o + B]; o += B
        self.labels = self.in_pack[o:o + n_lab_words].view(torch.float32)[: B * NI].view(B, NI); o += n_lab_words
        self.in_ob = self.in_pack[o:o + B] if dr else None
        self.pools = {}                               # SasrecEngine.set_input_pool: (Adam state, objective) -> [pool, phase]
        self.idx_all = torch.zeros(N, dtype=torch.int32, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        # forward
        self.xg = f(N, D)
        inc = int(getattr(eng, "inc_bs", 0))
        # the gathered seq rows ARE the encoder input, except with isInC (the encoder input has 2T tokens per row)
        self.x = [self.xg[: 2 * M] if not inc else f(2 * M, D), f(2 * M, D), f(2 * M, D)]
        if inc:
            # B == bs: the whole batch on this GPU.  world * B == bs: a data-parallel shard of a global batch of bs rows -- the module's
            # softmax over the batch and Linear(bs, 1) span the GLOBAL batch (engine._enqueue_inc_fwd / _bwd: the ranks all-gather
            # their scores and all-reduce the partial token sums S and dZ)
            if B != inc and (inc % B or inc // B > 64):
                raise ValueError(f"isInC: the batch must hold exactly bs = {inc} rows (trans_bs is Linear(bs, 1) over the batch, "
                                 f"model_seq.py:457) or an equal data-parallel shard of them, got {B}")
            self.inc_world = inc // B
            if self.inc_world > 1:
                self.inc_s_g = f(2, inc)
            self.inc_s, self.inc_gate, self.inc_sw = f(2, B), f(2, B), f(2)
            self.inc_S, self.inc_Z = f(2, T, D), f(2, T, D)
        self.q = [f(2 * M, D) for _ in range(2)]
        self.k = [f(2 * M, D) for _ in range(2)]
        self.v = [f(2 * M, D) for _ in range(2)]
        self.o = [f(2 * M, D) for _ in range(2)]
        self.stats = [f(2 * M, H, 2) for _ in range(2)]
        self._alloc_model_fwd(eng, f)
        self.u = f(2, B, D)
        if getattr(eng, "itc_bs", 0):
            # B == bs: the whole batch on this GPU.  world * B == bs: a data-parallel shard of a global batch of bs rows (InterComp's softmax
            # and Linear(bs, 1) run over the GLOBAL batch: the ranks all-gather the B pair-max scalars and user vectors of every shard and
            # each evaluates the module on all bs rows, engine._enqueue_user_vectors)
            if B != eng.itc_bs and (eng.itc_bs % B or eng.itc_bs // B > 64):
                raise ValueError(f"isItC: the batch must hold exactly bs = {eng.itc_bs} rows (trans_bs is Linear(bs, 1) over the batch, "
                                 f"model_seq.py:480) or an equal data-parallel shard of them, got {B}")
            self.itc_world = eng.itc_bs // B
            Bg = eng.itc_bs
            self.u_raw, self.du_raw = f(2, B, D), f(2, B, D)
            self.itc_s, self.itc_gate, self.itc_z, self.itc_sw = f(B), f(Bg), f(2, D), f(2)
            if self.itc_world > 1:      # the global batch's copies: gathered inputs, the module's outputs for all bs rows
                self.u_raw_g, self.u_g, self.du_g, self.du_raw_g, self.itc_s_g = f(2, Bg, D), f(2, Bg, D), f(2, Bg, D), f(2, Bg, D), f(Bg)
        self.p1 = f(B, NI)
        self.p2 = f(B, NI)
        self.dp1 = torch.zeros(B, NI, dtype=torch.float32, device=dev)
        self.dp2 = torch.zeros(B, NI, dtype=torch.float32, device=dev)
        self.loss_part = torch.zeros(B, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        if dr:            # the two extra heads' outputs / output gradients, and per-row partials of (loss_cls, loss_dr_e, loss_dr_r)
            self.ips1, self.ips2, self.g1, self.g2 = (f(B, NI) for _ in range(4))
            self.dips1, self.dips2, self.dg1, self.dg2 = (torch.zeros(B, NI, dtype=torch.float32, device=dev) for _ in range(4))
            self.dr_loss_part = torch.zeros(B, 3, dtype=torch.float32, device=dev)
            self.dr_losses = torch.zeros(3, dtype=torch.float32, device=dev)
        if not need_grad:
            return
        # backward
        self.dxg = f(N, D)
        if inc:
            self.dx0 = f(2 * M, D)                      # encoder-input gradient; its first halves + InnerComp's share -> dxg
            self.inc_dZ, self.inc_dS, self.inc_rows = f(2, T, D), f(2, T, D), f(2, T, 2)
        self.dxbuf = f(2 * M, D)
        self.du = f(2, B, D)
        self.d_o = f(2 * M, D)
        self.dq, self.dk, self.dv = f(2 * M, D), f(2 * M, D), f(2 * M, D)
        tpg_ln = self.stpg if self.strip else self.tpg
        self.ln1_part = [f(2 * tpg_ln, 2, D) for _ in range(2)]
        self.ln2_part = [f(2 * tpg_ln, 2, D) for _ in range(2)]
        # the train step's own backward walks the LIVE sequences only (engine._own_rows): half the rows, re-tiled over the CUs
        # (csrc/sasrec_bwd.hip TileGeomB::row_domain); its LayerNorm partials have their own slots and reduce table
        self.live_rows = bool(self.LIVE_ROWS_BWD)
        if self.live_rows and self.strip:          # the strip kernels tile live and all rows alike: same partial slots, same reduce table
            self.ln1_part_v, self.ln2_part_v = self.ln1_part, self.ln2_part
        elif self.live_rows:
            self.rpt_v = L.value("amid_rows_per_tile", (M + 1) // 2)
            self.tpg_v = (M + self.rpt_v - 1) // self.rpt_v
            self.rt_suffix_v = (("_rt3" if self.rpt_v <= 48 else "_rt4" if self.rpt_v <= 64 else "_rt5" if self.rpt_v <= 80 else "")
                                if eng.SHORT_TILE_BUILDS else "")
            self.ln1_part_v = [f(2 * self.tpg_v, 2, D) for _ in range(2)]
            self.ln2_part_v = [f(2 * self.tpg_v, 2, D) for _ in range(2)]
        # the fused per-sequence backward (csrc/sasrec_strip.hip seq_bwd_kernel) tiles one live sequence per workgroup: its LayerNorm
        # partials have a slot per sequence and their own reduce table
        self.seq_bwd = bool(self.strip and self.live_rows and getattr(eng, "SEQ_BACKWARD", "0") not in ("0", False) and not inc
                            and L.value("amid_sas_seq_bwd_supported", B, shp.Tenc, D, H))
        if self.seq_bwd:
            self.ln1_part_s = [f(2 * B, 2, D) for _ in range(2)]
            self.ln2_part_s = [f(2 * B, 2, D) for _ in range(2)]
        self.last_part = f(2 * B, 2, D)
        self._alloc_model_bwd(eng, f)
        self.sc_P = L.value("amid_scorer_part_floats", D, hid)
        self.sc_part = f(B, self.sc_P)
        if dr:
            self.sc_part_ips, self.sc_part_g = f(B, self.sc_P), f(B, self.sc_P)
        # sparse side
        self.sort_ws = torch.zeros(L.value("amid_sort_unique_workspace_bytes", N), dtype=torch.uint8, device=dev)
        self.pos_sorted = torch.zeros(N, dtype=torch.int32, device=dev)
        self.uniq_ids = torch.zeros(N, dtype=torch.int32, device=dev)
        self.seg_off = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        self.seg_of = torch.zeros(N, dtype=torch.int32, device=dev)
        self.n_uniq = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seg_ws = torch.empty(L.value("amid_segreduce_workspace_bytes", N, D), dtype=torch.uint8, device=dev)
        self.uniq_grad = f(N, D)
        self.tail_ticket = torch.zeros(16, dtype=torch.int32, device=dev)      # amid_grad_tail_opt_f32's "last chunk block" ticket: zero between launches
        self.red_entries, self.red_n, self.red_max = self._build_reduce_table(eng)
        if self.live_rows and self.strip:
            self.red_entries_v, self.red_n_v, self.red_max_v = self.red_entries, self.red_n, self.red_max
        elif self.live_rows:
            self.red_entries_v, self.red_n_v, self.red_max_v = self._build_reduce_table(eng, live=True)
        if self.seq_bwd:
            self.red_entries_s, self.red_n_s, self.red_max_s = self._build_reduce_table(eng, live=True, seq=True)
        self.graph = None
        self.graphs = {}

    # ---- model-specific pieces (BertPlan overrides these three) -------------------------------------
    def _alloc_model_fwd(self, eng: "SasrecEngine", f) -> None:
        M, D = self.shape.M, eng.D
        self.tmq = torch.zeros(2 * M, D // 4, dtype=torch.uint8, device=eng.device)
        self.qn = [f(2 * M, D) for _ in range(2)]
        self.r = [f(2 * M, D) for _ in range(2)]
        self.y = [f(2 * M, D) for _ in range(2)]
        self.h = [f(2 * M, D) for _ in range(2)]

    def _alloc_model_bwd(self, eng: "SasrecEngine", f) -> None:
        M, D, B, T = self.shape.M, eng.D, self.shape.B, self.shape.Tenc
        # per LAYER copies of the six dY tensors of the weight gradients: both layers' weight gradients run as one launch at the
        # end of backward (2 layers x 2 domains x 6 weights x 21 splits = 504 workgroups, two per CU)
        self.dpre1, self.dpre2, self.dr = ([f(2 * M, D), f(2 * M, D)] for _ in range(3))
        self.dq_l, self.dk_l, self.dv_l = [self.dq, f(2 * M, D)], [self.dk, f(2 * M, D)], [self.dv, f(2 * M, D)]
        self.splits = max(1, min(int(self.WGRAD_SPLITS or 21), M // 128))
        self.w_part = [f(2, 6, self.splits, D * D) for _ in range(2)]
        self.b_part = [f(2, 6, self.splits, D) for _ in range(2)]
        self.pos_splits = max(1, min(8, B // 16))
        self.dpos_part = f(self.pos_splits, 2, T, D)

    LIVE_ROWS_BWD = True       # BertPlan: False (its backward kernels take no row_domain hint)
    WGRAD_SPLITS = 0           # row splits of the weight-gradient launch (0: the default, 21 -- BertPlan: 21 / 10 by the products' mode)

    def _build_reduce_table(self, eng: "SasrecEngine", live: bool = False, seq: bool = False, pos: bool = True):
        L = lib()
        D, hid, B = eng.D, eng.hid, self.shape.B
        fp, G = eng.dense, eng.dense.grad
        ent: List[Tuple[int, int, int, int, int]] = []      # src_ptr, dst_ptr, stride, n_part, count

        def add(src_t: torch.Tensor, src_off: int, dst_ptr: int, stride: int, n_part: int, count: int):
            ent.append((src_t.data_ptr() + 4 * src_off, dst_ptr, stride, n_part, count))

        if not pos:         # (the live-sequence step's tail sums the position rows' gradients from the rows: amid_grad_tail_live_f32)
            self._model_reduce_entries(eng, add, pos=False)      # (strip plans: the live tiles share the slots of the full tiling)
        elif seq:
            self._model_reduce_entries(eng, add, seq=True)
        elif live:
            self._model_reduce_entries(eng, add, live=True)
        else:
            self._model_reduce_entries(eng, add)
        P = self.sc_P
        heads = [("predictModule", self.sc_part)]
        if getattr(eng, "dr", False):
            heads += [("predict_ips", self.sc_part_ips), ("predict_gfunc", self.sc_part_g)]
            for c in range(3):                                                           # loss_cls, loss_dr_e, loss_dr_r
                ent.append((self.dr_loss_part.data_ptr() + 4 * c, self.dr_losses.data_ptr() + 4 * c, 3, B, 1))
        else:
            ent.append((self.loss_part.data_ptr(), self.loss.data_ptr(), 1, B, 1))      # loss = sum of the per-row partials
        for head, part in heads:
            if not pos:         # (the folded step: the gradient tail forms the scorer's weight gradients from per-sample hidden gradients)
                continue
            add(part, 0, fp.ptr(f"{head}.fc.0.weight", G), P, B, hid * 2 * D)
            add(part, hid * 2 * D, fp.ptr(f"{head}.fc.0.bias", G), P, B, hid)
            add(part, hid * 2 * D + hid, fp.ptr(f"{head}.fc.2.weight", G), P, B, hid)
            add(part, hid * 2 * D + 2 * hid, fp.ptr(f"{head}.fc.2.bias", G), P, B, 1)
        # the blocks of an entry with many partials (the head's per-row partials: one per batch row) run the longest chains of
        # dependent loads: dispatch them first, so that they do not form the tail of the launch
        ent.sort(key=lambda e: -e[3])
        # algorithmic HBM bytes of the partial-sum reduce (every partial read once, every sum written once): bench.py prices the launch
        sfx = "_t" if not pos else "_s" if seq else "_v" if live else ""
        setattr(self, "red_bytes" + sfx, sum(4 * (n + 1) * c for *_, n, c in ent))
        # the parameters whose gradient NO entry of this table produces (kernels write them straight into dense.grad: the comp modules'),
        # as ONE range of the flat buffer [lo, hi) -- what amid_grad_tail_opt_f32 applies Adam to beside the entries' slices; () when every
        # parameter is covered, None when the uncovered slots do not form one run (the caller then keeps the optimizer launch)
        import numpy as np
        cov = np.zeros(fp.numel, dtype=bool)
        g0 = G.data_ptr()
        for s_, d_, st_, n_, c_ in ent:
            o = (d_ - g0) // 4
            if 0 <= o < fp.numel and (d_ - g0) % 4 == 0:
                cov[o:o + c_] = True
        names = list(fp.slots)
        unc = []
        for i, nm in enumerate(names):
            off, shp_ = fp.slots[nm]
            cnt = int(np.prod(shp_)) if len(shp_) else 1
            if not cov[off:off + cnt].all():
                unc.append(i)
        if not unc:
            left = ()
        elif unc == list(range(unc[0], unc[-1] + 1)):
            off_l, shp_l = fp.slots[names[unc[-1]]]
            left = (fp.slots[names[unc[0]]][0], min(fp.numel, (off_l + (int(np.prod(shp_l)) if len(shp_l) else 1) + 3) & ~3))
        else:
            left = None
        setattr(self, "red_left" + sfx, left)
        esz = L.value("amid_reduce_entry_bytes")
        host = (ctypes.c_ubyte * (esz * len(ent)))()
        for i, (s, d, st, n, c) in enumerate(ent):
            L.call("amid_reduce_entry_pack", ctypes.addressof(host), i, s, d, st, n, c)
        # blocks per entry for the gradient tail (amid_grad_tail_f32 blk_off): what the entry's size needs -- 1024 elements per block
        # when it has at most 32 aligned partials (csrc/reduce_partials.h), 128 otherwise
        off = [0]
        for s, d, st, n, c in ent:
            per = 1024 if (n <= 32 and c % 4 == 0 and st % 4 == 0 and s % 16 == 0 and d % 16 == 0) else 128
            off.append(off[-1] + min(512, (c + per - 1) // per))
        blk = torch.tensor(off, dtype=torch.int32).to(eng.device)
        setattr(self, "red_blk" + sfx, (blk, off[-1]))
        return torch.frombuffer(bytearray(host), dtype=torch.uint8).to(eng.device), len(ent), max(c for *_, c in ent)

    def _model_reduce_entries(self, eng: "SasrecEngine", add, live: bool = False, seq: bool = False, pos: bool = True) -> None:
        D, B = eng.D, self.shape.B
        fp, G = eng.dense, eng.dense.grad
        S = self.splits
        ln1, ln2, tpg = (self.ln1_part_v, self.ln2_part_v, self.tpg_v) if live else (self.ln1_part, self.ln2_part, self.tpg)
        if getattr(self, "strip", False):
            tpg = self.stpg
        if seq:             # one slot per sequence and domain (amid_sas_seq_bwd_f32)
            ln1, ln2, tpg = self.ln1_part_s, self.ln2_part_s, B
        for l in (0, 1):
            for g in (0, 1):
                pre = f"sac{g + 1}"
                wbase = lambda w: ((g * 6 + w) * S) * D * D      # noqa: E731
                bbase = lambda w: ((g * 6 + w) * S) * D          # noqa: E731
                for j in range(3):
                    add(self.w_part[l], wbase(j), fp.ptr(f"{pre}.attention_layers.{l}.in_proj_weight", G, j * D * D), D * D, S, D * D)
                    add(self.b_part[l], bbase(j), fp.ptr(f"{pre}.attention_layers.{l}.in_proj_bias", G, j * D), D, S, D)
                add(self.w_part[l], wbase(3), fp.ptr(f"{pre}.attention_layers.{l}.out_proj.weight", G), D * D, S, D * D)
                add(self.b_part[l], bbase(3), fp.ptr(f"{pre}.attention_layers.{l}.out_proj.bias", G), D, S, D)
                add(self.w_part[l], wbase(4), fp.ptr(f"{pre}.forward_layers.{l}.conv1.weight", G), D * D, S, D * D)
                add(self.b_part[l], bbase(4), fp.ptr(f"{pre}.forward_layers.{l}.conv1.bias", G), D, S, D)
                add(self.w_part[l], wbase(5), fp.ptr(f"{pre}.forward_layers.{l}.conv2.weight", G), D * D, S, D * D)
                add(self.b_part[l], bbase(5), fp.ptr(f"{pre}.forward_layers.{l}.conv2.bias", G), D, S, D)
                tb = g * tpg * 2 * D
                add(ln1[l], tb, fp.ptr(f"{pre}.attention_layernorms.{l}.weight", G), 2 * D, tpg, D)
                add(ln1[l], tb + D, fp.ptr(f"{pre}.attention_layernorms.{l}.bias", G), 2 * D, tpg, D)
                add(ln2[l], tb, fp.ptr(f"{pre}.forward_layernorms.{l}.weight", G), 2 * D, tpg, D)
                add(ln2[l], tb + D, fp.ptr(f"{pre}.forward_layernorms.{l}.bias", G), 2 * D, tpg, D)
        for g in (0, 1):
            pre = f"sac{g + 1}"
            add(self.last_part, g * B * 2 * D, fp.ptr(f"{pre}.last_layernorm.weight", G), 2 * D, B, D)
            add(self.last_part, g * B * 2 * D + D, fp.ptr(f"{pre}.last_layernorm.bias", G), 2 * D, B, D)
        T = self.shape.Tenc
        for g in (0, 1):
            if pos:
                add(self.dpos_part, g * T * D, fp.ptr(f"sac{g + 1}.pos_emb.weight", G), 2 * T * D, self.pos_splits, T * D)
