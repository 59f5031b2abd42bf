"""Host-side driver of the SASRec training hot path on one MI355X.

Mirrors the loop body of the reference's ``train()`` (train_sr.py:190-217) for
``SASRec`` (model_seq.py:390-443): forward, masked BCE, backward, optimizer --
as one sequence of launches of the hand-written HIP kernels in
``libamid_hip.so`` on a single HIP stream, capturable into a hipGraph.
PyTorch only provides device memory and the stream.

Memory layout (all fp32, resident in HBM for the life of the engine):
  table / m / v      [n_rows, D]   item embedding + lazy-Adam moments; ``last`` [n_rows] int32
  dense              flat buffer holding every other parameter in state_dict order (16-B aligned
                     slots); grad / Adam m / Adam v mirror it, so dense Adam and the data-parallel
                     all-reduce are one call each
  plan workspace     activations saved for backward, gradient scratch, sort / segment-reduce
                     workspaces -- allocated once per (B, T, n_items) shape
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import lib, ptr_array

from .engine_dp import DataParallelMixin, HipMergeBackend      # noqa: F401  (re-exported)
from .engine_graph import GraphMixin
from .engine_io import InputMixin
from .plan import (DR_HEADS, SASREC_HEADS, SASREC_LN_EPS, SASREC_P_DROP, FlatParams, SasrecPlan, Shape,      # noqa: F401  (re-exported)
                   sasrec_dense_names)


class SasrecEngine(InputMixin, GraphMixin, DataParallelMixin):
    """Parameters, optimizer state and launch sequences for SASRec (isInC = isItC = isDR = False)."""

    HEADS = SASREC_HEADS
    PLAN_CLS = SasrecPlan
    EMB_DIMS = (64, 128)
    # ---- path switches (class attributes so that tests and the A/B tools can force a path; production never touches them) -----------
    #   STRIP_KERNELS / BF16_STRIP   the encoder on the register-resident strip kernels (fp32 / compute = "bf16")          [here]
    #   SEQ_FORWARD, LIVE_FORWARD    the forward as ONE launch; over the live sequences only in a train step            [at _live_list]
    #   SEQ_BACKWARD                 the backward's data gradients as ONE launch ("auto": n_CU < B <= 2 n_CU)          [at _seq_backward]
    #   FUSED_HEAD                   head forward + backward as one launch                                              [at _enqueue_fwd_bwd]
    #   SORT_RIDERS                  the step's index sort as extra workgroups of five main-stream launches             [at enqueue_sort]
    #   COMPACT_LIVE, COMPACT_MIN_IDX  the sparse side on the live sequences' positions only, for long index lists      [at live_forward_ok]
    #   DENSE_EXCHANGE               how the dense gradient crosses the ranks (engine_dp.DataParallelMixin)
    SHORT_TILE_BUILDS = False    # (BERT4Rec's row-tile kernels also exist as *_rt3 / *_rt4 / *_rt5: 48- / 64- / 80-row tiles, csrc/Makefile)
    STRIP_KERNELS = True         # fp32: the layer's GEMM chains run as register-resident strip kernels (csrc/sasrec_strip.hip)
    BF16_STRIP = True            # compute = "bf16" on the strip path (the forward's products in bf16)
    # (No path switch is read from the environment: A/B tools and tests set these class attributes -- bench.py --set NAME=VALUE,
    # profiles/tools/*.py -- and diagnostic BUILDS of the library are selected with AMID_LIB_PATH, amid_amd/_lib.py.)
    SEQ_FWD_VARIANT = 0          # != 0: amid_sas_seq_fwd_variant (which build of the one-launch forward runs: diagnostics)
    SEQ_BWD_VARIANT = 0          # != 0: amid_sas_seq_bwd_variant

    def _dense_names(self) -> List[Tuple[str, Tuple[int, ...]]]:
        return sasrec_dense_names(self.Tpos, self.D, self.hid, self.itc_bs, self.dr, self.inc_bs)

    def _alloc_model_buffers(self) -> None:
        D = self.D
        # transposed copies of the 24 square projection weights
        self.wT = torch.zeros(2, 2, 6, D * D, dtype=torch.float32, device=self.device)     # [layer][domain][q,k,v,o,c1,c2]

    def __init__(self, item_length: int, emb_dim: int, seq_len: int, hid_dim: int, device="cuda:0", lr: float = 5e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8, seed: int = 0, itc_bs: int = 0, itc_threshold: float = 0.5, dr: bool = False,
                 dr_e_w: float = 0.1, compute: str = "f32", inc_bs: int = 0, inc_threshold: float = 0.5):
        """itc_bs > 0: SASRec(isItC=True, bs=itc_bs, threshold2=itc_threshold) -- InterComp after the encoders
        (model_seq.py:426-431); every batch must then hold exactly itc_bs rows (trans_bs is Linear(bs, 1) over the batch)."""
        L = lib()        # raises AmidLibraryError when the HIP library is missing: no fallback
        if self.SEQ_FWD_VARIANT:          # A/B measurements of the fused forward's builds (amid_sas_seq_fwd_variant)
            L.value("amid_sas_seq_fwd_variant", int(self.SEQ_FWD_VARIANT))
        if self.SEQ_BWD_VARIANT:          # ... and of the fused backward's (amid_sas_seq_bwd_variant)
            L.value("amid_sas_seq_bwd_variant", int(self.SEQ_BWD_VARIANT))
        self.itc_bs, self.itc_threshold = int(itc_bs), float(itc_threshold)
        # inc_bs > 0: SASRec(isInC=True, bs=inc_bs, threshold1=inc_threshold) -- InnerComp on the gathered rows before the encoders,
        # which then see 2 * seq_len tokens per row (model_seq.py:398-401, :422-424; csrc/innercomp.hip); batches of exactly inc_bs rows
        self.inc_bs, self.inc_threshold = int(inc_bs), float(inc_threshold)
        # dr: SASRec(isDR=True) -- predict_ips / predict_gfunc heads and the two objectives of train_sr_dr.py; dr_mode selects
        # the objective of the next train step (0: loss_cls + dr_e_w * loss_dr_e, 1: loss_dr_r), see select_optimizer()
        self.dr, self.dr_e_w, self.dr_mode = bool(dr), float(dr_e_w), 0
        # compute = "bf16": the dense projections' matrix products take bf16 operands (fp32 accumulate, fp32 everything else);
        # BASELINE.json configs[2].  "f32" (default): exact fp32 products.
        if compute not in ("f32", "bf16"):
            raise ValueError(f"compute must be 'f32' or 'bf16', got {compute!r}")
        if compute == "bf16" and emb_dim != 128:
            raise ValueError("the bf16 matrix-core kernels are built for emb_dim 128")
        self.compute, self.mma_bf16 = compute, 1 if compute == "bf16" else 0
        if emb_dim not in self.EMB_DIMS:
            raise ValueError(f"amid_amd {type(self).__name__} kernels are built for emb_dim in {self.EMB_DIMS}, got {emb_dim}")
        self.device = torch.device(device)
        # every kernel of the engine runs on this (non-default, hence capturable) HIP stream
        self.stream = torch.cuda.Stream(device=self.device)
        # the index sort only feeds the segment reduce after backward: it runs on this side stream, beside the forward pass
        self.side = torch.cuda.Stream(device=self.device)
        self.ev_idx = torch.cuda.Event()
        self.ev_sorted = torch.cuda.Event()
        self.n_rows, self.D, self.T, self.hid, self.H = int(item_length), int(emb_dim), int(seq_len), int(hid_dim), self.HEADS
        self.Tpos = 2 * self.T if getattr(self, "inc_bs", 0) else self.T          # rows of the pos_emb tables
        D = self.D
        self.dense = FlatParams(self._dense_names(), self.device)
        self.table = torch.zeros(self.n_rows, D, dtype=torch.float32, device=self.device)
        self.table_m: Optional[torch.Tensor] = None        # allocated on the first optimizer use
        self.table_v: Optional[torch.Tensor] = None
        self.table_last: Optional[torch.Tensor] = None
        self._alloc_model_buffers()
        self.hyper = dict(lr=lr, beta1=betas[0], beta2=betas[1], eps=eps)
        self.seed = int(seed)
        self.step = 0
        self._st_bytes = L.value("amid_step_state_bytes")
        # empty, not zeros: a zero fill on torch's stream is unordered against the copy _push_step_state issues on the engine's stream and
        # could land after it (a step then ran with seed 0 and zero Adam coefficients: found by tests/test_gpu_fused_opt.py, round 6)
        self.step_state = torch.empty(self._st_bytes, dtype=torch.uint8, device=self.device)
        self._push_step_state()
        self.plans: Dict[Tuple[int, int, int, bool], SasrecPlan] = {}
        self.grad_scale = 1.0
        self.opt_bank, self._banks = 0, {}          # several Adam states over the same parameters (train_sr_dr.py:668-669)
        torch.cuda.synchronize(self.device)
        self._ptr_cache: Dict[str, object] = {}

    @property
    def s(self) -> int:
        return self.stream.cuda_stream

    def sync(self) -> None:
        self.join_sort()
        self.stream.synchronize()

    # ------------------------------------------------------------------ state
    def _push_step_state(self) -> None:
        L = lib()
        host = (ctypes.c_ubyte * self._st_bytes)()
        L.call("amid_step_state_pack", ctypes.addressof(host), self.seed, self.step, self.hyper["lr"], self.hyper["beta1"],
               self.hyper["beta2"], self.hyper["eps"])
        with torch.cuda.stream(self.stream):
            self.step_state.copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8), non_blocking=False)

    @staticmethod
    def rank_seed(seed: int, rank: int) -> int:
        """Dropout seed of data-parallel rank `rank`: the counters are indexed by the LOCAL row, so ranks sharing one seed would apply
        the same masks to their different shards -- correlated noise a single-GPU run at the global batch does not have.  Weights
        are initialised from `seed` itself on every rank (identical replicas)."""
        return (int(seed) + 0x9E3779B97F4A7C15 * int(rank)) & 0x7FFFFFFFFFFFFFFF

    def set_step(self, step: int, seed: Optional[int] = None) -> None:
        self.step = int(step)
        if seed is not None:
            self.seed = int(seed)
        self._push_step_state()

    def set_lr(self, lr: float) -> None:
        self.hyper["lr"] = float(lr)
        self._push_step_state()

    def _ensure_opt_state(self) -> None:
        if self.table_m is None:
            self.table_m = torch.zeros_like(self.table)
            self.table_v = torch.zeros_like(self.table)
            self.table_last = torch.zeros(self.n_rows, dtype=torch.int32, device=self.device)
            # the zero fills run on torch's stream, the step that needs them on the engine's: without this a first step enqueued straight away
            # read moments and stamps that were not zero yet (a catch-up replaying garbage gaps: found by tests/test_gpu_fused_opt.py, round 6)
            torch.cuda.synchronize(self.device)

    def select_optimizer(self, k: int, lr: Optional[float] = None) -> None:
        """Switch to Adam state `k` (its own moments, step counter and learning rate over the SAME parameters): the reference's
        doubly-robust trainer alternates two torch.optim.Adam instances (train_sr_dr.py:668-669, :222-224, :396-398).  Pending
        lazy updates of the state being left are applied first (one pass over the table), so rows never owe moves to an idle state."""
        if k == self.opt_bank:
            if lr is not None and lr != self.hyper["lr"]:
                self.set_lr(lr)
            return
        self.flush_table()
        self.sync()
        fp = self.dense
        self._banks[self.opt_bank] = dict(m=fp.m, v=fp.v, tm=self.table_m, tv=self.table_v, tl=self.table_last, st=self.step_state,
                                          step=self.step, lr=self.hyper["lr"], seed=self.seed)
        b = self._banks.get(k)
        if b is None:
            b = dict(m=torch.zeros_like(fp.m), v=torch.zeros_like(fp.v), tm=None, tv=None, tl=None,
                     st=torch.empty(self._st_bytes, dtype=torch.uint8, device=self.device), step=0,
                     lr=self.hyper["lr"] if lr is None else float(lr), seed=self.seed + 0x9E3779B9 * k)
        fp.m, fp.v, self.table_m, self.table_v, self.table_last = b["m"], b["v"], b["tm"], b["tv"], b["tl"]
        self.step_state, self.step, self.seed = b["st"], b["step"], b["seed"]
        self.hyper["lr"] = b["lr"] if lr is None else float(lr)
        self.opt_bank = k
        self._push_step_state()
        torch.cuda.synchronize(self.device)

    def _graph_key(self) -> Tuple[int, int]:
        return (self.opt_bank, self.dr_mode if self.dr else 0)

    def plan(self, B: int, T: int, NI: int, need_grad: bool) -> SasrecPlan:
        key = (B, T, NI, need_grad)
        if key not in self.plans:
            inc = getattr(self, "inc_bs", 0)
            if (2 * T if inc else T) > self.Tpos:
                raise ValueError(f"sequence length {T} exceeds pos_emb size {self.Tpos}" + (" / 2 (isInC)" if inc else ""))
            self.plans[key] = self.PLAN_CLS(self, Shape(B, T, NI, 2 * T if inc else 0), need_grad)
            torch.cuda.synchronize(self.device)      # buffers were zero-filled on torch's stream
        return self.plans[key]

    # ------------------------------------------------------------------ pointer helpers
    def _pp(self, fmt: str, buf: Optional[torch.Tensor] = None, extra: int = 0):
        """Host array (domain 0, domain 1) of device pointers of a per-domain parameter."""
        key = (fmt, 0 if buf is None else buf.data_ptr(), extra)
        c = self._ptr_cache.get(key)
        if c is None:
            c = ptr_array([self.dense.ptr(fmt.format(d=d), buf, extra) for d in (1, 2)])
            self._ptr_cache[key] = c
        return c

    def _wT(self, layer: int, which: int):
        """Host array (domain 0, domain 1) of the transposed projection weight `which` (q, k, v, o, c1, c2) of `layer`: fp32 transposes,
        or -- pl.strip with compute = "bf16" -- their bf16 fragment images (amid_sas_weights_bf16, refreshed by enqueue_backward)."""
        bf = self._bf16_bwd
        p3 = self._p3_bwd
        key = ("wT16x3" if p3 else "wT16" if bf else "wT", layer, which)
        c = self._ptr_cache.get(key)
        if c is None:
            buf = self.wT16x3 if p3 else self.wT16 if bf else self.wT
            c = ptr_array([buf[layer, g, which].data_ptr() for g in (0, 1)])
            self._ptr_cache[key] = c
        return c

    _bf16_bwd = False          # set per backward: the strip backward's products take bf16 images (compute = "bf16" on the strip path)
    _p3_bwd = False            # set per backward: ... three-plane images (compute = "fp32", products on bf16 pieces)
    # compute = "fp32": the strip backward's data-gradient products on the bf16 matrix cores at fp32 accuracy (three bf16 pieces per operand,
    # six piece pairs: csrc/strip_gemm.h strip_mma16x6, strip_chain.h RingP3), from three-plane images of the TRANSPOSED weights; "0": fp32
    # matrix instructions
    BWD_SPLIT = True

    # ------------------------------------------------------------------ launch sequences
    def enqueue_prepare(self, pl: SasrecPlan, sparse: bool, bump_step: bool = False, defer_sort: bool = False) -> None:
        """defer_sort: only mark the fork point; the caller launches the sort with enqueue_sort() AFTER the main stream's next
        kernel.  (In the captured graph the branch that is enqueued first is dispatched first: with the sort ahead of the catch-up
        kernel the latter started ~20 us late in every replay.)"""
        L, s, shp = lib(), self.s, pl.shape
        # a train step (bump_step) of a loss that masks the other domain of every sample: the batch's live-sequence list
        # (amid_live_list_i32) rides in the packing launch; enqueue_forward then finds it in place (pl.live_packed)
        with_live = bool(bump_step and getattr(pl, "strip", False) and not self.itc_bs)
        pl.live_packed = with_live
        # ... and then sort, segment reduce and row Adam run on the live sequences' positions only (the dead sequences' gradient rows
        # are exact zeros; a row only they hold is not touched this step -- the lazy Adam replays it when it is next read).  InnerComp
        # couples the rows of a batch in front of the encoders: every position keeps its gradient there.
        pl.compact = bool(with_live and sparse and self.compact_ok(pl))
        # ... and the step's index sort rides in main-stream launches (catch-up + the three strip backward launches carry one phase
        # each as extra workgroups: no side stream, no fork, no join) when the keys fit the riders' 1024-bin build
        pl.riding = bool(bump_step and sparse and defer_sort and self.SORT_RIDERS and getattr(pl, "strip", False) and not pl.compact
                         and not self._seq_backward(pl)             # (the fused backward is one launch: three of the riders' hosts are gone)
                         and (shp.n_idx + 2047) // 2048 <= 2 * shp.Tenc and self._sort_plan(pl) is not None)
        # ... or, for the steps without five launches to ride in (the one-launch backward at T <= 32, BERT4Rec, the comp modules): the WHOLE sort
        # chained inside the catch-up launch (amid_lazy_adam_catchup_positions_sort_f32 phase 6) -- no side stream, no fork, no join
        pl.chain = bool(bump_step and sparse and defer_sort and self.SORT_CHAIN and not pl.riding and not pl.compact
                        and shp.n_idx <= lib().value("amid_sort_chain_max_indices") and self.n_rows <= (1 << 20) and self._sort_plan(pl) is not None)
        ent = self.input_pool(pl)
        # compute = "bf16": the step folds like the fp32 one when its shape does (the decisions below then see "f32": _ceff)
        self._bf16_as_f32 = False
        if self.compute == "bf16" and self.BF16_FOLD and bump_step and shp.B <= self.BF16_FOLD_MAX_B:
            self._bf16_as_f32 = True
            if not (ent is not None and sparse and defer_sort and with_live and self._tail2_ok(pl)):
                self._bf16_as_f32 = False
            else:
                pl.riding = False            # (decided above with the mode's own launches in mind: the folded step rides on the compact list's plan)
        # the live-sequence step on an input pool in twelve launches (FUSED_TAIL): packing, catch-up and phase 1 of the sort of the COMPACT
        # index list are one launch; the later phases ride in the backward strips and the weight gradients; no embedding-backward launch
        pl.tail2 = bool(ent is not None and bump_step and sparse and defer_sort and with_live and self._tail2_ok(pl))
        pl.gather_on_fwd = False
        if pl.tail2:
            pool, phase = ent
            self._ensure_opt_state()
            head = (pool.data_ptr(), pool.stride(0), pool.shape[0], phase, pl.in_pack.data_ptr(), pl.in_words, shp.B, shp.T,
                    shp.NI - 1, self.n_rows, pl.idx_all.data_ptr(), pl.idx_c.data_ptr(), pl.row_c.data_ptr(), pl.live.data_ptr(), pl.err.data_ptr(),
                    self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), self.D,
                    self.step_state.data_ptr(), self._sort_plan_c(pl))
            # the step's gather as the prologue of its forward (amid_sas_seq_fwd_gather_*_f32): K1's riders -- the weight images of the forward
            # and of the backward strips -- move to this launch's last workgroups
            pl.gather_on_fwd = bool(self.GATHER_ON_FWD and self._fwd_on_pieces(pl, shp.B, shp.Tenc) and self._p3_bwd_for(pl) and shp.T == shp.Tenc)
            if pl.gather_on_fwd:
                src, w16 = self._w16_images(3)
                L.call("amid_step_head_w16_f32", *head, src, 24, 3, w16.data_ptr(), self._wT16x3_buf().data_ptr(), s)
            else:
                L.call("amid_step_head_f32", *head, s)
            self.step += 1
            pl.compact, pl.riding, pl.chain = True, True, False
            self._sort_owed = False
            return
        if ent is not None:
            pool, phase = ent
            if not bump_step:
                raise ValueError("an input pool advances with the step counter: enqueue_prepare(bump_step=True) only")
            if with_live:
                L.call("amid_pack_indices_pool_live", pool.data_ptr(), pool.stride(0), pool.shape[0], phase, pl.in_pack.data_ptr(),
                       pl.in_words, shp.B, shp.T, shp.NI - 1, self.n_rows, pl.idx_all.data_ptr(), pl.err.data_ptr(),
                       self.step_state.data_ptr(), pl.live.data_ptr(), s)
            else:
                L.call("amid_pack_indices_pool", pool.data_ptr(), pool.stride(0), pool.shape[0], phase, pl.in_pack.data_ptr(),
                       pl.in_words, shp.B, shp.T, shp.NI - 1, self.n_rows, pl.idx_all.data_ptr(), pl.err.data_ptr(),
                       self.step_state.data_ptr(), s)
            self.step += 1
            if sparse and (pl.riding or pl.chain):
                self._sort_owed = False
            elif sparse:
                self.ev_idx.record(self.stream)
                self._sort_owed = bool(defer_sort)
                if not defer_sort:
                    self.enqueue_sort(pl)
            return
        if with_live:
            L.call("amid_pack_indices_live", pl.in_i_node.data_ptr(), pl.in_neg.data_ptr(), pl.in_seq_d1.data_ptr(), pl.in_seq_d2.data_ptr(),
                   shp.B, shp.T, shp.NI - 1, self.n_rows, pl.idx_all.data_ptr(), pl.err.data_ptr(),
                   self.step_state.data_ptr() if bump_step else None, pl.domain.data_ptr(), pl.live.data_ptr(), s)
        else:
            L.call("amid_pack_indices", pl.in_i_node.data_ptr(), pl.in_neg.data_ptr(), pl.in_seq_d1.data_ptr(), pl.in_seq_d2.data_ptr(),
                   shp.B, shp.T, shp.NI - 1, self.n_rows, pl.idx_all.data_ptr(), pl.err.data_ptr(),
                   self.step_state.data_ptr() if bump_step else None, s)
        if bump_step:
            self.step += 1
        if sparse and (pl.riding or pl.chain):
            self._sort_owed = False
        elif sparse:
            self.ev_idx.record(self.stream)       # fork point: the index list is complete
            self._sort_owed = bool(defer_sort)
            if not defer_sort:
                self.enqueue_sort(pl)

    def enqueue_sort(self, pl: SasrecPlan) -> None:
        """Sort / unique on the side stream (joined by the gradient tail just before the segment reduce)."""
        L, shp = lib(), pl.shape
        self.side.wait_event(self.ev_idx)
        if getattr(pl, "compact", False):
            L.call("amid_sort_unique_rows_i32", pl.idx_c.data_ptr(), pl.row_c.data_ptr(), pl.n_compact, self.n_rows, pl.sort_ws.data_ptr(),
                   pl.pos_sorted.data_ptr(), pl.uniq_ids.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), pl.n_uniq.data_ptr(),
                   self.side.cuda_stream)
        else:
            L.call("amid_sort_unique_i32", pl.idx_all.data_ptr(), shp.n_idx, self.n_rows, pl.sort_ws.data_ptr(), pl.pos_sorted.data_ptr(),
                   pl.uniq_ids.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), pl.n_uniq.data_ptr(), self.side.cuda_stream)
        self.ev_sorted.record(self.side)
        self._sort_pending = True
        self._sort_owed = False

    def _wgrad_mode(self, D: int) -> int:
        """The mma mode of amid_sas_wgrad(_rows)_f32: 1 = operands rounded to bf16 (compute "bf16"), 2 / 3 = fp32 operands as three bf16
        pieces (nine / six piece pairs), 0 = fp32 matrix instructions."""
        if D != 128:
            return 0
        if self.BF16_WGRAD and getattr(self, "_bf16_bwd", False):
            return 1
        if self._bf16_as_f32 and self.BF16_WGRAD and self.BF16_FWD_ONE and getattr(self, "_wgrad_one", False):
            return 4                    # (the folded bf16 step's weight-gradient launch: ONE piece per operand -- asked for at the launch site only)
        return {"9": 2, "6": 3}.get(self.WGRAD_SPLIT, 0)

    def join_sort(self) -> None:
        """Make the main stream wait for the side-stream sort (no-op if nothing is pending)."""
        if getattr(self, "_sort_pending", False):
            self.stream.wait_event(self.ev_sorted)
            self._sort_pending = False

    SORT_RIDERS = True
    # compute = "bf16" on a step that folds (round 6): the fp32 step's nine launches -- its forward on bf16 pieces included, which is MORE than the mode
    # asks for -- with the backward strips' data-gradient products on ONE bf16 piece (they read the hi plane of the three-plane images).  Decided per
    # step in enqueue_prepare (_bf16_as_f32); BF16_FOLD_MAX_B: batches beyond it keep the unfolded bf16 launches.
    BF16_FOLD = True
    BF16_FOLD_MAX_B = 1 << 30          # (with the forward on one piece too the fold wins at B 512 as well: cfg 3 bf16 0.4915 -> 0.4751)
    BF16_FWD_ONE = True          # ... and its forward multiplies ONE piece per operand too (amid_sas_seq_fwd_gather_*_p1_f32): bf16 products
    _bf16_as_f32 = False

    def _ceff(self) -> str:
        """The compute mode the step's path decisions see: "f32" for a compute = "bf16" step that takes the folded launches."""
        return "f32" if self._bf16_as_f32 else self.compute
    FOLD_SHORT = True            # 16 < T <= 32 on an input pool: the folded strips step instead of the one-launch backward (round 6)
    # the whole sort chained inside the catch-up launch where the riders have no five launches (round 6).  Off by default: the chain takes ~ 45 us
    # (five phases, four barriers with agent-scope fences), which the catch-up hides only when it replays long gaps -- cfg 4's real epoch in its
    # steady state 0.2383 -> 0.2331 ms, but + 17 % on a T 20 step without lagging rows and + 5 % on BERT4Rec (DESIGN_HISTORY.md, "Round 6")
    SORT_CHAIN = False
    # compute = "bf16": the weight gradients' products on the bf16 matrix cores too (amid_sas_wgrad_rows_f32 mma_bf16); 0: fp32 products
    BF16_WGRAD = True
    # compute = "fp32": the weight gradients' products on the bf16 matrix cores at fp32 accuracy -- every operand element as three bf16
    # pieces, six ("6", the default) or nine ("9") piece pairs (csrc/sasrec_bwd.hip sas_wgrad_split_kernel; D = 128); "0": fp32 matrix
    # instructions.  Error against the fp64 product 3.8e-7 of the largest entry either way, 4.4e-7 for the fp32 instructions.
    WGRAD_SPLIT = "6"
    # compute = "fp32": the one-launch forward's twelve projections on the bf16 matrix cores at fp32 accuracy too (three bf16 pieces per
    # operand, six piece pairs; the pieces are made once, by the wave that produced the operand, and cross the strip's column parts as
    # operand fragments: amid_sas_seq_fwd_split_f32, csrc/sasrec_seqn.hip seqn_fwd_px_kernel; weight images by the gather's extra
    # workgroups).  Every saved tensor within 3e-6 of the fp32 build's; 153 k cycles against 187 k, step 0.3562 -> 0.3510 ms at cfg 2 (the
    # bf16-dense kernel runs at a lower clock: 15 % fewer cycles, 7 % less time).  "0": fp32 matrix instructions.
    FWD_SPLIT = True
    # The train step's encoder backward (data gradients) as ONE launch over the live sequences where csrc/sasrec_strip.hip covers the
    # shape (amid_sas_seq_bwd_f32): "auto" = where it wins.  It tiles one sequence per workgroup, a whole CU each, so the step's sort
    # riders find no free CU in it and the sort goes back to the side stream (a fork and a join, ~10 us of a replayed graph).  Measured
    # (MI355X, T 50): B 256 = one round of workgroups either way, a tie (0.391 against 0.389 ms with the riders); B 512 = two full
    # rounds against 400 row tiles in two uneven ones, 0.690 against 0.740 ms; B 4096 = 16 rounds against 12.5 rounds of denser row
    # tiles, 4.75 against 4.48 ms.  So: more live sequences than CUs, at most two rounds.  T <= 32 (csrc/sasrec_seqn_bwd.hip: two strips x
    # four column parts per sequence): B 256, T 20 runs 80 strip tiles per launch on 256 CUs -- 111 us in five launches against 67.8 in one,
    # 0.267 -> 0.243 ms per step with the sort back on its side stream; taken for B <= n_CU.  SEQ_BACKWARD = "1" / "0" forces it on / off.
    SEQ_BACKWARD = "auto"

    def _seq_backward(self, pl: SasrecPlan) -> bool:
        if not getattr(pl, "seq_bwd", False) or self.itc_bs or self.SEQ_BACKWARD in ("0", False):
            return False
        if self.SEQ_BACKWARD in ("1", True):
            return True
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        if pl.shape.Tenc <= 32:         # short sequences: the strip launches fill a third of the chip, a workgroup per sequence all of it
            # ... unless the step folds (round 6; 16 < T: the head rides on the forward's tail): the strips host the riders, the gather and the
            # embedding backward ride too and no side stream forks -- cfg 4's real epoch 238.4 -> 229 us per step (profiles/tools/trace_cfg4.sh)
            if self.FOLD_SHORT and pl.shape.Tenc > 16 and self.D == 128 and self._folded_step_shape(pl):
                return False
            return pl.shape.B <= n_cu
        if self.D != 128:               # D 64 at T 50: the five strip launches win (B 512: 0.352 against 0.393 ms; B 256: a tie)
            return False
        if self._folded_step_shape(pl):      # the strips host the folded step's riders (round 5; cfg 3, B 512: 0.6006 against 0.6144 ms
            return False                    # for the one-launch backward in front of the fifteen-launch tail)
        return n_cu < pl.shape.B <= 2 * n_cu

    def _sort_plan(self, pl: SasrecPlan):
        """Host address of the plan of the step's index sort (amid_sort_plan_pack), or None when the riders do not cover it."""
        if not hasattr(pl, "_sort_plan_buf"):
            L = lib()
            buf = (ctypes.c_ubyte * L.value("amid_sort_plan_bytes"))()
            rc = L._fn["amid_sort_plan_pack"](ctypes.addressof(buf), pl.idx_all.data_ptr(), None, pl.shape.n_idx, self.n_rows,
                                              pl.sort_ws.data_ptr(), pl.pos_sorted.data_ptr(), pl.uniq_ids.data_ptr(), pl.seg_off.data_ptr(),
                                              pl.seg_of.data_ptr(), pl.n_uniq.data_ptr())
            pl._sort_plan_buf = buf if rc == 0 else None
        return ctypes.addressof(pl._sort_plan_buf) if pl._sort_plan_buf is not None else None

    def _sort_plan_c(self, pl: SasrecPlan):
        """Host address of the sort plan over the plan's COMPACT list (idx_c, row_c), or None when the riders do not cover it."""
        if not hasattr(pl, "_sort_plan_c_buf"):
            L = lib()
            buf = (ctypes.c_ubyte * L.value("amid_sort_plan_bytes"))()
            rc = L._fn["amid_sort_plan_pack"](ctypes.addressof(buf), pl.idx_c.data_ptr(), pl.row_c.data_ptr(), pl.n_compact, self.n_rows,
                                              pl.sort_ws.data_ptr(), pl.pos_sorted.data_ptr(), pl.uniq_ids.data_ptr(), pl.seg_off.data_ptr(),
                                              pl.seg_of.data_ptr(), pl.n_uniq.data_ptr())
            pl._sort_plan_c_buf = buf if rc == 0 else None
        return ctypes.addressof(pl._sort_plan_c_buf) if pl._sort_plan_c_buf is not None else None

    # The live-sequence train step on an input pool with its head and tail folded (round 5): amid_step_head_f32 (packing + catch-up + sort
    # phase 1 over the compact list), the embedding backward on the last strip launch (amid_sas_strip_qkv_bwd_emb_f32), the sort's last
    # phase in the weight-gradient launch, the position rows' gradients in the gradient tail (amid_grad_tail_live_f32) and the segment
    # reduce's second phase in the optimizer launch (amid_optimizer_step_spans_f32): 12 launches instead of 15.  False: round 4's sequence
    # (tests compare the two).
    FUSED_OPT = True             # the folded step's optimizer inside its gradient tail: ten launches (round 6; False: eleven)
    GATHER_ON_FWD = True         # ... and its gather K1 as the prologue of the forward's workgroups: nine (the weight images by the step head's riders)
    FUSED_SPANS = True           # every single-GPU train step: the segment reduce's second phase inside the optimizer launch (round 5)
    FUSED_TAIL = True

    def _coll(self, fn) -> None:
        """A collective in the MIDDLE of a step (InterComp's / InnerComp's gathers and reductions under data parallel).  Eagerly: run it.
        While engine_dp captures the step in segments (_seg_capture: a list), a collective cannot sit inside a captured graph: the capture
        ends here, the graph so far and the collective are recorded in order, and a new capture begins behind it -- the replay launches
        graph | collective | graph | ... (engine_dp.train_step_dp)."""
        seg = getattr(self, "_seg_capture", None)
        if seg is None:
            fn()
            return
        L = lib()
        if getattr(self, "_sort_owed", False):      # a deferred side-stream sort whose fork point has not come yet: start it now ...
            self.ev_idx.record(self.stream)
            self.enqueue_sort(self._seg_plan)
        self.join_sort()                            # ... and join it: a capture cannot end with forked work (the segment then holds the whole sort)
        out = ctypes.c_void_p()
        L.call("amid_graph_capture_end", self.s, ctypes.byref(out))
        try:
            seg.append(("graph", out.value))
            seg.append(("call", fn))
        finally:      # the caller's `finally` ends a capture: the stream is capturing again whatever happened in between
            L.call("amid_graph_capture_begin", self.s)

    # The data-parallel step's local half (enqueue_local_grads: graph A of the graph pair, the local-gradients graph, the eager step) in the
    # folded form too (round 6): step head, head on the forward's tail, embedding backward on the last strip, position rows in the tail; phase B
    # of the segment reduce stays a launch (with the chunk's packing riding in it) because the exchange ships the finished rows.
    FUSED_TAIL_DP = True

    def _fold_ctx(self) -> bool:
        """Whether the launches being enqueued belong to a step that may take the folded form."""
        return bool(getattr(self, "_in_train_step", False) or (getattr(self, "_in_local_grads", False) and self.FUSED_TAIL_DP))

    def _folded_step_shape(self, pl: SasrecPlan) -> bool:
        """The conditions of the folded twelve-launch step that do not depend on how the backward runs (_tail2_ok adds those)."""
        return bool(self.FUSED_TAIL and self._fold_ctx() and self.SORT_RIDERS and pl.need_grad and self.D == 128
                    and self._ceff() == "f32" and not self.dr and not self.itc_bs and not self.inc_bs and not getattr(self, "comp", "")
                    and pl.shape.NI > 1 and self.FUSED_HEAD and self.BWD_SPLIT and pl.strip
                    and self.input_pool(pl) is not None and self.live_forward_ok(pl) and self._wgrad_mode(self.D) == 3
                    and not self._fold_catchup(pl) and self._sort_plan_c(pl) is not None and (pl.n_compact + 2047) // 2048 <= 12 * pl.splits)

    def _tail2_ok(self, pl: SasrecPlan) -> bool:
        shp = pl.shape
        # (inside enqueue_train_step the segment reduce's runs across chunks are finished by THIS step's optimizer launch; inside
        # enqueue_local_grads -- what a data-parallel exchange ships -- by a launch of their own behind the tail, amid_grad_tail_live_dp_f32)
        return bool(self.FUSED_TAIL and self._fold_ctx() and self.SORT_RIDERS and pl.need_grad and self.D == 128 and self._ceff() == "f32" and not self.dr
                    and not self.itc_bs and not self.inc_bs and not getattr(self, "comp", "") and pl.shape.NI > 1
                    and self.FUSED_HEAD and self.live_forward_ok(pl) and self._p3_bwd_for(pl) and self._wgrad_mode(self.D) == 3
                    and not self._fold_catchup(pl) and self._sort_plan_c(pl) is not None
                    and (pl.n_compact + 2047) // 2048 <= 12 * pl.splits)

    def _tail2_table(self, pl: SasrecPlan):
        """The gradient tail's reduce table without the position rows' partial sums and without the scorer's per-sample partials
        (amid_grad_tail_live_f32 sums the former from the rows and forms the latter from the head's per-sample hidden gradients)."""
        if not hasattr(pl, "red_entries_t"):
            pl.red_entries_t, pl.red_n_t, pl.red_max_t = pl._build_reduce_table(self, live=True, pos=False)
        return pl.red_entries_t, pl.red_n_t, pl.red_blk_t

    def _ln_stat(self, pl: SasrecPlan):
        """(tensors, host pointer array) of the per-layer row statistics [2 M][4] a forward stores instead of qn / y."""
        if not hasattr(pl, "ln_stat"):
            pl.ln_stat = [torch.zeros(2 * pl.shape.M, 4, dtype=torch.float32, device=self.device) for _ in range(2)]
            pl._ln_stat_ptrs = ptr_array([t.data_ptr() for t in pl.ln_stat])
            torch.cuda.synchronize(self.device)      # zero-filled on torch's stream, used on the engine's
        return pl.ln_stat, pl._ln_stat_ptrs

    def _hidg(self, pl: SasrecPlan) -> torch.Tensor:
        if not hasattr(pl, "hidg"):
            pl.hidg = torch.zeros(pl.shape.B, lib().value("amid_scorer_vec_floats", pl.shape.NI, self.hid), dtype=torch.float32, device=self.device)
            torch.cuda.synchronize(self.device)
        return pl.hidg

    COMPACT_LIVE = True
    COMPACT_MIN_IDX = 65536    # shorter index lists gain nothing from the compact list (the tail is bound by the dense partial sums,
                               # the row Adam by its dense half) and lose the early fork of the side-stream sort: cfg 2 keeps the full list

    def live_forward_ok(self, pl: SasrecPlan) -> bool:
        """Whether this engine's train step on `pl` encodes the live sequences only (see _enqueue_fwd_bwd)."""
        return bool(getattr(pl, "strip", False) and not self.itc_bs and not self.dr and not self.inc_bs and self.FUSED_HEAD and self.LIVE_FORWARD
                    and lib().value("amid_attn_live_supported", pl.shape.Tenc, self.D, self.H, 1))

    def n_sparse_train(self, pl: SasrecPlan, dp: bool = False) -> int:
        """n_sparse() of this engine's train steps on `pl` (known before the step is enqueued); dp: of its data-parallel steps."""
        if self.compact_ok(pl):
            return pl.n_compact
        flag = "_in_local_grads" if dp else "_in_train_step"      # (the folded step forces the compact list whatever the index count)
        saved = getattr(self, flag, False)
        setattr(self, flag, True)
        try:
            folded = self.input_pool(pl) is not None and self._tail2_ok(pl)
        finally:
            setattr(self, flag, saved)
        return pl.n_compact if folded else pl.shape.n_idx

    def compact_ok(self, pl: SasrecPlan) -> bool:
        return bool(self.COMPACT_LIVE and pl.shape.n_idx >= self.COMPACT_MIN_IDX and self.live_forward_ok(pl))

    def n_sparse(self, pl: SasrecPlan) -> int:
        """Entries of the step's index list as the sparse side (sort, segment reduce, row Adam) sees it."""
        return pl.n_compact if getattr(pl, "compact", False) else pl.shape.n_idx

    # The lazy-Adam catch-up of a step can be FOLDED into the gather K1 (amid_embed_fwd_replay_f32): a lagging row's owed zero-gradient steps
    # are replayed in registers for the value the forward reads, and once more by the optimizer launch (which replays lagging rows anyway)
    # in front of the real step -- one launch less, the replay done twice.  MEASURED (MI355X, cfg 2: ~2.5 k rows of a step lag by ~60 steps,
    # an epoch is 60 batches): the catch-up launch is not latency but replay arithmetic (two quarter-rate instructions per element and
    # step), so doing it twice LOSES -- catch-up 14.4 + K1 8.1 us -> K1 24.5 us, optimizer 7.4 -> 15.6 us, 0.3695 -> 0.3829 ms per step;
    # cfg 4 0.239 -> 0.268.  Off by default; "auto" folds only where the replay is next to nothing (an epoch of at most FOLD_MAX_GAP
    # batches, e.g. a handful of batches replayed over and over), "1" always.
    FOLD_CATCHUP = "0"
    FOLD_MAX_GAP = 8
    catchup_gap_hint: Optional[int] = None          # batches per epoch when no input pool says so (the CLI's per-batch path)

    def _fold_catchup(self, pl: SasrecPlan) -> bool:
        if self.inc_bs or getattr(self, "comp", "") or self.FOLD_CATCHUP in ("0", False):
            return False                             # (InnerComp / BERT4Rec comp gather with amid_gather_rows_f32: no replay there)
        if self.FOLD_CATCHUP in ("1", True):
            return True
        ent = self.input_pool(pl)
        gap = ent[0].shape[0] if ent is not None else self.catchup_gap_hint
        return gap is not None and gap <= self.FOLD_MAX_GAP

    def _w16_images(self, planes: int):
        """(host pointer array of the 24 encoder weights [layer][domain][q, k, v, o, conv1, conv2], the image buffer) for `planes` bf16
        planes per weight (1: operands rounded to bf16; 3: hi + mid + lo = the fp32 weight exactly)."""
        D, fp = self.D, self.dense
        # one buffer per plane count, never re-allocated: captured graphs hold these addresses, and a compute = "bf16" engine alternates between
        # three planes (its folded train step, round 6) and one (its evaluation forward)
        bufs = self.__dict__.setdefault("_w16_bufs", {})
        if planes not in bufs:
            bufs[planes] = torch.empty(2, 2, 6, planes, D * D, dtype=torch.bfloat16, device=self.device)
        self.w16 = bufs[planes]
        if getattr(self, "_w16_src", None) is None:
            srcs = []
            for l in (0, 1):
                for g in (1, 2):
                    srcs += [fp.ptr(f"sac{g}.attention_layers.{l}.in_proj_weight", None, j * D * D) for j in range(3)]
                    srcs += [fp.ptr(f"sac{g}.attention_layers.{l}.out_proj.weight"), fp.ptr(f"sac{g}.forward_layers.{l}.conv1.weight"),
                             fp.ptr(f"sac{g}.forward_layers.{l}.conv2.weight")]
            self._w16_src = ptr_array(srcs)
        return self._w16_src, self.w16

    def _p3_bwd_for(self, pl: SasrecPlan) -> bool:
        """Whether this plan's backward strips take their data-gradient products on bf16 pieces (three-plane images of the transposes)."""
        return bool(self._ceff() != "bf16" and self.BWD_SPLIT and pl.strip and self.D == 128 and not self._seq_backward(pl))

    def _wT16x3_buf(self):
        if not hasattr(self, "wT16x3"):
            self.wT16x3 = torch.empty(2, 2, 6, 3, self.D * self.D, dtype=torch.bfloat16, device=self.device)
        return self.wT16x3

    def _fwd_on_pieces(self, pl: SasrecPlan, B: int, T: int) -> bool:
        """Whether this step's encoder forward is the one-launch kernel with its products on bf16 pieces (amid_sas_seq_fwd_split_f32)."""
        return bool(self._ceff() != "bf16" and self.FWD_SPLIT and self.D == 128 and pl.strip and self.SEQ_FORWARD and not self.inc_bs
                    and lib().value("amid_sas_seq_supported", B, T, self.D, self.H))

    def _enqueue_k1(self, pl: SasrecPlan, pos0, pos1, tmq, tr: int, p_drop: float, lf) -> None:
        """The gather K1 of a forward in the variant the step needs: over every sequence or the live list `lf`, writing the compact index
        list, with the folded catch-up (+ phase 1 of a riding sort)."""
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T, NI = shp.B, shp.Tenc, shp.NI
        st = self.step_state.data_ptr()
        # what this launch's riders write is decided HERE: a flag left by an earlier forward (e.g. a train-mode forward without a backward)
        # must not make the next backward skip its images
        pl.w16_written = pl.wT16x3_written = False
        compact = lf is not None and getattr(pl, "compact", False) and not getattr(pl, "tail2", False)      # (tail2: the step head wrote the list)
        ic, rc = (pl.idx_c.data_ptr(), pl.row_c.data_ptr()) if compact else (None, None)
        if getattr(pl, "fold_catchup", False):
            pl.fold_catchup = False
            ride = getattr(pl, "riding", False)
            L.call("amid_embed_fwd_replay_f32", self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(),
                   pl.idx_all.data_ptr(), pos0, pos1, B, T, D, B * NI, pl.xg.data_ptr(), tmq, st, tr, p_drop, lf, ic, rc, st,
                   self._sort_plan(pl) if ride else None, 1 if ride else 0, s)
        elif self._fwd_on_pieces(pl, B, T):          # the gather's extra workgroups write this step's weight images (three bf16 planes each)
            src, w16 = self._w16_images(3)
            wt = self._wT16x3_buf() if tr and self._p3_bwd_for(pl) else None     # (+ the transposes' images for this step's backward strips)
            L.call("amid_embed_fwd_w16_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), pos0, pos1, B, T, D, B * NI, pl.xg.data_ptr(), tmq, st,
                   tr, p_drop, lf, ic, rc, src, 24, 3, w16.data_ptr(), wt.data_ptr() if wt is not None else None, s)
            pl.w16_written = True
            pl.wT16x3_written = wt is not None
        elif compact:
            L.call("amid_embed_fwd_live_compact_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), pos0, pos1, B, T, D, B * NI, pl.xg.data_ptr(),
                   tmq, st, tr, p_drop, lf, ic, rc, s)
        elif lf is not None:
            L.call("amid_embed_fwd_live_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), pos0, pos1, B, T, D, B * NI, pl.xg.data_ptr(), tmq, st,
                   tr, p_drop, lf, s)
        else:
            L.call("amid_embed_fwd_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), pos0, pos1, B, T, D, B * NI, pl.xg.data_ptr(), tmq, st, tr,
                   p_drop, s)
        if getattr(self, "_sort_owed", False):      # a deferred side-stream sort starts behind K1 (the compact list is K1's by-product; with
            if compact:                              # the folded catch-up the fork waited so that the main branch is captured first)
                self.ev_idx.record(self.stream)
            self.enqueue_sort(pl)

    def enqueue_catchup(self, pl: SasrecPlan) -> None:
        """Replay pending zero-gradient Adam steps of the rows this batch is about to gather (by position: no sort needed) -- as a
        launch of its own, or folded into the gather that follows (_fold_catchup)."""
        self._ensure_opt_state()
        if getattr(pl, "tail2", False):           # the step head replayed them
            pl.fold_catchup = False
            return
        chain = getattr(pl, "chain", False)
        pl.fold_catchup = self._fold_catchup(pl) and not chain       # (the chain rides in the catch-up's own launch)
        if pl.fold_catchup:
            return
        if chain:                                 # the step's whole sort rides here (five phases, the riders' own barrier between them)
            lib().call("amid_lazy_adam_catchup_positions_sort_f32", self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(),
                       self.table_last.data_ptr(), pl.idx_all.data_ptr(), pl.shape.n_idx, self.D, self.step_state.data_ptr(),
                       self._sort_plan(pl), 6, self.s)
            return
        if getattr(pl, "riding", False):          # phase 1 of the step's sort rides here
            lib().call("amid_lazy_adam_catchup_positions_sort_f32", self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(),
                       self.table_last.data_ptr(), pl.idx_all.data_ptr(), pl.shape.n_idx, self.D, self.step_state.data_ptr(),
                       self._sort_plan(pl), 1, self.s)
            return
        lib().call("amid_lazy_adam_catchup_positions_f32", self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(),
                   self.table_last.data_ptr(), pl.idx_all.data_ptr(), pl.shape.n_idx, self.D, self.step_state.data_ptr(), self.s)

    def enqueue_forward(self, pl: SasrecPlan, train: bool, with_loss: bool, sum_loss: bool = True) -> None:
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T, NI, M = shp.B, shp.Tenc, shp.NI, shp.M
        st = self.step_state.data_ptr()
        tr = 1 if train else 0
        fp = self.dense
        pl.w16_written = pl.wT16x3_written = False      # (set again by the gather K1 of THIS forward when its riders write the images)
        pl.head_done = False                            # (set by THIS forward when its workgroups run the head: HEAD_ON_FWD)
        pl.lnstat_fwd = False                           # (set by THIS forward when it stores row statistics instead of qn / y)
        # the train step's own loss reads only the sequence (domain_id[b], b) of every sample (see _enqueue_fwd_bwd): those B "live"
        # sequences are listed on the device; with live_fwd the forward encodes nothing else
        lv = self._live_list(pl)
        live_fwd = lv is not None and getattr(self, "_live_fwd", False)
        lf = lv if live_fwd else None
        if lv is not None and not getattr(pl, "live_packed", False):
            L.call("amid_live_list_i32", pl.domain.data_ptr(), B, pl.live.data_ptr(), s)
        if self.inc_bs:      # plain gather, InnerComp's token group, then the 2T-token encoder input (csrc/innercomp.hip)
            L.call("amid_gather_rows_f32", self.table.data_ptr(), self.n_rows, D, pl.idx_all.data_ptr(), 0, shp.n_idx, pl.xg.data_ptr(), None, s)
            L.call("amid_inc_score_f32", pl.xg.data_ptr(), B, shp.T, D, pl.inc_s.data_ptr(), s)
            wts = (self._pp("inc_d{d}.trans_nn.weight"), self._pp("inc_d{d}.trans_nn.bias"), self._pp("inc_d{d}.trans_bs.weight"),
                   self._pp("inc_d{d}.trans_bs.bias"))
            out = (pl.inc_gate.data_ptr(), pl.inc_S.data_ptr(), pl.inc_Z.data_ptr(), pl.inc_sw.data_ptr(), pl.x[0].data_ptr(),
                   pl.tmq.data_ptr(), st, tr, SASREC_P_DROP, s)
            if getattr(pl, "inc_world", 1) > 1:
                # data parallel: the softmax over the batch and Linear(bs, 1) span the GLOBAL batch (model_seq.py:465-469).  The ranks
                # all-gather their scores (per domain, rank order = sample order), each forms the gates of its own rows and its partial
                # token sums S, the partial sums are all-reduced, and every rank finishes Z -- the same group on every rank
                ex = self._inc_exchange(pl)
                self._coll(lambda: [ex.all_gather_packed(pl.inc_s[g], pl.inc_s_g[g]) for g in (0, 1)])
                shard = (B, shp.T, D, self.inc_bs, ex.rank * B)
                L.call("amid_inc_embed_fwd_shard_f32", pl.xg.data_ptr(), pl.inc_s_g.data_ptr(), *wts, self.inc_threshold,
                       fp.ptr("sac1.pos_emb.weight"), fp.ptr("sac2.pos_emb.weight"), *shard, 1, *out)
                self._coll(lambda: ex.all_reduce_dense(pl.inc_S))
                L.call("amid_inc_embed_fwd_shard_f32", pl.xg.data_ptr(), pl.inc_s_g.data_ptr(), *wts, self.inc_threshold,
                       fp.ptr("sac1.pos_emb.weight"), fp.ptr("sac2.pos_emb.weight"), *shard, 2, *out)
            else:
                L.call("amid_inc_embed_fwd_f32", pl.xg.data_ptr(), pl.inc_s.data_ptr(), *wts, self.inc_threshold,
                       fp.ptr("sac1.pos_emb.weight"), fp.ptr("sac2.pos_emb.weight"), B, shp.T, D, *out)
        else:
            gat = bool(getattr(pl, "gather_on_fwd", False) and getattr(pl, "tail2", False) and lf is not None and tr)
            pl.gather_on_fwd = False                    # (a decision of THIS step's enqueue_prepare)
            if gat:                                     # the forward's workgroups gather their own rows; the step head wrote the images
                pl.w16_written = pl.wT16x3_written = True
            else:
                self._enqueue_k1(pl, fp.ptr("sac1.pos_emb.weight"), fp.ptr("sac2.pos_emb.weight"), pl.tmq.data_ptr(), tr, SASREC_P_DROP, lf)
        def layer_ptrs(l):
            pre = f"sac{{d}}"
            return ((self._pp(f"{pre}.attention_layernorms.{l}.weight"), self._pp(f"{pre}.attention_layernorms.{l}.bias"),
                     self._pp(f"{pre}.attention_layers.{l}.in_proj_weight"), self._pp(f"{pre}.attention_layers.{l}.in_proj_bias")),
                    (self._pp(f"{pre}.attention_layers.{l}.out_proj.weight"), self._pp(f"{pre}.attention_layers.{l}.out_proj.bias"),
                     self._pp(f"{pre}.forward_layernorms.{l}.weight"), self._pp(f"{pre}.forward_layernorms.{l}.bias"),
                     self._pp(f"{pre}.forward_layers.{l}.conv1.weight"), self._pp(f"{pre}.forward_layers.{l}.conv1.bias"),
                     self._pp(f"{pre}.forward_layers.{l}.conv2.weight"), self._pp(f"{pre}.forward_layers.{l}.conv2.bias")))

        qkv0, rest0 = layer_ptrs(0)
        qkv1, rest1 = layer_ptrs(1)

        def attn_fwd(l):
            if live_fwd:
                L.call("amid_attn_fwd_live_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), B, T, D, self.H, 1, l, st, tr,
                       SASREC_P_DROP, pl.o[l].data_ptr(), pl.stats[l].data_ptr(), lf, s)
            else:
                L.call("amid_attn_fwd_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), None, B, T, D, self.H, 1, l, st, tr,
                       SASREC_P_DROP, pl.o[l].data_ptr(), pl.stats[l].data_ptr(), s)

        if pl.strip and self.SEQ_FORWARD and not self.inc_bs and L.value("amid_sas_seq_supported", B, T, D, self.H):
            # the whole encoder -- both layers, attention cores included -- as ONE launch, a workgroup per sequence (csrc/sasrec_seq.hip)
            fam = lambda fmt: ptr_array([fp.ptr(fmt.format(d=d, l=l)) for l in (0, 1) for d in (1, 2)])      # noqa: E731  [layer][domain]
            key = ("seq_fwd", pl.x[0].data_ptr())
            c = self._ptr_cache.get(key)
            if c is None:
                tl = lambda ts: ptr_array([t.data_ptr() for t in ts])      # noqa: E731
                c = (tl(pl.x[:2]), fam("sac{d}.attention_layernorms.{l}.weight"), fam("sac{d}.attention_layernorms.{l}.bias"),
                     fam("sac{d}.attention_layers.{l}.in_proj_weight"), fam("sac{d}.attention_layers.{l}.in_proj_bias"),
                     fam("sac{d}.attention_layers.{l}.out_proj.weight"), fam("sac{d}.attention_layers.{l}.out_proj.bias"),
                     fam("sac{d}.forward_layernorms.{l}.weight"), fam("sac{d}.forward_layernorms.{l}.bias"),
                     fam("sac{d}.forward_layers.{l}.conv1.weight"), fam("sac{d}.forward_layers.{l}.conv1.bias"),
                     fam("sac{d}.forward_layers.{l}.conv2.weight"), fam("sac{d}.forward_layers.{l}.conv2.bias"),
                     tl(pl.qn), tl(pl.q), tl(pl.k), tl(pl.v), tl(pl.o), tl(pl.stats), tl(pl.r), tl(pl.y), tl(pl.h))
                self._ptr_cache[key] = c
            split = self._fwd_on_pieces(pl, B, T)
            if self._ceff() == "bf16" or split:       # this step's weights as bf16 fragment images (one plane: operands rounded to bf16;
                planes = 3 if split else 1            # three: hi + mid + lo = the fp32 weight exactly), then the forward on them
                src, w16 = self._w16_images(planes)
                if not getattr(pl, "w16_written", False):      # (the train step's gather K1 wrote them with extra workgroups)
                    L.call("amid_sas_weights_bf16_planes", src, 24, D, 0, planes, w16.data_ptr(), s)
                pl.w16_written = False
                if split and getattr(pl, "tail2", False) and lf is not None:
                    # the folded step: qn / y are not stored -- row statistics instead (pl.ln_stat); the weight gradients rebuild them
                    # (c: x, 12 parameter families, qn, q, k, v, o, stats, r, y, h)
                    gather = (self.table.data_ptr(), pl.idx_all.data_ptr(), fp.ptr("sac1.pos_emb.weight"), fp.ptr("sac2.pos_emb.weight"))
                    if (self.HEAD_ON_FWD and getattr(self, "_fuse_head", False) and with_loss and not sum_loss and 16 < T <= 64
                            and self.hid <= 32 and NI <= 64):
                        # ... and a live sequence is a sample: its workgroup finishes with the sample's head (forward + loss + backward,
                        # what amid_head_fwd_bwd_own_vec_f32 does in enqueue_backward otherwise); the last layer's output is not stored
                        # (compute = "bf16" on the folded step: the same launch multiplying ONE piece per operand, BF16_FWD_ONE)
                        p1 = "_p1" if (gat and self._bf16_as_f32 and self.BF16_FWD_ONE) else ""
                        L.call(f"amid_sas_seq_fwd_gather_head{p1}_f32" if gat else "amid_sas_seq_fwd_split_lnstat_head_f32", 2, c[0],
                               pl.x[2].data_ptr() if self.HEAD_ON_FWD_KEEPS_X else None, *c[1:13], self._ln_stat(pl)[1], *c[14:20], c[21],
                               pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, self.H, lf, st, tr, SASREC_P_DROP, w16.data_ptr(),
                               self._pp("sac{d}.last_layernorm.weight"), self._pp("sac{d}.last_layernorm.bias"),
                               pl.xg.data_ptr() + 4 * 2 * shp.Mi * D, fp.ptr("predictModule.fc.0.weight"), fp.ptr("predictModule.fc.0.bias"),
                               fp.ptr("predictModule.fc.2.weight"), fp.ptr("predictModule.fc.2.bias"), pl.labels.data_ptr(),
                               pl.domain.data_ptr(), NI, self.hid, pl.u.data_ptr(), pl.p1.data_ptr(), pl.p2.data_ptr(), pl.dp1.data_ptr(),
                               pl.dp2.data_ptr(), pl.loss_part.data_ptr(), pl.dxbuf.data_ptr(), pl.dxg.data_ptr() + 4 * 2 * shp.Mi * D,
                               pl.last_part.data_ptr(), self._hidg(pl).data_ptr(), *(gather if gat else ()), s)
                        pl.head_done = True
                    elif gat:
                        p1 = "_p1" if (self._bf16_as_f32 and self.BF16_FWD_ONE) else ""
                        L.call(f"amid_sas_seq_fwd_gather{p1}_f32", 2, c[0], pl.x[2].data_ptr(), *c[1:13], self._ln_stat(pl)[1], *c[14:20], c[21],
                               pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, self.H, lf, st, tr, SASREC_P_DROP, w16.data_ptr(),
                               pl.xg.data_ptr() + 4 * 2 * shp.Mi * D, NI, *gather, s)
                    else:
                        L.call("amid_sas_seq_fwd_split_lnstat_f32", 2, c[0], pl.x[2].data_ptr(), *c[1:13], self._ln_stat(pl)[1], *c[14:20], c[21],
                               pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, self.H, lf, st, tr, SASREC_P_DROP, w16.data_ptr(), s)
                    pl.lnstat_fwd = True
                else:
                    pl.lnstat_fwd = False
                    L.call("amid_sas_seq_fwd_split_f32" if split else "amid_sas_seq_fwd_bf16w_f32", 2, c[0], pl.x[2].data_ptr(), *c[1:],
                           pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, self.H, lf, st, tr, SASREC_P_DROP, w16.data_ptr(), s)
            else:
                L.call("amid_sas_seq_fwd_f32", 2, c[0], pl.x[2].data_ptr(), *c[1:], pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, self.H, lf, st, tr,
                       SASREC_P_DROP, s)
        elif pl.strip:       # register-resident strip chains (csrc/sasrec_strip.hip): same operations, operands and saved tensors
            L.call("amid_sas_strip_qkv_fwd_f32", pl.x[0].data_ptr(), *qkv0, SASREC_LN_EPS, B, T, D, lf, pl.qn[0].data_ptr(), pl.q[0].data_ptr(),
                   pl.k[0].data_ptr(), pl.v[0].data_ptr(), s)
            attn_fwd(0)
            L.call("amid_sas_strip_oproj_ffn_fwd_f32", pl.o[0].data_ptr(), pl.qn[0].data_ptr(), *rest0, pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, lf,
                   0, st, tr, SASREC_P_DROP, pl.r[0].data_ptr(), pl.y[0].data_ptr(), pl.h[0].data_ptr(), pl.x[1].data_ptr(), *qkv1,
                   pl.qn[1].data_ptr(), pl.q[1].data_ptr(), pl.k[1].data_ptr(), pl.v[1].data_ptr(), s)
            attn_fwd(1)
            L.call("amid_sas_strip_oproj_ffn_fwd_f32", pl.o[1].data_ptr(), pl.qn[1].data_ptr(), *rest1, pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, lf,
                   1, st, tr, SASREC_P_DROP, pl.r[1].data_ptr(), pl.y[1].data_ptr(), pl.h[1].data_ptr(), pl.x[2].data_ptr(), None, None, None,
                   None, None, None, None, None, s)
        else:
            L.call("amid_sas_qkv_fwd_f32" + pl.rt_suffix, pl.x[0].data_ptr(), *qkv0, SASREC_LN_EPS, M, D, pl.rpt, pl.qn[0].data_ptr(), pl.q[0].data_ptr(),
                   pl.k[0].data_ptr(), pl.v[0].data_ptr(), self.mma_bf16, s)
            for l in (0, 1):
                attn_fwd(l)
                rest = rest0 if l == 0 else rest1
                if l == 0:      # layer 0's out-projection + feed-forward and layer 1's LayerNorm + q / k / v: one launch
                    L.call("amid_sas_oproj_ffn_qkv_fwd_f32" + pl.rt_suffix, pl.o[0].data_ptr(), pl.qn[0].data_ptr(), *rest, pl.tmq.data_ptr(), SASREC_LN_EPS, M, D,
                           pl.rpt, 0, st, tr, SASREC_P_DROP, pl.r[0].data_ptr(), pl.y[0].data_ptr(), pl.h[0].data_ptr(), pl.x[1].data_ptr(),
                           *qkv1, pl.qn[1].data_ptr(), pl.q[1].data_ptr(), pl.k[1].data_ptr(), pl.v[1].data_ptr(), self.mma_bf16, s)
                else:
                    L.call("amid_sas_oproj_ffn_fwd_f32" + pl.rt_suffix, pl.o[l].data_ptr(), pl.qn[l].data_ptr(), *rest, pl.tmq.data_ptr(), SASREC_LN_EPS, M, D,
                           pl.rpt, l, st, tr, SASREC_P_DROP, pl.r[l].data_ptr(), pl.y[l].data_ptr(), pl.h[l].data_ptr(), pl.x[l + 1].data_ptr(),
                           self.mma_bf16, s)
        items = pl.xg.data_ptr() + 4 * 2 * shp.Mi * D
        if (self.dr or self.itc_bs) and getattr(self, "_fuse_scorers", False) and with_loss and not sum_loss:
            self._enqueue_user_vectors(pl)           # train step: the scorers run as ONE forward + loss + backward launch in enqueue_backward
            return
        if self.dr:
            self._enqueue_head_dr_fwd(pl, items, with_loss)
            return
        if self.itc_bs:
            self._enqueue_head_itc_fwd(pl, items, with_loss)
            if with_loss and sum_loss:
                L.call("amid_sum_vector_f32", pl.loss_part.data_ptr(), B, pl.loss.data_ptr(), s)
            return
        if getattr(self, "_fuse_head", False) and with_loss and not sum_loss:
            return                                   # train step: the head runs as ONE forward + backward launch in enqueue_backward
        L.call("amid_head_fwd_f32", pl.x[2].data_ptr(), self._pp("sac{d}.last_layernorm.weight"), self._pp("sac{d}.last_layernorm.bias"), items,
               fp.ptr("predictModule.fc.0.weight"), fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"),
               fp.ptr("predictModule.fc.2.bias"), pl.labels.data_ptr() if with_loss else None, pl.domain.data_ptr() if with_loss else None,
               B, T, NI, D, self.hid, SASREC_LN_EPS, pl.u.data_ptr(), pl.p1.data_ptr(), pl.p2.data_ptr(),
               pl.dp1.data_ptr() if with_loss else None, pl.dp2.data_ptr() if with_loss else None,
               pl.loss_part.data_ptr() if with_loss else None, s)
        if with_loss and sum_loss:
            L.call("amid_sum_vector_f32", pl.loss_part.data_ptr(), B, pl.loss.data_ptr(), s)

    # ---- isItC head: last LayerNorm + mean -> pair-max -> batch-softmax gate + mix -> scorer (csrc/intercomp.hip) ----
    def _enqueue_user_vectors(self, pl: SasrecPlan) -> None:
        """pl.u [2, B, D]: mean over time of the last LayerNorm (:385, :432-434), mixed with the InterComp group when isItC."""
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T = shp.B, shp.Tenc
        fp = self.dense
        if not self.itc_bs:
            L.call("amid_lnmean_fwd_f32", pl.x[2].data_ptr(), fp.ptr("sac1.last_layernorm.weight"), fp.ptr("sac1.last_layernorm.bias"),
                   fp.ptr("sac2.last_layernorm.weight"), fp.ptr("sac2.last_layernorm.bias"), B, T, D, SASREC_LN_EPS, pl.u.data_ptr(), s)
        else:            # the pair-max kernel has every LayerNorm'd row of (b) in LDS: it emits the means too
            L.call("amid_itc_pairmax_f32", pl.x[2].data_ptr(), self._pp("sac{d}.last_layernorm.weight"), self._pp("sac{d}.last_layernorm.bias"),
                   B, T, D, SASREC_LN_EPS, pl.itc_s.data_ptr(), pl.u_raw.data_ptr(), s)
            u_raw, itc_s, u_out, Bm = pl.u_raw, pl.itc_s, pl.u, B
            if pl.itc_world > 1:
                # data parallel: the module couples the rows of the GLOBAL batch (softmax over the batch model_seq.py:491, Linear(bs, 1)
                # over it :495).  Every rank gathers the shards' pair-max scalars and user vectors (rank r holds rows [r B, (r + 1) B)) and
                # evaluates the module on all bs rows -- the same arithmetic on the same values on every rank --, then keeps its rows.
                ex = self._itc_exchange(pl)
                self._coll(lambda: [ex.all_gather_packed(pl.u_raw[g].reshape(-1), pl.u_raw_g[g].reshape(-1)) for g in (0, 1)]
                           + [ex.all_gather_packed(pl.itc_s, pl.itc_s_g)])
                u_raw, itc_s, u_out, Bm = pl.u_raw_g, pl.itc_s_g, pl.u_g, self.itc_bs
            L.call("amid_itc_mix_fwd_f32", u_raw.data_ptr(), itc_s.data_ptr(), self._pp("itc_d{d}.trans_nn.weight"),
                   self._pp("itc_d{d}.trans_nn.bias"), self._pp("itc_d{d}.trans_bs.weight"), self._pp("itc_d{d}.trans_bs.bias"),
                   self.itc_threshold, Bm, D, pl.itc_gate.data_ptr(), pl.itc_z.data_ptr(), pl.itc_sw.data_ptr(), u_out.data_ptr(), s)
            if pl.itc_world > 1:
                r = self._itc_exchange(pl).rank
                pl.u.copy_(pl.u_g[:, r * B:(r + 1) * B])

    def _inc_exchange(self, pl: SasrecPlan):
        """The exchange of the running data-parallel step (train_step_dp) for a plan that holds a shard of InnerComp's global batch."""
        ex = getattr(self, "_dp_exchange", None)
        if ex is None or ex.world != pl.inc_world:
            raise ValueError(f"isInC: the batch must hold exactly bs = {self.inc_bs} rows (trans_bs is Linear(bs, 1) over the batch, "
                             f"model_seq.py:457); a batch of 1 / {pl.inc_world} of them is a data-parallel shard and can only be stepped by "
                             f"train_step_dp with an exchange over {pl.inc_world} ranks")
        return ex

    def _itc_exchange(self, pl: SasrecPlan):
        """The exchange of the running data-parallel step (train_step_dp) for a plan that holds a shard of InterComp's global batch."""
        ex = getattr(self, "_dp_exchange", None)
        if ex is None or ex.world != pl.itc_world:
            raise ValueError(f"isItC: the batch must hold exactly bs = {self.itc_bs} rows (trans_bs is Linear(bs, 1) over the batch, "
                             f"model_seq.py:480); a batch of 1 / {pl.itc_world} of them is a data-parallel shard and can only be stepped by "
                             f"train_step_dp with an exchange over {pl.itc_world} ranks")
        return ex

    def _enqueue_user_vectors_bwd(self, pl: SasrecPlan) -> None:
        """pl.du (gradient of pl.u) -> InterComp parameter gradients (isItC) -> dx of the last layer's output + LayerNorm partials."""
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T = shp.B, shp.Tenc
        fp, G = self.dense, self.dense.grad
        du = pl.du
        if self.itc_bs:
            du_in, u_raw, du_out, Bm = pl.du, pl.u_raw, pl.du_raw, B
            if pl.itc_world > 1:
                # data parallel: the gradients of every shard's user vectors are gathered and the module's backward runs on the global
                # batch on every rank: each rank's own rows of d u_raw then carry the terms of ALL ranks' losses (the group token is built
                # from everybody's rows), which that rank alone can send on through its encoders
                ex = self._itc_exchange(pl)
                self._coll(lambda: [ex.all_gather_packed(pl.du[g].reshape(-1), pl.du_g[g].reshape(-1)) for g in (0, 1)])
                du_in, u_raw, du_out, Bm = pl.du_g, pl.u_raw_g, pl.du_raw_g, self.itc_bs
            L.call("amid_itc_mix_bwd_f32", du_in.data_ptr(), u_raw.data_ptr(), pl.itc_gate.data_ptr(), pl.itc_z.data_ptr(),
                   pl.itc_sw.data_ptr(), self._pp("itc_d{d}.trans_nn.weight"), self._pp("itc_d{d}.trans_nn.bias"),
                   self._pp("itc_d{d}.trans_bs.weight"), Bm, D, du_out.data_ptr(), self._pp("itc_d{d}.trans_nn.weight", G),
                   self._pp("itc_d{d}.trans_nn.bias", G), self._pp("itc_d{d}.trans_bs.weight", G), self._pp("itc_d{d}.trans_bs.bias", G), s)
            if pl.itc_world > 1:
                ex = self._itc_exchange(pl)
                pl.du_raw.copy_(pl.du_raw_g[:, ex.rank * B:(ex.rank + 1) * B])
                # the module's own parameter gradients came out of the GLOBAL batch, identically on every rank: the step's dense exchange
                # sums the ranks' gradients, so each rank contributes its 1 / world share
                for d in (1, 2):
                    for name in ("trans_nn.weight", "trans_nn.bias", "trans_bs.weight", "trans_bs.bias"):
                        self.dense.view(f"itc_d{d}.{name}", G).mul_(1.0 / ex.world)
            du = pl.du_raw
        L.call("amid_lnmean_bwd_f32", pl.x[2].data_ptr(), du.data_ptr(), fp.ptr("sac1.last_layernorm.weight"),
               fp.ptr("sac2.last_layernorm.weight"), B, T, D, SASREC_LN_EPS, pl.dxbuf.data_ptr(), pl.last_part.data_ptr(), s)

    def _scorer_fwd(self, pl: SasrecPlan, head: str, items: int, p1, p2, with_loss: bool) -> None:
        shp, fp = pl.shape, self.dense
        lib().call("amid_scorer_fwd_f32", pl.u.data_ptr(), items, fp.ptr(f"{head}.fc.0.weight"), fp.ptr(f"{head}.fc.0.bias"),
                   fp.ptr(f"{head}.fc.2.weight"), fp.ptr(f"{head}.fc.2.bias"), pl.labels.data_ptr() if with_loss else None,
                   pl.domain.data_ptr() if with_loss else None, shp.B, shp.NI, self.D, self.hid, p1.data_ptr(), p2.data_ptr(),
                   pl.dp1.data_ptr() if with_loss else None, pl.dp2.data_ptr() if with_loss else None,
                   pl.loss_part.data_ptr() if with_loss else None, self.s)

    def _scorer_bwd(self, pl: SasrecPlan, head: str, items: int, ditems: int, p1, p2, dp1, dp2, part, accumulate: int) -> None:
        shp, fp = pl.shape, self.dense
        lib().call("amid_scorer_bwd_f32", pl.u.data_ptr(), items, fp.ptr(f"{head}.fc.0.weight"), fp.ptr(f"{head}.fc.0.bias"),
                   fp.ptr(f"{head}.fc.2.weight"), fp.ptr(f"{head}.fc.2.bias"), p1.data_ptr(), p2.data_ptr(), dp1.data_ptr(), dp2.data_ptr(),
                   shp.B, shp.NI, self.D, self.hid, pl.du.data_ptr(), ditems, part.data_ptr(), accumulate, self.s)

    # ---- isItC head: last LayerNorm + mean -> pair-max -> batch-softmax gate + mix -> scorer (csrc/intercomp.hip) ----
    def _enqueue_head_itc_fwd(self, pl: SasrecPlan, items: int, with_loss: bool) -> None:
        self._enqueue_user_vectors(pl)
        self._scorer_fwd(pl, "predictModule", items, pl.p1, pl.p2, with_loss)

    def _enqueue_head_itc_bwd(self, pl: SasrecPlan, items: int, ditems: int) -> None:
        self._scorer_bwd(pl, "predictModule", items, ditems, pl.p1, pl.p2, pl.dp1, pl.dp2, pl.sc_part, 0)
        self._enqueue_user_vectors_bwd(pl)

    # ---- isDR head (next-4): three scorers on the same (u, items) + the doubly-robust objective of the current mode ----
    def _enqueue_head_dr_fwd(self, pl: SasrecPlan, items: int, with_loss: bool) -> None:
        L, shp = lib(), pl.shape
        self._enqueue_user_vectors(pl)
        self._scorer_fwd(pl, "predictModule", items, pl.p1, pl.p2, False)
        self._scorer_fwd(pl, "predict_ips", items, pl.ips1, pl.ips2, False)
        self._scorer_fwd(pl, "predict_gfunc", items, pl.g1, pl.g2, False)
        if with_loss:
            pa = lambda a, b: ptr_array([a.data_ptr(), b.data_ptr()])       # noqa: E731
            L.call("amid_dr_loss_f32", pa(pl.p1, pl.p2), pa(pl.ips1, pl.ips2), pa(pl.g1, pl.g2), pl.labels.data_ptr(), pl.domain.data_ptr(),
                   pl.in_ob.data_ptr(), self.dr_mode, self.dr_e_w, shp.B, shp.NI, pa(pl.dp1, pl.dp2), pa(pl.dips1, pl.dips2),
                   pa(pl.dg1, pl.dg2), pl.dr_loss_part.data_ptr(), self.s)

    def _enqueue_head_dr_bwd(self, pl: SasrecPlan, items: int, ditems: int) -> None:
        self._scorer_bwd(pl, "predictModule", items, ditems, pl.p1, pl.p2, pl.dp1, pl.dp2, pl.sc_part, 0)
        self._scorer_bwd(pl, "predict_ips", items, ditems, pl.ips1, pl.ips2, pl.dips1, pl.dips2, pl.sc_part_ips, 1)
        self._scorer_bwd(pl, "predict_gfunc", items, ditems, pl.g1, pl.g2, pl.dg1, pl.dg2, pl.sc_part_g, 1)
        self._enqueue_user_vectors_bwd(pl)

    def _enqueue_scorers_fused(self, pl: SasrecPlan, items: int, ditems: int, tr_src, tr_dst, n_tr: int) -> None:
        """Every scorer of the model (one, or the three of isDR) forward + the row's loss terms + backward on pl.u in ONE launch
        (amid_scorer_multi_fwd_bwd_f32): pl.du, d items, per-head weight-gradient partials, loss partials."""
        shp, fp = pl.shape, self.dense
        heads = DR_HEADS if self.dr else DR_HEADS[:1]
        outs = [(pl.p1, pl.p2, pl.dp1, pl.dp2, pl.sc_part)]
        if self.dr:
            outs += [(pl.ips1, pl.ips2, pl.dips1, pl.dips2, pl.sc_part_ips), (pl.g1, pl.g2, pl.dg1, pl.dg2, pl.sc_part_g)]
        pa = lambda seq: ptr_array(list(seq))          # noqa: E731
        lib().call("amid_scorer_multi_fwd_bwd_f32", pl.u.data_ptr(), items, pa(fp.ptr(f"{h}.fc.0.weight") for h in heads),
                   pa(fp.ptr(f"{h}.fc.0.bias") for h in heads), pa(fp.ptr(f"{h}.fc.2.weight") for h in heads),
                   pa(fp.ptr(f"{h}.fc.2.bias") for h in heads), len(heads), pl.labels.data_ptr(), pl.domain.data_ptr(),
                   pl.in_ob.data_ptr() if self.dr else None, self.dr_mode if self.dr else 0, self.dr_e_w, shp.B, shp.NI, self.D, self.hid,
                   pa(o[0].data_ptr() for o in outs), pa(o[1].data_ptr() for o in outs), pa(o[2].data_ptr() for o in outs),
                   pa(o[3].data_ptr() for o in outs), pl.loss_part.data_ptr(), pl.dr_loss_part.data_ptr() if self.dr else None,
                   pl.du.data_ptr(), ditems, pa(o[4].data_ptr() for o in outs), tr_src, tr_dst, n_tr, self.s)

    def enqueue_backward(self, pl: SasrecPlan, train: bool) -> None:
        """Backward from pl.dp1 / pl.dp2 (dLoss/dp) to pl.uniq_grad (table rows) and dense.grad."""
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T, NI, M = shp.B, shp.Tenc, shp.NI, shp.M
        st = self.step_state.data_ptr()
        tr = 1 if train else 0
        fp = self.dense
        # transposed weights: in_proj q/k/v blocks, out_proj, conv1, conv2 for both layers and domains
        src, dst = [], []
        for l in (0, 1):
            for g in (0, 1):
                pre = f"sac{g + 1}"
                for j in range(3):
                    src.append(fp.ptr(f"{pre}.attention_layers.{l}.in_proj_weight", None, j * D * D))
                src.append(fp.ptr(f"{pre}.attention_layers.{l}.out_proj.weight"))
                src.append(fp.ptr(f"{pre}.forward_layers.{l}.conv1.weight"))
                src.append(fp.ptr(f"{pre}.forward_layers.{l}.conv2.weight"))
                dst += [self.wT[l, g, w].data_ptr() for w in range(6)]
        # compute = "bf16" on the strip path: the strip backward's data-gradient products take bf16 fragment images of the transposed weights
        # (the fp32 transposes are still refreshed: the row-tile fallbacks and tests read them)
        self._bf16_bwd = bool(self._ceff() == "bf16" and pl.strip)
        # (the fused per-sequence backward -- amid_sas_seq_bwd_f32 -- keeps the fp32 matrix instructions)
        self._p3_bwd = self._p3_bwd_for(pl)
        bf = 3 if self._p3_bwd else 1 if self._bf16_bwd else 0
        if self._bf16_as_f32 and self._p3_bwd:
            bf = 1                      # one bf16 piece: the strips' rings stream the hi plane of the three-plane images
        if self._bf16_bwd:
            if not hasattr(self, "wT16"):
                self.wT16 = torch.empty(2, 2, 6, D * D, dtype=torch.bfloat16, device=self.device)
            L.call("amid_sas_weights_bf16", ptr_array(src), len(src), D, 1, self.wT16.data_ptr(), s)
        if self._p3_bwd:
            if not getattr(pl, "wT16x3_written", False):           # (the train step's gather K1 wrote them with extra workgroups)
                L.call("amid_sas_weights_bf16_planes", ptr_array(src), len(src), D, 1, 3, self._wT16x3_buf().data_ptr(), s)
            pl.wT16x3_written = False
        items = pl.xg.data_ptr() + 4 * 2 * shp.Mi * D
        ditems = pl.dxg.data_ptr() + 4 * 2 * shp.Mi * D
        if (self.dr or self.itc_bs) and getattr(self, "_fuse_scorers", False):
            self._enqueue_scorers_fused(pl, items, ditems, ptr_array(src), ptr_array(dst), len(src))
            self._enqueue_user_vectors_bwd(pl)
        elif self.dr:
            L.call("amid_transpose_weights_f32", ptr_array(src), ptr_array(dst), len(src), D, s)
            self._enqueue_head_dr_bwd(pl, items, ditems)
        elif self.itc_bs:
            L.call("amid_transpose_weights_f32", ptr_array(src), ptr_array(dst), len(src), D, s)
            self._enqueue_head_itc_bwd(pl, items, ditems)
        elif getattr(pl, "head_done", False):
            pl.head_done = False          # (the forward's workgroups ran the head: amid_sas_seq_fwd_split_lnstat_head_f32 in enqueue_forward)
        elif getattr(self, "_fuse_head", False) and getattr(pl, "tail2", False) and getattr(self, "_live_fwd", False) and self._live_list(pl) is not None:
            # the folded step: the scorer's weight gradients leave the head as per-sample hidden gradients (pl.hidg; the gradient tail sums
            # them, amid_grad_tail_live_f32) and the fp32 transposes are not refreshed (this step's strips read the three-plane images)
            L.call("amid_head_fwd_bwd_own_vec_f32", pl.x[2].data_ptr(), self._pp("sac{d}.last_layernorm.weight"), self._pp("sac{d}.last_layernorm.bias"),
                   items, fp.ptr("predictModule.fc.0.weight"), fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"),
                   fp.ptr("predictModule.fc.2.bias"), pl.labels.data_ptr(), pl.domain.data_ptr(), B, T, NI, D, self.hid, SASREC_LN_EPS,
                   pl.u.data_ptr(), pl.p1.data_ptr(), pl.p2.data_ptr(), pl.dp1.data_ptr(), pl.dp2.data_ptr(), pl.loss_part.data_ptr(),
                   pl.dxbuf.data_ptr(), ditems, pl.last_part.data_ptr(), self._hidg(pl).data_ptr(), None, None, 0, s)
        elif getattr(self, "_fuse_head", False):
            L.call("amid_head_fwd_bwd_own_f32" if getattr(self, "_live_fwd", False) and self._live_list(pl) is not None else "amid_head_fwd_bwd_f32", pl.x[2].data_ptr(), self._pp("sac{d}.last_layernorm.weight"), self._pp("sac{d}.last_layernorm.bias"),
                   items, fp.ptr("predictModule.fc.0.weight"), fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"),
                   fp.ptr("predictModule.fc.2.bias"), pl.labels.data_ptr(), pl.domain.data_ptr(), B, T, NI, D, self.hid, SASREC_LN_EPS,
                   pl.u.data_ptr(), pl.p1.data_ptr(), pl.p2.data_ptr(), pl.dp1.data_ptr(), pl.dp2.data_ptr(), pl.loss_part.data_ptr(),
                   pl.dxbuf.data_ptr(), ditems, pl.last_part.data_ptr(), pl.sc_part.data_ptr(), ptr_array(src), ptr_array(dst), len(src), s)
        else:
            L.call("amid_head_bwd_f32", pl.x[2].data_ptr(), self._pp("sac{d}.last_layernorm.weight"), pl.u.data_ptr(), items,
                   fp.ptr("predictModule.fc.0.weight"), fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"),
                   fp.ptr("predictModule.fc.2.bias"), pl.p1.data_ptr(), pl.p2.data_ptr(), pl.dp1.data_ptr(), pl.dp2.data_ptr(), B, T, NI, D,
                   self.hid, SASREC_LN_EPS, pl.dxbuf.data_ptr(), ditems, pl.last_part.data_ptr(), pl.sc_part.data_ptr(),
                   ptr_array(src), ptr_array(dst), len(src), s)
        def ffn_bwd_args(l):
            return (pl.tmq.data_ptr(), pl.h[l].data_ptr(), pl.r[l].data_ptr(), self._pp(f"sac{{d}}.forward_layernorms.{l}.weight"),
                    self._wT(l, 4), self._wT(l, 5), self._wT(l, 3))

        lv = self._live_list(pl)
        live_attn = lv is not None and bool(lib().value("amid_attn_live_supported", T, D, self.H, 1))

        def attn_bwd(l):
            if live_attn:        # the live sequences only; the others' rows of dq / dk / dv are never read (strip kernels, wgrad hint)
                L.call("amid_attn_bwd_live_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.o[l].data_ptr(),
                       pl.stats[l].data_ptr(), pl.d_o.data_ptr(), B, T, D, self.H, 1, l, st, tr, SASREC_P_DROP, pl.dq_l[l].data_ptr(),
                       pl.dk_l[l].data_ptr(), pl.dv_l[l].data_ptr(), lv, s)
                return
            L.call("amid_attn_bwd_rows_f32", pl.q[l].data_ptr(), pl.k[l].data_ptr(), pl.v[l].data_ptr(), pl.o[l].data_ptr(),
                   pl.stats[l].data_ptr(), pl.d_o.data_ptr(), None, B, T, D, self.H, 1, l, st, tr, SASREC_P_DROP, pl.dq_l[l].data_ptr(),
                   pl.dk_l[l].data_ptr(), pl.dv_l[l].data_ptr(), self._own_rows(pl), s)

        # live: the three row-tile kernels walk the sequences that carry a gradient only (see _own_rows), re-tiled over the CUs
        dom = self._own_rows(pl)
        live = dom is not None and pl.live_rows
        tm, h1, r1, lnw1, w1T1, w2T1, woT1 = ffn_bwd_args(1)
        tm, h0, r0, lnw0, w1T0, w2T0, woT0 = ffn_bwd_args(0)
        dx_in = (pl.dx0 if self.inc_bs else pl.dxg).data_ptr()
        seq = bool(pl.strip and lv is not None and live and live_attn and self._seq_backward(pl))
        pl.seq_bwd_used = seq              # (what THIS backward ran: _seq_backward() depends on the step being enqueued; bench.py and tests read this)
        if seq:
            # the five launches below as one workgroup-long chain per live sequence
            pa = lambda ts: ptr_array([t.data_ptr() for t in ts])          # noqa: E731
            key = ("seq_bwd", id(pl), bf)
            c = self._ptr_cache.get(key)
            if c is None:
                def wts(which):
                    return ptr_array([(self.wT16 if bf else self.wT)[l, g, which].data_ptr() for l in (0, 1) for g in (0, 1)])
                def lnw(fmt):
                    return ptr_array([fp.ptr(fmt.format(d=g + 1, l=l)) for l in (0, 1) for g in (0, 1)])
                c = dict(h=pa(pl.h), r=pa(pl.r), x=pa(pl.x[:2]), q=pa(pl.q), k=pa(pl.k), v=pa(pl.v), o=pa(pl.o), stats=pa(pl.stats),
                         ln1=lnw("sac{d}.attention_layernorms.{l}.weight"), ln2=lnw("sac{d}.forward_layernorms.{l}.weight"),
                         wq=wts(0), wk=wts(1), wv=wts(2), wo=wts(3), w1=wts(4), w2=wts(5), dpre2=pa(pl.dpre2), dpre1=pa(pl.dpre1),
                         dr=pa(pl.dr), dq=pa(pl.dq_l), dk=pa(pl.dk_l), dv=pa(pl.dv_l), ln1p=pa(pl.ln1_part_s), ln2p=pa(pl.ln2_part_s))
                self._ptr_cache[key] = c
            L.call("amid_sas_seq_bwd_f32", 2, pl.dxbuf.data_ptr(), tm, c["h"], c["r"], c["x"], c["q"], c["k"], c["v"], c["o"], c["stats"],
                   c["ln1"], c["ln2"], c["wq"], c["wk"], c["wv"], c["wo"], c["w1"], c["w2"], SASREC_LN_EPS, B, T, D, self.H, lv, st, tr,
                   SASREC_P_DROP, c["dpre2"], c["dpre1"], c["dr"], pl.d_o.data_ptr(), c["dq"], c["dk"], c["dv"], dx_in, c["ln1p"], c["ln2p"], bf, s)
        elif pl.strip:
            ln1p, ln2p = pl.ln1_part, pl.ln2_part
            # a train step's index sort rides in these three launches (phases 2, 3, 4; phase 1 rode in the catch-up launch)
            riding = getattr(pl, "riding", False)
            sfx = "_sort" if riding else ""
            t2 = getattr(pl, "tail2", False)
            plan_addr = (self._sort_plan_c(pl) if t2 else self._sort_plan(pl)) if riding else None
            ride = lambda ph: (plan_addr, ph) if riding else ()      # noqa: E731
            L.call(f"amid_sas_strip_ffn_bwd{sfx}_f32", pl.dxbuf.data_ptr(), tm, h1, r1, lnw1, w1T1, w2T1, woT1, SASREC_LN_EPS, B, T, D, lv, 1, st, tr,
                   SASREC_P_DROP, pl.dpre2[1].data_ptr(), pl.dpre1[1].data_ptr(), pl.dr[1].data_ptr(), pl.d_o.data_ptr(), ln2p[1].data_ptr(),
                   *ride(2), bf, s)
            attn_bwd(1)
            # layer 1's q / k / v + LayerNorm1 backward and layer 0's feed-forward / out-projection backward: one launch
            if t2:      # ... whose idle CUs also sum the scorer's weight gradients from the head's per-sample hidden gradients
                G = fp.grad
                L.call("amid_sas_strip_qkv_bwd_sort_scorer_f32", pl.dq_l[1].data_ptr(), pl.dk_l[1].data_ptr(), pl.dv_l[1].data_ptr(), pl.dr[1].data_ptr(),
                       pl.x[1].data_ptr(), self._pp("sac{d}.attention_layernorms.1.weight"), self._wT(1, 0), self._wT(1, 1), self._wT(1, 2),
                       SASREC_LN_EPS, B, T, D, lv, ln1p[1].data_ptr(), tm, h0, r0, lnw0, w1T0, w2T0, woT0, 0, st, tr, SASREC_P_DROP,
                       pl.dpre2[0].data_ptr(), pl.dpre1[0].data_ptr(), pl.dr[0].data_ptr(), pl.d_o.data_ptr(), ln2p[0].data_ptr(), plan_addr, 3, bf,
                       self._hidg(pl).data_ptr(), pl.u.data_ptr(), items, NI, self.hid, fp.ptr("predictModule.fc.0.weight", G),
                       fp.ptr("predictModule.fc.0.bias", G), fp.ptr("predictModule.fc.2.weight", G), fp.ptr("predictModule.fc.2.bias", G), s)
            else:
                L.call(f"amid_sas_strip_qkv_bwd{sfx}_f32", pl.dq_l[1].data_ptr(), pl.dk_l[1].data_ptr(), pl.dv_l[1].data_ptr(), pl.dr[1].data_ptr(),
                       pl.x[1].data_ptr(), self._pp("sac{d}.attention_layernorms.1.weight"), self._wT(1, 0), self._wT(1, 1), self._wT(1, 2),
                       SASREC_LN_EPS, B, T, D, lv, None, ln1p[1].data_ptr(), tm, h0, r0, lnw0, w1T0, w2T0, woT0, 0, st, tr, SASREC_P_DROP,
                       pl.dpre2[0].data_ptr(), pl.dpre1[0].data_ptr(), pl.dr[0].data_ptr(), pl.d_o.data_ptr(), ln2p[0].data_ptr(), *ride(3), bf, s)
            attn_bwd(0)
            if t2:      # ... with the embedding layer's backward applied on the strip (no amid_embed_bwd launch)
                L.call("amid_sas_strip_qkv_bwd_emb_f32", pl.dq_l[0].data_ptr(), pl.dk_l[0].data_ptr(), pl.dv_l[0].data_ptr(), pl.dr[0].data_ptr(),
                       pl.x[0].data_ptr(), self._pp("sac{d}.attention_layernorms.0.weight"), self._wT(0, 0), self._wT(0, 1), self._wT(0, 2),
                       SASREC_LN_EPS, B, T, D, lv, dx_in, ln1p[0].data_ptr(), pl.tmq.data_ptr(), st, tr, SASREC_P_DROP, plan_addr, 4, bf, s)
            else:
                L.call(f"amid_sas_strip_qkv_bwd{sfx}_f32", pl.dq_l[0].data_ptr(), pl.dk_l[0].data_ptr(), pl.dv_l[0].data_ptr(), pl.dr[0].data_ptr(),
                       pl.x[0].data_ptr(), self._pp("sac{d}.attention_layernorms.0.weight"), self._wT(0, 0), self._wT(0, 1), self._wT(0, 2),
                       SASREC_LN_EPS, B, T, D, lv, dx_in, ln1p[0].data_ptr(), None, None, None, None, None, None, None, 0, None, 0, 0.0, None,
                       None, None, None, None, *ride(4), bf, s)
        else:
            rows, suf, rpt = ("_rows", pl.rt_suffix_v, pl.rpt_v) if live else ("", pl.rt_suffix, pl.rpt)
            ln1p, ln2p = (pl.ln1_part_v, pl.ln2_part_v) if live else (pl.ln1_part, pl.ln2_part)
            hint = (dom, B, T) if live else ()
            L.call(f"amid_sas_ffn_bwd{rows}_f32" + suf, pl.dxbuf.data_ptr(), tm, h1, r1, lnw1, w1T1, w2T1, woT1, SASREC_LN_EPS, M, D, rpt, 1, st, tr,
                   SASREC_P_DROP, pl.dpre2[1].data_ptr(), pl.dpre1[1].data_ptr(), pl.dr[1].data_ptr(), pl.d_o.data_ptr(),
                   ln2p[1].data_ptr(), self.mma_bf16, *hint, s)
            attn_bwd(1)
            # layer 1's q / k / v + LayerNorm1 backward and layer 0's feed-forward / out-projection backward: one launch
            L.call(f"amid_sas_qkv_ffn_bwd{rows}_f32" + suf, pl.dq_l[1].data_ptr(), pl.dk_l[1].data_ptr(), pl.dv_l[1].data_ptr(), pl.dr[1].data_ptr(),
                   pl.x[1].data_ptr(), self._pp("sac{d}.attention_layernorms.1.weight"), self._wT(1, 0), self._wT(1, 1), self._wT(1, 2),
                   SASREC_LN_EPS, M, D, rpt, pl.dxbuf.data_ptr(), ln1p[1].data_ptr(), tm, h0, r0, lnw0, w1T0, w2T0, woT0, 0, st, tr,
                   SASREC_P_DROP, pl.dpre2[0].data_ptr(), pl.dpre1[0].data_ptr(), pl.dr[0].data_ptr(), pl.d_o.data_ptr(),
                   ln2p[0].data_ptr(), self.mma_bf16, *hint, s)
            attn_bwd(0)
            L.call(f"amid_sas_qkv_bwd{rows}_f32" + suf, pl.dq_l[0].data_ptr(), pl.dk_l[0].data_ptr(), pl.dv_l[0].data_ptr(), pl.dr[0].data_ptr(),
                   pl.x[0].data_ptr(), self._pp("sac{d}.attention_layernorms.0.weight"), self._wT(0, 0), self._wT(0, 1), self._wT(0, 2),
                   SASREC_LN_EPS, M, D, rpt, dx_in, ln1p[0].data_ptr(), self.mma_bf16, *hint, s)
        dy, xx = [], []
        for l in (0, 1):
            dy += [pl.dq_l[l].data_ptr(), pl.dk_l[l].data_ptr(), pl.dv_l[l].data_ptr(), pl.dr[l].data_ptr(), pl.dpre1[l].data_ptr(),
                   pl.dpre2[l].data_ptr()]
            xx += [pl.qn[l].data_ptr(), pl.x[l].data_ptr(), pl.x[l].data_ptr(), pl.o[l].data_ptr(), pl.y[l].data_ptr(), pl.h[l].data_ptr()]
        # NOTE: pl.x[0] is the gathered-row buffer xg, still intact here (its gradient lives in dxg)
        t2 = bool(getattr(pl, "tail2", False) and pl.strip and not seq)
        if t2:      # the last phase of the step's index sort (run heads) rides in the weight gradients; the embedding backward is done
            if getattr(pl, "lnstat_fwd", False):      # the forward stored row statistics instead of qn / y: the operands of q's and conv1's
                for l in (0, 1):                       # gradients are rebuilt from x / r while they are staged
                    xx[6 * l + 0], xx[6 * l + 4] = pl.x[l].data_ptr(), pl.r[l].data_ptr()
                fam = lambda fmt: ptr_array([fp.ptr(fmt.format(d=d, l=l)) for l in (0, 1) for d in (1, 2)])      # noqa: E731  [layer][domain]
                self._wgrad_one = True              # (the folded bf16 step: this launch on ONE piece per operand, _wgrad_mode -> 4)
                try:
                    wmode = self._wgrad_mode(D)
                finally:
                    self._wgrad_one = False
                L.call("amid_sas_wgrad_rows_sort_ln_f32", ptr_array(dy), ptr_array(xx), 2, M, D, pl.splits,
                       ptr_array([pl.w_part[0].data_ptr(), pl.w_part[1].data_ptr()]), ptr_array([pl.b_part[0].data_ptr(), pl.b_part[1].data_ptr()]),
                       self._own_rows(pl), B, T, wmode, self._sort_plan_c(pl), self._ln_stat(pl)[1],
                       fam("sac{d}.attention_layernorms.{l}.weight"), fam("sac{d}.attention_layernorms.{l}.bias"),
                       fam("sac{d}.forward_layernorms.{l}.weight"), fam("sac{d}.forward_layernorms.{l}.bias"), s)
            else:
                L.call("amid_sas_wgrad_rows_sort_f32", ptr_array(dy), ptr_array(xx), 2, M, D, pl.splits,
                       ptr_array([pl.w_part[0].data_ptr(), pl.w_part[1].data_ptr()]), ptr_array([pl.b_part[0].data_ptr(), pl.b_part[1].data_ptr()]),
                       self._own_rows(pl), B, T, self._wgrad_mode(D), self._sort_plan_c(pl), s)
            self._enqueue_grad_tail(pl, live, seq)
            return
        L.call("amid_sas_wgrad_rows_f32", ptr_array(dy), ptr_array(xx), 2, M, D, pl.splits, ptr_array([pl.w_part[0].data_ptr(), pl.w_part[1].data_ptr()]),
               ptr_array([pl.b_part[0].data_ptr(), pl.b_part[1].data_ptr()]), self._own_rows(pl), B, T,
               self._wgrad_mode(D), s)
        if getattr(pl, "riding", False):     # the last phase of the step's index sort (run heads) rides here
            L.call("amid_embed_bwd_sort_f32", (pl.dx0 if self.inc_bs else pl.dxg).data_ptr(), pl.tmq.data_ptr(), B, T, D, pl.pos_splits,
                   pl.dpos_part.data_ptr(), st, tr, SASREC_P_DROP, dom if live else None, self._sort_plan(pl), 5, s)
        elif live:     # the dead sequences' rows of the encoder-input gradient were never written: zero-filled here, not read
            L.call("amid_embed_bwd_rows_f32", (pl.dx0 if self.inc_bs else pl.dxg).data_ptr(), pl.tmq.data_ptr(), B, T, D, pl.pos_splits,
                   pl.dpos_part.data_ptr(), st, tr, SASREC_P_DROP, dom, s)
        else:
            L.call("amid_embed_bwd_f32", (pl.dx0 if self.inc_bs else pl.dxg).data_ptr(), pl.tmq.data_ptr(), B, T, D, pl.pos_splits,
                   pl.dpos_part.data_ptr(), st, tr, SASREC_P_DROP, s)
        if self.inc_bs:      # InnerComp's parameter gradients; the rows' own halves + its share -> the table-row gradient buffer
            G = self.dense.grad
            head = (pl.dpos_part.data_ptr(), pl.pos_splits, pl.xg.data_ptr(), pl.dx0.data_ptr(), pl.inc_gate.data_ptr(), pl.inc_S.data_ptr(),
                    pl.inc_sw.data_ptr(), self._pp("inc_d{d}.trans_nn.weight"), self._pp("inc_d{d}.trans_nn.bias"),
                    self._pp("inc_d{d}.trans_bs.weight"), B, shp.T, D)
            outs = (pl.inc_dZ.data_ptr(), pl.inc_dS.data_ptr(), pl.inc_rows.data_ptr(), self._pp("inc_d{d}.trans_nn.weight", G),
                    self._pp("inc_d{d}.trans_nn.bias", G), self._pp("inc_d{d}.trans_bs.weight", G), self._pp("inc_d{d}.trans_bs.bias", G),
                    pl.dxg.data_ptr(), s)
            if getattr(pl, "inc_world", 1) > 1:
                # data parallel: the token group is shared by the GLOBAL batch, so its gradient dZ is the sum of every rank's pos_emb
                # partials of rows T .. 2T - 1: all-reduced between the two phases.  W_nn / b_nn / b_bs gradients then come out alike on
                # every rank (1 / world each: the dense exchange sums the ranks); trans_bs.weight's gradient is this rank's slice
                ex = self._inc_exchange(pl)
                shard = (self.inc_bs, ex.rank * B)
                L.call("amid_inc_bwd_shard_f32", *head, *shard, 1, 1.0 / ex.world, *outs)
                self._coll(lambda: ex.all_reduce_dense(pl.inc_dZ))
                L.call("amid_inc_bwd_shard_f32", *head, *shard, 2, 1.0 / ex.world, *outs)
            else:
                L.call("amid_inc_bwd_f32", *head, *outs)
        self._enqueue_grad_tail(pl, live, seq)

    def _enqueue_grad_tail(self, pl: SasrecPlan, live: bool = False, seq: bool = False) -> None:
        """The two independent, bandwidth-bound ends of backward side by side in one launch: the fixed-order sum of every partial
        buffer (dense gradients + loss) and the segment reduce of the table-row gradients."""
        L, s, shp = lib(), self.s, pl.shape
        if getattr(self, "_sort_owed", False):    # no fork point of this engine's backward launched the deferred sort: do it now
            self.ev_idx.record(self.stream)
            self.enqueue_sort(pl)
        self.join_sort()                          # pos_sorted / seg_off come from the side-stream sort
        blk = (pl.red_blk_s if seq else getattr(pl, "red_blk_v", None) if live else None) or pl.red_blk   # per-entry block ranges of the partial sums
        ent, n_ent, ent_max = ((pl.red_entries_s, pl.red_n_s, pl.red_max_s) if seq else (pl.red_entries_v, pl.red_n_v, pl.red_max_v) if live
                               else (pl.red_entries, pl.red_n, pl.red_max))
        pk = getattr(self, "_tail_pack", None)
        pl.spans_done = False
        if getattr(pl, "tail2", False) and not getattr(self, "_in_train_step", False):
            # the folded step under data parallel: the exchange ships finished rows, so phase B of the segment reduce is a launch of its own
            # behind the tail -- and in graph A of the graph pair the packing of this rank's chunk (ids | rows | dense) rides in it
            ent_t, n_ent_t, blk_t = self._tail2_table(pl)
            fp = self.dense
            head = (pl.dxg.data_ptr(), pl.pos_sorted.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), pl.n_compact, self.D,
                    pl.seg_ws.data_ptr())
            mid = (ent_t.data_ptr(), n_ent_t, blk_t[0].data_ptr(), blk_t[1], pl.live.data_ptr(), shp.B, shp.Tenc,
                   fp.ptr("sac1.pos_emb.weight", fp.grad), fp.ptr("sac2.pos_emb.weight", fp.grad))
            if self.FUSED_OPT:        # ... all of it as ONE launch (amid_grad_tail_opt_f32's workgroups, shipping instead of applying)
                left_lo = fp.slots["predictModule.fc.0.weight"][0]      # the scorer's gradients: final before this launch (the strip riders)
                dense = (fp.grad.data_ptr(), fp.numel, left_lo, fp.numel, pl.uniq_ids.data_ptr(), pl.n_uniq.data_ptr(), pl.n_compact)
                if pk is not None:
                    from .dist import packed_rows
                    send, umax, with_dense = pk
                    id_rows, rows = packed_rows(umax, self.D)
                    L.call("amid_grad_tail_live_dp1_f32", *head, send.data_ptr() + 4 * id_rows * self.D, *mid, *dense, umax, self.n_rows,
                           send.data_ptr(), send.data_ptr() + 4 * rows * self.D if with_dense else None, pl.err.data_ptr(),
                           pl.tail_ticket.data_ptr(), s)
                else:
                    L.call("amid_grad_tail_live_dp1_f32", *head, pl.uniq_grad.data_ptr(), *mid, *dense, 0, 0, None, None, None,
                           pl.tail_ticket.data_ptr(), s)
            elif pk is not None:
                from .dist import packed_rows
                send, umax, with_dense = pk
                id_rows, rows = packed_rows(umax, self.D)
                L.call("amid_grad_tail_live_dp_f32", *head, send.data_ptr() + 4 * id_rows * self.D, *mid, pl.uniq_ids.data_ptr(),
                       pl.n_uniq.data_ptr(), umax, self.n_rows, send.data_ptr(), self.dense.grad.data_ptr() if with_dense else None,
                       send.data_ptr() + 4 * rows * self.D if with_dense else None, self.dense.numel if with_dense else 0,
                       pl.err.data_ptr(), s)
            else:
                L.call("amid_grad_tail_live_dp_f32", *head, pl.uniq_grad.data_ptr(), *mid, None, None, 0, 0, None, None, None, 0, None, s)
            pl.spans_done = True
            return
        pl.opt_done = False
        if getattr(pl, "tail2", False) and self.FUSED_OPT and self.table_m is not None and not self.dr:
            # the folded step's tail AND its optimizer as one launch (round 6): every producer of a gradient slice applies Adam on the spot
            ent_t, n_ent_t, blk_t = self._tail2_table(pl)
            fp = self.dense
            left_lo = fp.slots["predictModule.fc.0.weight"][0]          # the scorer's slots (the flat buffer's tail): summed by the strip riders
            L.call("amid_grad_tail_opt_f32", pl.dxg.data_ptr(), pl.pos_sorted.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), pl.n_compact,
                   self.D, pl.seg_ws.data_ptr(), pl.uniq_grad.data_ptr(), ent_t.data_ptr(), n_ent_t, blk_t[0].data_ptr(), blk_t[1],
                   pl.live.data_ptr(), shp.B, shp.Tenc, fp.ptr("sac1.pos_emb.weight", fp.grad), fp.ptr("sac2.pos_emb.weight", fp.grad),
                   fp.data.data_ptr(), fp.m.data_ptr(), fp.v.data_ptr(), fp.grad.data_ptr(), fp.numel, left_lo, fp.numel,
                   self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), pl.uniq_ids.data_ptr(),
                   pl.n_uniq.data_ptr(), pl.n_compact, self.grad_scale, self.step_state.data_ptr(), pl.tail_ticket.data_ptr(), s)
            pl.opt_done = True
            return
        if getattr(pl, "tail2", False):
            ent_t, n_ent_t, blk_t = self._tail2_table(pl)
            fp = self.dense
            L.call("amid_grad_tail_live_f32", pl.dxg.data_ptr(), pl.pos_sorted.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), pl.n_compact,
                   self.D, pl.seg_ws.data_ptr(), pl.uniq_grad.data_ptr(), ent_t.data_ptr(), n_ent_t, blk_t[0].data_ptr(), blk_t[1],
                   pl.live.data_ptr(), shp.B, shp.Tenc, fp.ptr("sac1.pos_emb.weight", fp.grad), fp.ptr("sac2.pos_emb.weight", fp.grad),
                   None, None, None, 0, 0, None, None, None, None, s)      # (the scorer sums rode in the middle strip launch)
            return
        if pk is not None:       # graph A of the data-parallel step: the tail also packs this rank's exchange chunk (ids | rows | dense)
            from .dist import packed_rows
            send, umax, with_dense = pk
            id_rows, rows = packed_rows(umax, self.D)
            L.call("amid_grad_tail_pack_f32", pl.dxg.data_ptr(), pl.pos_sorted.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), self.n_sparse(pl),
                   self.D, pl.seg_ws.data_ptr(), send.data_ptr() + 4 * id_rows * self.D, ent.data_ptr(),
                   n_ent, ent_max, pl.uniq_ids.data_ptr(), pl.n_uniq.data_ptr(), umax,
                   self.n_rows, send.data_ptr(), self.dense.grad.data_ptr() if with_dense else None,
                   send.data_ptr() + 4 * rows * self.D if with_dense else None, self.dense.numel if with_dense else 0,
                   pl.err.data_ptr(), blk[0].data_ptr(), blk[1], s)
            return
        # a single-GPU train step's optimizer launch follows: it finishes the runs that cross chunks itself (amid_optimizer_step_spans_f32;
        # the same additions in the same order as the spans launch this saves).  Not for enqueue_local_grads: the exchange ships uniq_grad
        spans_later = bool(self.FUSED_SPANS and getattr(self, "_in_train_step", False) and self.D in (64, 128, 256))
        left = getattr(pl, "red_left_s" if seq else "red_left_v" if (live and hasattr(pl, "red_left_v")) else "red_left", None)
        if spans_later and self.FUSED_OPT and self.table_m is not None and left is not None:
            # ... and the optimizer itself: the tail's workgroups apply Adam to every slice they finish (amid_grad_tail_opt_f32 without its
            # position-row role: here every dense gradient is an entry of the reduce table) -- one launch fewer in every single-GPU step
            fp = self.dense
            L.call("amid_grad_tail_opt_f32", pl.dxg.data_ptr(), pl.pos_sorted.data_ptr(), pl.seg_off.data_ptr(), pl.seg_of.data_ptr(), self.n_sparse(pl),
                   self.D, pl.seg_ws.data_ptr(), pl.uniq_grad.data_ptr(), ent.data_ptr(), n_ent, blk[0].data_ptr(), blk[1],
                   None, 0, 0, None, None, fp.data.data_ptr(), fp.m.data_ptr(), fp.v.data_ptr(), fp.grad.data_ptr(), fp.numel, *(left or (0, 0)),
                   self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), pl.uniq_ids.data_ptr(),
                   pl.n_uniq.data_ptr(), self.n_sparse(pl), self.grad_scale, self.step_state.data_ptr(), pl.tail_ticket.data_ptr(), s)
            pl.opt_done = True
            return
        pl.spans_owed = self.n_sparse(pl) if spans_later else 0
        L.call("amid_grad_tail_nospans_f32" if spans_later else "amid_grad_tail_f32", pl.dxg.data_ptr(), pl.pos_sorted.data_ptr(), pl.seg_off.data_ptr(),
               pl.seg_of.data_ptr(), self.n_sparse(pl), self.D, pl.seg_ws.data_ptr(), pl.uniq_grad.data_ptr(), ent.data_ptr(), n_ent, ent_max,
               blk[0].data_ptr(), blk[1], s)

    def enqueue_optimizer(self, pl: SasrecPlan, sparse=None) -> None:
        """Dense Adam on the flat buffer + lazy row Adam on (uniq_ids, uniq_grad, n_uniq); `sparse`
        overrides the plan's local lists with the world-merged ones under data parallelism."""
        L, s = lib(), self.s
        self._ensure_opt_state()
        fp = self.dense
        if sparse is None:
            ids, rows, nu, cap = pl.uniq_ids, pl.uniq_grad, pl.n_uniq, self.n_sparse(pl)
        else:
            ids, rows, nu = sparse
            cap = ids.numel()
        if getattr(pl, "opt_done", False):          # the gradient tail applied the step (amid_grad_tail_opt_f32)
            pl.opt_done = False
            if sparse is not None:
                raise RuntimeError("the gradient tail already applied this step's local gradients")
            return
        owed = int(getattr(pl, "spans_owed", 0))
        pl.spans_owed = 0
        t2 = getattr(pl, "tail2", False) and not getattr(pl, "spans_done", False)
        if sparse is None and (t2 or owed):      # the segment reduce's runs across chunks are finished (and applied) here
            L.call("amid_optimizer_step_spans_f32", fp.data.data_ptr(), fp.m.data_ptr(), fp.v.data_ptr(), fp.grad.data_ptr(), fp.numel,
                   self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), ids.data_ptr(),
                   nu.data_ptr(), cap, rows.data_ptr(), self.D, self.grad_scale, self.step_state.data_ptr(), pl.seg_off.data_ptr(),
                   pl.seg_of.data_ptr(), pl.n_compact if t2 else owed, pl.seg_ws.data_ptr(), s)
            return
        if owed:
            raise RuntimeError("the gradient tail left the chunk-crossing runs to an optimizer launch that takes merged lists")
        L.call("amid_optimizer_step_f32", fp.data.data_ptr(), fp.m.data_ptr(), fp.v.data_ptr(), fp.grad.data_ptr(), fp.numel,
               self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), ids.data_ptr(),
               nu.data_ptr(), cap, rows.data_ptr(), self.D, self.grad_scale, self.step_state.data_ptr(), s)

    def enqueue_step_begin(self) -> None:
        lib().call("amid_step_begin", self.step_state.data_ptr(), self.s)
        self.step += 1

    def enqueue_train_step(self, pl: SasrecPlan) -> None:
        """Whole step t on the current stream: t += 1; unique; catch-up; forward; loss; backward; Adam."""
        self._in_train_step = True
        try:
            self.enqueue_prepare(pl, sparse=True, bump_step=True, defer_sort=True)
            self.enqueue_catchup(pl)
            self._fork_sort(pl)
            self._enqueue_fwd_bwd(pl)
            self.enqueue_optimizer(pl)
        finally:
            self._in_train_step = False
            self._bf16_as_f32 = False

    # ------------------------------------------------------------------ data parallel (one process per GPU)
    def enqueue_local_grads(self, pl: SasrecPlan) -> None:
        """Everything of step t that needs no communication: t += 1 .. local segment-reduced gradients."""
        self._in_local_grads = True
        try:
            self.enqueue_prepare(pl, sparse=True, bump_step=True, defer_sort=True)
            self.enqueue_catchup(pl)
            self._fork_sort(pl)
            self._enqueue_fwd_bwd(pl)
        finally:
            self._in_local_grads = False
            self._bf16_as_f32 = False

    def _fork_sort(self, pl: SasrecPlan) -> None:
        """Start the side-stream sort behind the catch-up launch, beside the forward (forks behind the forward or beside the weight
        gradients were measured and lost: DESIGN.md section 5)."""
        if getattr(pl, "compact", False) or getattr(pl, "fold_catchup", False):
            return                                    # the compact index list is K1's by-product / no catch-up launch: the fork follows K1
        if getattr(self, "_sort_owed", False):
            self.enqueue_sort(pl)

    FUSED_HEAD = True          # the plain SASRec head (no isItC / isDR) can run forward + backward as one launch
    HEAD_ON_FWD = True         # the folded step's head on the tail of the forward's workgroups (one launch fewer: eleven)
    HEAD_ON_FWD_KEEPS_X = False   # (tests: the last layer's output is stored all the same)

    def _enqueue_fwd_bwd(self, pl: SasrecPlan) -> None:
        """Forward (loss included; its sum rides in the gradient tail) + backward of a training step."""
        self._fuse_head = self.FUSED_HEAD and not self.dr and not self.itc_bs
        self._fuse_scorers = self.FUSED_HEAD and bool(self.dr or self.itc_bs) and (not self.dr or pl.shape.NI <= 16)
        # the step's own loss masks row b's terms of domain 1 - domain_id[b] with zero (train_sr.py:205-211; the doubly-robust
        # objectives likewise, train_sr_dr.py:216-221, :392-394): the encoder of that domain gets an all-zero gradient for row b.
        # InterComp AFTER the encoders mixes the rows' user vectors, so every sequence has a gradient there.
        self._own_domain_only = not self.itc_bs
        # ... and neither do the other domain's logits of a sample: with the plain head (no isDR heads, no InterComp / InnerComp) and
        # the matrix-core attention kernels the forward encodes the B live sequences only.  model.forward, which RETURNS both
        # domains' logits (model_seq.py:442), keeps encoding everything.
        self._live_fwd = self._own_domain_only and self._fuse_head and self.live_forward_ok(pl)
        try:
            self.enqueue_forward(pl, train=True, with_loss=True, sum_loss=False)
            self.enqueue_backward(pl, train=True)
        finally:
            self._fuse_head = self._fuse_scorers = self._own_domain_only = self._live_fwd = False

    LIVE_FORWARD = True        # the fused train step may skip the forward of the sequences its loss never reads
    SEQ_FORWARD = True         # the encoder forward as one launch per step where csrc/sasrec_seq.hip covers the shape

    def _live_list(self, pl: SasrecPlan):
        """Device pointer of the plan's live-sequence list (filled by amid_live_list_i32 at the head of enqueue_forward) when the
        step's loss masks the other domain of every sample and the encoder runs on the strip kernels, else None."""
        return pl.live.data_ptr() if getattr(self, "_own_domain_only", False) and pl.strip else None

    def _own_rows(self, pl: SasrecPlan):
        """Device pointer of the batch's domain ids when backward may treat the other domain's sequences as gradient-free (see
        _enqueue_fwd_bwd), else None: a backward driven by someone else's loss (the autograd path) makes no such promise."""
        return pl.domain.data_ptr() if getattr(self, "_own_domain_only", False) else None

    # ------------------------------------------------------------------ evaluation: test(), train_sr.py:31-128
    # The evaluation batch in three launches (round 6; four while K1 is its own launch, GATHER_ON_FWD = False): index marshal + live list, K1 over the live sequences only (no candidate rows: the
    # head gathers them), the inference forward over the live sequences (nothing saved), amid_eval_head_f32 (LN_last + mean, the scorer over
    # the 1 + neg_nums candidates, masked BCE, the positive's rank).  test() reads only the own domain's logits of a sample (utils.py:21-40)
    # and masks the other domain's loss terms (train_sr.py:63-64), so nothing else is computed; model.forward keeps returning both.
    EVAL_FUSED = True

    def eval_fused_ok(self, pl: SasrecPlan) -> bool:
        return bool(self.EVAL_FUSED and not self.itc_bs and not self.inc_bs and not self.dr and not getattr(self, "comp", "") and pl.strip
                    and self.SEQ_FORWARD and self.input_pool(pl) is None
                    and lib().value("amid_sas_seq_supported", pl.shape.B, pl.shape.Tenc, self.D, self.H))

    def _eval_out(self, pl: SasrecPlan):
        """The batch's results, one int32 image [rank B | rank_raw B | loss_part B (fp32 bits)] (+ the scores, for tests)."""
        if not hasattr(pl, "ev_out"):
            B = pl.shape.B
            pl.ev_out = torch.zeros(3 * B, dtype=torch.int32, device=self.device)
            pl.ev_rank, pl.ev_rank_raw, pl.ev_loss_part = pl.ev_out[:B], pl.ev_out[B:2 * B], pl.ev_out[2 * B:].view(torch.float32)
            pl.ev_p = torch.zeros(B, pl.shape.NI, dtype=torch.float32, device=self.device)
            pl.ev_u = torch.zeros(B, self.D, dtype=torch.float32, device=self.device)
            torch.cuda.synchronize(self.device)
        return pl.ev_out

    def enqueue_eval(self, pl: SasrecPlan, fix_value: float, with_loss: bool = True, want_scores: bool = False, build_images: bool = True) -> None:
        """One test() batch from the plan's static inputs (load_batch / load_packed) to pl.ev_rank / ev_rank_raw / ev_loss_part.
        build_images=False: the forward's weight images are current (eval_epoch builds them once for all its batches: the weights do not
        change inside an evaluation)."""
        if not self.eval_fused_ok(pl):
            raise ValueError("enqueue_eval: this model / shape evaluates through enqueue_forward (eval_fused_ok)")
        L, s, shp, D = lib(), self.s, pl.shape, self.D
        B, T, NI = shp.B, shp.Tenc, shp.NI
        fp, st = self.dense, self.step_state.data_ptr()
        self._eval_out(pl)
        L.call("amid_pack_indices_live", pl.in_i_node.data_ptr(), pl.in_neg.data_ptr(), pl.in_seq_d1.data_ptr(), pl.in_seq_d2.data_ptr(),
               B, shp.T, NI - 1, self.n_rows, pl.idx_all.data_ptr(), pl.err.data_ptr(), None, pl.domain.data_ptr(), pl.live.data_ptr(), s)
        lf = pl.live.data_ptr()
        pos = (fp.ptr("sac1.pos_emb.weight"), fp.ptr("sac2.pos_emb.weight"))
        split = self._fwd_on_pieces(pl, B, T)
        pl.w16_written = pl.wT16x3_written = False
        gat = bool(split and self.GATHER_ON_FWD)          # the forward's workgroups gather their own rows: three launches a batch
        if gat:
            src, w16 = self._w16_images(3)
            if build_images:
                L.call("amid_sas_weights_bf16_planes", src, 24, D, 0, 3, w16.data_ptr(), s)
        elif split and build_images:          # the gather's extra workgroups write the weights' three-plane images (the forward's operands)
            src, w16 = self._w16_images(3)
            L.call("amid_embed_fwd_w16_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), *pos, B, T, D, 0, pl.xg.data_ptr(), pl.tmq.data_ptr(),
                   st, 0, SASREC_P_DROP, lf, None, None, src, 24, 3, w16.data_ptr(), None, s)
        else:
            if split:
                src, w16 = self._w16_images(3)
            L.call("amid_embed_fwd_live_f32", self.table.data_ptr(), pl.idx_all.data_ptr(), *pos, B, T, D, 0, pl.xg.data_ptr(), pl.tmq.data_ptr(),
                   st, 0, SASREC_P_DROP, lf, s)
        fam = lambda fmt: ptr_array([fp.ptr(fmt.format(d=d, l=l)) for l in (0, 1) for d in (1, 2)])      # noqa: E731  [layer][domain]
        key = ("eval_fwd", id(pl))
        c = self._ptr_cache.get(key)
        if c is None:
            c = (fam("sac{d}.attention_layernorms.{l}.weight"), fam("sac{d}.attention_layernorms.{l}.bias"),
                 fam("sac{d}.attention_layers.{l}.in_proj_weight"), fam("sac{d}.attention_layers.{l}.in_proj_bias"),
                 fam("sac{d}.attention_layers.{l}.out_proj.weight"), fam("sac{d}.attention_layers.{l}.out_proj.bias"),
                 fam("sac{d}.forward_layernorms.{l}.weight"), fam("sac{d}.forward_layernorms.{l}.bias"),
                 fam("sac{d}.forward_layers.{l}.conv1.weight"), fam("sac{d}.forward_layers.{l}.conv1.bias"),
                 fam("sac{d}.forward_layers.{l}.conv2.weight"), fam("sac{d}.forward_layers.{l}.conv2.bias"))
            self._ptr_cache[key] = c
        if gat:
            L.call("amid_sas_seq_fwd_gather_infer_f32", 2, pl.x[2].data_ptr(), *c, SASREC_LN_EPS, B, T, D, self.H, lf, w16.data_ptr(),
                   self.table.data_ptr(), pl.idx_all.data_ptr(), *pos, s)
        elif split:
            L.call("amid_sas_seq_fwd_split_infer_f32", 2, pl.x[0].data_ptr(), pl.x[2].data_ptr(), *c, pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D,
                   self.H, lf, w16.data_ptr(), s)
        else:              # D 64 / compute = "bf16" / FWD_SPLIT off: the saving forward over the live sequences
            tl = lambda ts: ptr_array([t.data_ptr() for t in ts])      # noqa: E731
            saved = (tl(pl.qn), tl(pl.q), tl(pl.k), tl(pl.v), tl(pl.o), tl(pl.stats), tl(pl.r), tl(pl.y), tl(pl.h))
            if self.compute == "bf16":
                src, w16 = self._w16_images(1)
                L.call("amid_sas_weights_bf16_planes", src, 24, D, 0, 1, w16.data_ptr(), s)
                L.call("amid_sas_seq_fwd_bf16w_f32", 2, tl(pl.x[:2]), pl.x[2].data_ptr(), *c, *saved, pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D,
                       self.H, lf, st, 0, SASREC_P_DROP, w16.data_ptr(), s)
            else:
                L.call("amid_sas_seq_fwd_f32", 2, tl(pl.x[:2]), pl.x[2].data_ptr(), *c, *saved, pl.tmq.data_ptr(), SASREC_LN_EPS, B, T, D, self.H,
                       lf, st, 0, SASREC_P_DROP, s)
        L.call("amid_eval_head_f32", pl.x[2].data_ptr(), self._pp("sac{d}.last_layernorm.weight"), self._pp("sac{d}.last_layernorm.bias"),
               self.table.data_ptr(), pl.idx_all.data_ptr() + 4 * 2 * shp.Mi, fp.ptr("predictModule.fc.0.weight"),
               fp.ptr("predictModule.fc.0.bias"), fp.ptr("predictModule.fc.2.weight"), fp.ptr("predictModule.fc.2.bias"),
               pl.labels.data_ptr() if with_loss else None, pl.domain.data_ptr(), B, T, NI, D, self.hid, SASREC_LN_EPS, float(fix_value),
               pl.ev_u.data_ptr() if want_scores else None, pl.ev_p.data_ptr() if want_scores else None, pl.ev_rank.data_ptr(),
               pl.ev_rank_raw.data_ptr(), pl.ev_loss_part.data_ptr() if with_loss else None, s)

    def capture_eval(self, pl: SasrecPlan, fix_value: float, with_loss: bool = True) -> None:
        """The evaluation batch as a hipGraph over the plan's static inputs (parameters are read at replay time and the forward's weight
        images are rebuilt by eval_epoch before its first replay: the graph stays valid across training steps; keyed by fix_value / with_loss)."""
        L = lib()
        self.enqueue_eval(pl, fix_value, with_loss)           # warm-up outside capture (dynamic-LDS attributes, code objects; builds the images)
        self.sync()
        L.call("amid_graph_capture_begin", self.s)
        try:
            self.enqueue_eval(pl, fix_value, with_loss, build_images=False)
        finally:
            out = ctypes.c_void_p()
            L.call("amid_graph_capture_end", self.s, ctypes.byref(out))
        if not hasattr(pl, "eval_graphs"):
            pl.eval_graphs = {}
        pl.eval_graphs[(float(fix_value), bool(with_loss))] = out.value

    def eval_epoch(self, pl: SasrecPlan, packed: torch.Tensor, fix_value: float, with_loss: bool = True, use_graph: bool = True) -> torch.Tensor:
        """Every batch of `packed` ([n_batches, in_words] int64: pack_epoch's images, resident in HBM) through the evaluation launches:
        one device copy of the image into the plan's static inputs, one graph replay, one device copy of the 3 B result words out.
        Returns [n_batches, 3 B] int32 (rank | rank_raw | loss_part bits) on the device; nothing is read back here."""
        if packed.dtype != torch.int64 or packed.dim() != 2 or packed.shape[1] != pl.in_words:
            raise ValueError(f"eval_epoch takes [n, {pl.in_words}] int64 batch images (pack_epoch)")
        if self.table_m is not None:
            self.flush_table()                   # rows with pending zero-gradient Adam steps must be current for evaluation
        out = torch.empty(packed.shape[0], 3 * pl.shape.B, dtype=torch.int32, device=self.device)
        key = (float(fix_value), bool(with_loss))
        if use_graph and key not in getattr(pl, "eval_graphs", {}):
            with torch.cuda.stream(self.stream):
                pl.in_pack.copy_(packed[0], non_blocking=True)
            self.capture_eval(pl, fix_value, with_loss)
        L = lib()
        with torch.cuda.stream(self.stream):
            if self._fwd_on_pieces(pl, pl.shape.B, pl.shape.Tenc):      # this evaluation's weight images, once (the batches' launches only read them)
                src, w16 = self._w16_images(3)
                L.call("amid_sas_weights_bf16_planes", src, 24, self.D, 0, 3, w16.data_ptr(), self.s)
            for i in range(packed.shape[0]):
                pl.in_pack.copy_(packed[i], non_blocking=True)
                if use_graph:
                    L.call("amid_graph_launch", pl.eval_graphs[key], self.s)
                else:
                    self.enqueue_eval(pl, fix_value, with_loss, build_images=False)
                out[i].copy_(self._eval_out(pl), non_blocking=True)
        return out

    # ------------------------------------------------------------------ parameter interchange
    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad(), torch.cuda.stream(self.stream):
            self.table.copy_(sd["item_emb_layer.emb_item.weight"].to(self.device, torch.float32))
            for name in self.dense.slots:
                self.dense.view(name).copy_(sd[name].to(self.device, torch.float32))
        self.sync()

    def state_dict(self) -> Dict[str, torch.Tensor]:
        out = {"item_emb_layer.emb_item.weight": self.table}
        for name in self.dense.slots:
            out[name] = self.dense.view(name)
        return out

    def check_index_error(self, pl: SasrecPlan) -> None:
        flags = int(pl.err.item())
        if flags != 0:
            pl.err.zero_()
            if flags & 2:      # AMID_FLAG_UMAX_EXCEEDED
                raise RuntimeError("amid_amd: a data-parallel step found more unique rows than the bound umax it was given "
                                   "(train_step_dp): its exchange chunk is corrupt")
            raise IndexError("amid_amd: item index out of range in the batch (nn.Embedding would raise here, model_seq.py:27-29)")
