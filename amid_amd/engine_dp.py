"""Data-parallel step of the engine (split out of engine.py): graph A (local gradients + this rank's packed chunk), the collective(s) of
amid_amd/dist.py, graph B (Adam over the world's chunks); and the HIP merge backend of the eager / owner-bucketed exchange.
The reference has no multi-GPU path (train_sr.py:473: DataParallel commented out)."""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import lib, ptr_array

from .plan import SasrecPlan


class DataParallelMixin:
    def enqueue_optimizer_gathered(self, be: "HipMergeBackend", recv: torch.Tensor, world: int, umax: int, dense_in_chunk: bool = True) -> None:
        """The data-parallel optimizer: ONE launch over the world's gathered chunks (ids | rows | dense gradient per rank) -- the
        rank-ordered sums of the dense parts and of the rows of equal ids happen inside it (amid_optimizer_step_gathered_f32), so
        the step needs no merge / segment-reduce launches after the all-gather.  dense_in_chunk=False: the chunks hold ids | rows
        only and dense.grad already is the world's sum (the caller's all-reduce)."""
        from .dist import packed_rows
        self._ensure_opt_state()
        fp, D = self.dense, self.D
        id_rows, rows = packed_rows(umax, D)
        lib().call("amid_optimizer_step_gathered_f32", fp.data.data_ptr(), fp.m.data_ptr(), fp.v.data_ptr(), fp.grad.data_ptr(), fp.numel,
                   self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(), self.table_last.data_ptr(), recv.data_ptr(),
                   world, umax, be.chunk_rows(umax, fp.grad if dense_in_chunk else None) * D, id_rows, rows * D if dense_in_chunk else -1, D,
                   self.n_rows, self.grad_scale, self.step_state.data_ptr(), self.s)

    # How the 1.7 MB flat dense gradient crosses the ranks in the graph-pair step: "gather" = behind the sparse rows inside the step's ONE
    # all-gather (every rank sums the world's copies in rank order: one collective's latency, world x 1.7 MB received per rank);
    # "allreduce" = its own RCCL all-reduce next to the all-gather of the sparse rows (two collectives, ~2 x 1.7 MB on the wire per
    # rank whatever the world size) -- what BASELINE.json's north_star words ("RCCL all-reduce of dense parameter grads").  Both are
    # bit-identical across replicas; bench.py --dense-exchange measures either.
    DENSE_EXCHANGE = "gather"
    DP_GRAPH_B = False           # the optimizer over the gathered chunks as a replayed one-kernel graph (True) or a plain launch (round 6)

    def train_step_dp(self, pl: SasrecPlan, exchange, use_graph: bool = False, umax: Optional[int] = None, dense: Optional[str] = None) -> None:
        """One data-parallel step: local grads -> dense all-reduce + ONE sparse all-gather -> merge -> Adam.
        umax: a bound on the world's largest unique-row count of this step if the host knows one (no host sync then, see
        dist.py) -- it MUST cover every rank's count: a step that finds more raises AMID_FLAG_UMAX_EXCEEDED in the plan's error word
        (check_index_error).  With use_graph and a known umax the step is graph A (local gradients + packing of this rank's chunk: ids,
        rows and, with dense="gather", the flat dense gradient behind them), the collective(s), graph B (rank-ordered sum of the dense
        parts, merge of the sparse parts, Adam): three or four host calls, and replicas that are bit-identical by construction; the
        pair of graphs is captured per distinct (umax, dense), so callers should pass a bucketed bound (bench.py: the pool's maximum).
        dense: "gather" | "allreduce" (default: DENSE_EXCHANGE)."""
        inc_dp = bool(self.inc_bs and exchange.active)
        if inc_dp:
            # InnerComp under data parallel: bs is the GLOBAL batch, the plan a shard of it; the module's softmax / Linear(bs, 1) over the
            # batch run IN FRONT of the encoders on all-gathered scores and all-reduced token sums, and the group's gradient is all-reduced
            # at the end of backward (engine.enqueue_train_step's isInC branches) -- collectives cannot sit inside a captured graph: the step is
            # captured in segments (below)
            if getattr(pl, "inc_world", 1) != exchange.world:
                raise ValueError(f"isInC data parallel: bs = {self.inc_bs} must be world x the per-rank batch ({exchange.world} x {pl.shape.B})")
        itc_dp = bool(self.itc_bs and exchange.active)
        if itc_dp:
            # InterComp under data parallel: bs is the GLOBAL batch, the plan a shard of it; the module's
            # softmax / Linear(bs, 1) over the batch run on gathered scalars and user vectors in the MIDDLE of forward and backward
            # (engine._enqueue_user_vectors*) -- collectives cannot sit inside a captured graph: the step is captured in segments (below)
            if getattr(pl, "itc_world", 1) != exchange.world:
                raise ValueError(f"isItC data parallel: bs = {self.itc_bs} must be world x the per-rank batch ({exchange.world} x {pl.shape.B})")
        # (round 5) with a known bound the comp steps are captured too, in SEGMENTS cut at their mid-step collectives (engine._coll):
        # graph | gathers | graph | ... | the step's exchange | graph B; without a bound (or before the capture) they run eagerly
        self._dp_mid_collectives = bool(itc_dp or inc_dp)
        self._dp_exchange = exchange if (itc_dp or inc_dp) else None
        try:
            self._train_step_dp(pl, exchange, use_graph, umax, dense)
        finally:
            self._dp_exchange = None

    def _train_step_dp(self, pl: SasrecPlan, exchange, use_graph: bool, umax: Optional[int], dense: Optional[str]) -> None:
        dense = dense or self.DENSE_EXCHANGE
        if dense not in ("gather", "allreduce"):
            raise ValueError(f"dense exchange must be 'gather' or 'allreduce', got {dense!r}")
        L = lib()
        if umax is not None:                   # a caller's bound may be rounded up past the plan's index count (e.g. to a multiple of 256):
            umax = max(1, min(int(umax), self.n_sparse_train(pl, dp=True)))      # clamp BEFORE it keys the captured graph pair (as dist.exchange_sparse does)
        with torch.cuda.stream(self.stream):
            self.grad_scale = exchange.grad_scale
            fast = (use_graph and umax is not None and exchange.active and hasattr(exchange.backend, "merge_packed")
                    and not exchange.use_owner(umax, self.D))      # the owner-bucketed exchange sizes its buffers per step: eager
            # (the segmented graph A's collectives are closures over the exchange of capture time: the pair is keyed by the exchange too)
            pair = getattr(pl, "dp_graphs", {}).get((self._graph_key(), umax, dense, id(exchange))) if fast else None
            if pair is not None:       # graph A (comp models: its segments with their collectives between them), the collective(s), graph B
                if isinstance(pair[0], list):
                    for kind, item in pair[0]:
                        if kind == "graph":
                            L.call("amid_graph_launch", item, self.s)
                        else:
                            item()
                else:
                    L.call("amid_graph_launch", pair[0], self.s)
                self.step += 1
                exchange.all_gather_packed(pair[2], pair[3])
                if dense == "allreduce":
                    exchange.all_reduce_dense(self.dense.grad)
                if self.DP_GRAPH_B:
                    L.call("amid_graph_launch", pair[1], self.s)
                else:          # graph B is ONE kernel: launched as such (a replayed graph starts ~10 us behind the collective, a plain launch less)
                    self.enqueue_optimizer_gathered(exchange.backend, pair[3], exchange.world, umax, dense_in_chunk=(dense == "gather"))
                return
            if use_graph and not getattr(self, "_dp_mid_collectives", False):
                L.call("amid_graph_launch", pl.graphs_local[self._graph_key()], self.s)
                self.step += 1
            else:
                self.enqueue_local_grads(pl)
            exchange.all_reduce_dense(self.dense.grad)
            merged = exchange.exchange_sparse(pl.uniq_ids, pl.uniq_grad, pl.n_uniq, umax=umax)
            self.enqueue_optimizer(pl, sparse=merged if exchange.active else None)
        if fast:                               # this step ran eagerly (it also warmed every kernel up); capture the pair for the next ones
            with torch.cuda.stream(self.stream):      # (the comp models' steps hold torch copies: they must land on the capturing stream)
                self._capture_dp_pair(pl, exchange, int(umax), dense)

    def _capture_dp_pair(self, pl: SasrecPlan, exchange, umax: int, dense: str = "gather") -> None:
        L, be = lib(), exchange.backend
        self.sync()
        if int(pl.n_uniq.item()) > umax:       # the eager step just ran with this bound: a bound that is already too small never gets captured
            raise ValueError(f"train_step_dp: umax = {umax} is smaller than this step's {int(pl.n_uniq.item())} unique rows")
        step0 = self.step
        in_chunk = dense == "gather"
        dgrad = self.dense.grad if in_chunk else None
        be.gather_buffer(exchange.world, umax, dense=dgrad)   # capacity errors are raised here, not in the middle of a stream capture
        if in_chunk:
            be.prepare_dense(exchange.world, umax, self.dense.grad)      # device tables are built here, not under capture
            # the one-launch tail writes the dense part of the chunk slice by slice as it finishes them: the 16-byte padding between the
            # slots is never written and must read as zero on every rank (the receivers sum the whole part) -- cleared here, once per bound
            from .dist import packed_rows
            with torch.cuda.stream(self.stream):
                be.send[packed_rows(umax, self.D)[1] * self.D: be.chunk_rows(umax, self.dense.grad) * self.D].zero_()
            self.sync()
        graphs = []
        segs = None

        def drop(items):                       # a capture that failed half-way: the graphs instantiated so far are destroyed, not leaked
            for g in items:
                if isinstance(g, list):
                    drop([item for kind, item in g if kind == "graph"])
                elif g:
                    L.call("amid_graph_destroy", g)

        try:
            for part in (0, 1):
                L.call("amid_graph_capture_begin", self.s)
                try:
                    if part == 0:          # the tail of backward packs the chunk itself: no padding launch (amid_grad_tail_pack_f32)
                        send = be.send[: be.chunk_rows(umax, dgrad) * self.D]
                        self._tail_pack = (send, umax, in_chunk)
                        segs = [] if getattr(self, "_dp_mid_collectives", False) else None
                        self._seg_capture, self._seg_plan = segs, pl      # (engine._coll cuts the capture at every mid-step collective)
                        try:
                            self.enqueue_local_grads(pl)
                        finally:
                            self._tail_pack = None
                            self._seg_capture = self._seg_plan = None
                    else:
                        recv = be.gather_buffer(exchange.world, umax, dense=dgrad)
                        self.enqueue_optimizer_gathered(be, recv, exchange.world, umax, dense_in_chunk=in_chunk)
                finally:
                    out = ctypes.c_void_p()
                    L.call("amid_graph_capture_end", self.s, ctypes.byref(out))
                if part == 0 and segs is not None:
                    segs.append(("graph", out.value))
                    graphs.append(segs)
                    segs = None
                else:
                    graphs.append(out.value)
        except BaseException:
            drop(graphs + ([segs] if segs else []))
            self.step = step0
            raise
        self.step = step0                      # capture does not execute
        if not hasattr(pl, "dp_graphs"):
            pl.dp_graphs = {}
        pl.dp_graphs[(self._graph_key(), umax, dense, id(exchange))] = (graphs[0], graphs[1], send, recv)

    def merge_backend(self, capacity: int) -> "HipMergeBackend":
        return HipMergeBackend(self, capacity)


class HipMergeBackend:
    """Merges the world's (ids, rows) lists with the same sort-unique + segment-reduce kernels the
    local backward uses (amid_amd.dist.MergeBackend on the GPU)."""

    def __init__(self, eng: SasrecEngine, capacity: int):
        L = lib()
        self.eng, self.cap = eng, int(capacity)
        dev, D = eng.device, eng.D
        self.sort_ws = torch.zeros(L.value("amid_sort_unique_workspace_bytes", self.cap), dtype=torch.uint8, device=dev)
        self.pos_sorted = torch.zeros(self.cap, dtype=torch.int32, device=dev)
        self.uniq_ids = torch.zeros(self.cap, dtype=torch.int32, device=dev)
        self.seg_off = torch.zeros(self.cap + 1, dtype=torch.int32, device=dev)
        self.seg_of = torch.zeros(self.cap, dtype=torch.int32, device=dev)
        self.n_uniq = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seg_ws = torch.empty(L.value("amid_segreduce_workspace_bytes", self.cap, D), dtype=torch.uint8, device=dev)
        self.uniq_rows = torch.empty(self.cap, D, dtype=torch.float32, device=dev)
        # exchange buffers: this rank's packed chunk and the world's gathered chunks (sliced per step, never reallocated);
        # capacity counts entries, so the id rows of the packed layout come on top (dist.packed_rows)
        # (+ the flat dense gradient, which rides behind the rows in the one-collective step: up to MAX_WORLD chunks of it)
        self.dense_rows = (eng.dense.numel + D - 1) // D
        self.send = torch.zeros((self.cap + (self.cap + D - 1) // D + 16 + self.dense_rows) * D, dtype=torch.float32, device=dev)
        self.all = torch.zeros((self.cap + (self.cap + D - 1) // D + 16 * 16 + self.MAX_WORLD * self.dense_rows) * D, dtype=torch.float32,
                               device=dev)
        self._entries = {}
        # owner-bucketed exchange (dist.SparseDenseExchange._exchange_owner): per-owner counts (+ the fill's overflow flag)
        self.owner_ws = torch.empty(L.value("amid_owner_workspace_bytes", self.cap), dtype=torch.uint8, device=dev)
        self.owner_counts = torch.zeros(self.MAX_WORLD + 1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)

    MAX_WORLD = 16

    @property
    def capacity(self) -> int:
        return self.cap

    def bucket_counts(self, uniq_ids: torch.Tensor, n_uniq: torch.Tensor, world: int) -> torch.Tensor:
        """[world] int32: how many of this rank's unique ids each owner (id % world) gets; also prepares fill_buckets()."""
        lib().call("amid_owner_count_i32", uniq_ids.data_ptr(), n_uniq.data_ptr(), uniq_ids.numel(), world, self.owner_ws.data_ptr(),
                   self.owner_counts.data_ptr(), self.eng.s)
        return self.owner_counts[:world]

    def fill_buckets(self, uniq_ids: torch.Tensor, uniq_rows: torch.Tensor, n_uniq: torch.Tensor, world: int, bmax: int) -> torch.Tensor:
        """The stable split of (ids, rows) by owner into `world` packed chunks of bmax entries (after bucket_counts() on the same list)."""
        from .dist import packed_rows
        D = self.eng.D
        id_rows, rows = packed_rows(bmax, D)
        if world * rows * D > self.send.numel():
            raise ValueError(f"{world} buckets of {bmax} entries exceed the backend capacity {self.cap}")
        send = self.send[: world * rows * D]
        lib().call("amid_owner_buckets_f32", uniq_ids.data_ptr(), uniq_rows.data_ptr(), n_uniq.data_ptr(), uniq_ids.numel(), D, world, bmax,
                   self.eng.n_rows, self.owner_ws.data_ptr(), send.data_ptr(), rows * D, id_rows, self.owner_counts.data_ptr(), self.eng.s)
        return send

    def _entry(self, key, src: int, dst: int, stride: int, n_part: int, count: int) -> torch.Tensor:
        """A one-entry table for amid_reduce_partials_f32 (device resident, cached: captured graphs keep pointing at it)."""
        ent = self._entries.get(key)
        if ent is None:
            L = lib()
            host = (ctypes.c_ubyte * L.value("amid_reduce_entry_bytes"))()
            L.call("amid_reduce_entry_pack", ctypes.addressof(host), 0, src, dst, stride, n_part, count)
            ent = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(self.eng.device)
            torch.cuda.synchronize(self.eng.device)
            self._entries[key] = ent
        return ent

    def prepare_dense(self, world: int, umax: int, dense: torch.Tensor) -> None:
        """Build the reduce tables pad_packed(dense=...) / sum_dense() will use for this (world, umax): they copy to the device and
        synchronise, which is not allowed while a stream is being captured."""
        from .dist import packed_rows
        D = self.eng.D
        rows = packed_rows(umax, D)[1]
        self._entry(("copy", umax), dense.data_ptr(), self.send.data_ptr() + 4 * rows * D, 0, 1, dense.numel())
        self._entry(("sum", self.all.data_ptr(), world, umax), self.all.data_ptr() + 4 * rows * D, dense.data_ptr(),
                    self.chunk_rows(umax, dense) * D, world, dense.numel())

    def chunk_rows(self, umax: int, dense: Optional[torch.Tensor]) -> int:
        from .dist import packed_rows
        return packed_rows(umax, self.eng.D)[1] + (self.dense_rows if dense is not None else 0)

    def sum_dense(self, gathered: torch.Tensor, world: int, umax: int, dense: torch.Tensor) -> None:
        """dense <- sum over the ranks, in rank order, of the dense parts that travelled behind the sparse rows."""
        from .dist import packed_rows
        D = self.eng.D
        off = packed_rows(umax, D)[1] * D
        stride = self.chunk_rows(umax, dense) * D
        ent = self._entry(("sum", gathered.data_ptr(), world, umax), gathered.data_ptr() + 4 * off, dense.data_ptr(), stride, world, dense.numel())
        lib().call("amid_reduce_partials_f32", ent.data_ptr(), 1, dense.numel(), self.eng.s)

    def pad_packed(self, uniq_ids: torch.Tensor, uniq_rows: torch.Tensor, n_uniq: torch.Tensor, umax: int,
                   dense: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dense: also append this flat fp32 buffer (the dense gradient) behind the rows, so that ONE all-gather moves everything."""
        from .dist import packed_rows
        D = self.eng.D
        id_rows, rows = packed_rows(umax, D)
        send = self.send[: self.chunk_rows(umax, dense) * D]
        # sentinel padding (one past the last table row) keeps every rank's list sorted, so merge_packed() is a merge, not a sort
        if dense is not None:          # the copy of the flat dense gradient behind the rows rides in the padding launch
            ent = self._entry(("copy", umax), dense.data_ptr(), send.data_ptr() + 4 * rows * D, 0, 1, dense.numel())
            lib().call("amid_sparse_pad_sum_f32", uniq_ids.data_ptr(), uniq_rows.data_ptr(), n_uniq.data_ptr(), umax, D, self.eng.n_rows,
                       send.data_ptr(), send.data_ptr() + 4 * id_rows * D, ent.data_ptr(), 1, dense.numel(), self.eng.s)
        else:
            lib().call("amid_sparse_pad_f32", uniq_ids.data_ptr(), uniq_rows.data_ptr(), n_uniq.data_ptr(), umax, D, self.eng.n_rows,
                       send.data_ptr(), send.data_ptr() + 4 * id_rows * D, self.eng.s)
        return send

    def gather_buffer(self, world: int, umax: int, dense: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = world * self.chunk_rows(umax, dense) * self.eng.D
        if n > self.all.numel() or world * umax > self.cap or (dense is not None and world > self.MAX_WORLD):
            raise ValueError(f"gather of {world} x {umax} entries exceeds the backend capacity {self.cap}")
        return self.all[:n]

    def merge_packed(self, gathered: torch.Tensor, world: int, umax: int, dense: Optional[torch.Tensor] = None, sum_dense: bool = False):
        """`world` packed chunks (sorted, sentinel-padded ids + rows [+ a dense tail the merge skips]) -> 2-launch stable merge +
        segment reduce.  sum_dense: the rank-ordered sum of the dense tails into `dense` (what sum_dense() does) rides in the
        merge's first launch."""
        from .dist import packed_rows
        L, eng = lib(), self.eng
        D = eng.D
        id_rows, _ = packed_rows(umax, D)
        rows = self.chunk_rows(umax, dense)
        n = world * umax
        if sum_dense and dense is not None:
            off = packed_rows(umax, D)[1] * D
            ent = self._entry(("sum", gathered.data_ptr(), world, umax), gathered.data_ptr() + 4 * off, dense.data_ptr(), rows * D, world,
                              dense.numel())
            L.call("amid_merge_sorted_lists_sum_i32", gathered.data_ptr(), world, umax, rows * D, id_rows, rows, eng.n_rows,
                   self.sort_ws.data_ptr(), self.pos_sorted.data_ptr(), self.uniq_ids.data_ptr(), self.seg_off.data_ptr(),
                   self.seg_of.data_ptr(), self.n_uniq.data_ptr(), ent.data_ptr(), 1, dense.numel(), eng.s)
        else:
            L.call("amid_merge_sorted_lists_i32", gathered.data_ptr(), world, umax, rows * D, id_rows, rows, eng.n_rows, self.sort_ws.data_ptr(),
                   self.pos_sorted.data_ptr(), self.uniq_ids.data_ptr(), self.seg_off.data_ptr(), self.seg_of.data_ptr(),
                   self.n_uniq.data_ptr(), eng.s)
        L.call("amid_embgrad_segreduce_f32", gathered.data_ptr(), self.pos_sorted.data_ptr(), self.seg_off.data_ptr(), self.seg_of.data_ptr(),
               n, D, self.seg_ws.data_ptr(), self.uniq_rows.data_ptr(), eng.s)
        return self.uniq_ids[:n], self.uniq_rows[:n], self.n_uniq

    def merge(self, ids: torch.Tensor, rows: torch.Tensor):
        """Arbitrary (unsorted) ids -> full radix sort + segment reduce."""
        L, eng = lib(), self.eng
        n = ids.numel()
        if n > self.cap:
            raise ValueError(f"merge of {n} entries exceeds the backend capacity {self.cap}")
        L.call("amid_sort_unique_i32", ids.data_ptr(), n, eng.n_rows, self.sort_ws.data_ptr(), self.pos_sorted.data_ptr(),
               self.uniq_ids.data_ptr(), self.seg_off.data_ptr(), self.seg_of.data_ptr(), self.n_uniq.data_ptr(), eng.s)
        L.call("amid_embgrad_segreduce_f32", rows.data_ptr(), self.pos_sorted.data_ptr(), self.seg_off.data_ptr(), self.seg_of.data_ptr(),
               n, eng.D, self.seg_ws.data_ptr(), self.uniq_rows.data_ptr(), eng.s)
        return self.uniq_ids[:n], self.uniq_rows[:n], self.n_uniq
