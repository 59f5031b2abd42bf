"""hipGraph capture / replay of the train step and state snapshots (split out of engine.py)."""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import lib, ptr_array

from .plan import SasrecPlan


class GraphMixin:
    def capture_local_grads(self, pl: SasrecPlan) -> None:
        L = lib()
        self._ensure_opt_state()
        saved = self.snapshot()
        self.enqueue_local_grads(pl)
        self.sync()
        self.restore(saved)
        self.sync()
        step0 = self.step
        L.call("amid_graph_capture_begin", self.s)
        try:
            self.enqueue_local_grads(pl)
        finally:
            out = ctypes.c_void_p()
            L.call("amid_graph_capture_end", self.s, ctypes.byref(out))
        self.step = step0
        if not isinstance(getattr(pl, "graphs_local", None), dict):
            pl.graphs_local = {}
        pl.graphs_local[self._graph_key()] = out.value      # (keyed like the train-step graphs: the Adam state and the DR objective are baked in)

    def has_local_graph(self, pl: SasrecPlan) -> bool:
        return self._graph_key() in (getattr(pl, "graphs_local", None) or {})


    # ------------------------------------------------------------------ graph replay
    def capture_train_step(self, pl: SasrecPlan) -> None:
        """Capture one train step into a hipGraph (inputs = the plan's static buffers)."""
        L = lib()
        self._ensure_opt_state()
        # warm-up outside capture: sets the dynamic-LDS attributes, pages code objects in
        saved = self.snapshot()
        self.enqueue_train_step(pl)
        self.sync()
        self.restore(saved)
        self.sync()
        s = self.s
        step0 = self.step
        L.call("amid_graph_capture_begin", s)
        try:
            self.enqueue_train_step(pl)
        finally:
            out = ctypes.c_void_p()
            L.call("amid_graph_capture_end", s, ctypes.byref(out))
        self.step = step0          # capture does not execute; the device counter did not move
        if not hasattr(pl, "graphs"):
            pl.graphs = {}
        pl.graphs[self._graph_key()] = out.value       # the Adam state's buffers and the DR objective are baked into a graph
        pl.graph = pl.graphs.get((0, 0), out.value)

    def capture_train_steps(self, pl: SasrecPlan, n_steps: int) -> None:
        """n_steps consecutive train steps as ONE hipGraph (replay_train_steps).  Only with an input pool: every step's first kernel
        picks its batch by the device step counter, so the steps of a graph see consecutive batches; a replayed graph costs ~8 us of
        idle device time between two launches, which a graph of several steps pays once."""
        if self.input_pool(pl) is None:
            raise ValueError("a graph of several train steps needs an input pool (set_input_pool)")
        L = lib()
        if not self.has_graph(pl):
            self.capture_train_step(pl)              # (also the warm-up outside capture)
        s, step0 = self.s, self.step
        L.call("amid_graph_capture_begin", s)
        try:
            for _ in range(n_steps):
                self.enqueue_train_step(pl)
        finally:
            out = ctypes.c_void_p()
            L.call("amid_graph_capture_end", s, ctypes.byref(out))
        self.step = step0
        if not hasattr(pl, "graphs_n"):
            pl.graphs_n = {}
        pl.graphs_n[(self._graph_key(), n_steps)] = out.value

    def replay_train_steps(self, pl: SasrecPlan, n_steps: int) -> None:
        lib().call("amid_graph_launch", pl.graphs_n[(self._graph_key(), n_steps)], self.s)
        self.step += n_steps

    def has_graph(self, pl: SasrecPlan) -> bool:
        return self._graph_key() in getattr(pl, "graphs", {})

    def replay_train_step(self, pl: SasrecPlan) -> None:
        lib().call("amid_graph_launch", pl.graphs[self._graph_key()], self.s)
        self.step += 1

    def flush_table(self) -> None:
        """Apply every pending zero-gradient Adam step (before eval / checkpoint / parity dumps)."""
        if self.table_m is None:
            return
        lib().call("amid_lazy_adam_flush_f32", self.table.data_ptr(), self.table_m.data_ptr(), self.table_v.data_ptr(),
                   self.table_last.data_ptr(), self.n_rows, self.D, self.step_state.data_ptr(), self.s)

    # ------------------------------------------------------------------ snapshots (tests / warm-up)
    def snapshot(self):
        self._ensure_opt_state()
        self.sync()
        fp = self.dense
        # the copies are enqueued on the ENGINE's stream: on torch's they were unordered against the step the caller enqueues next (the
        # warm-up of capture_train_step) -- with a table of a few GB the copy was still running when that step's optimizer moved rows, and
        # restore() put a half-stepped table back (found by test_folded_step_matches_the_fifteen_launch_step at 4.3 M rows, round 6)
        with torch.cuda.stream(self.stream):
            return dict(step=self.step, seed=self.seed, data=fp.data.clone(), m=fp.m.clone(), v=fp.v.clone(), table=self.table.clone(),
                        tm=self.table_m.clone(), tv=self.table_v.clone(), tl=self.table_last.clone())

    def restore(self, snap) -> None:
        fp = self.dense
        with torch.cuda.stream(self.stream):
            fp.data.copy_(snap["data"]); fp.m.copy_(snap["m"]); fp.v.copy_(snap["v"])
            self.table.copy_(snap["table"]); self.table_m.copy_(snap["tm"]); self.table_v.copy_(snap["tv"]); self.table_last.copy_(snap["tl"])
        self.set_step(snap["step"], snap["seed"])
