"""Batch inputs of the engine (split out of engine.py): single batches copied into the plan's static image, whole epochs packed once and
resident in HBM as an input pool that the step's first kernel walks by the device step counter.  Replaces the per-step
`.long()` / `.cuda()` marshalling of train_sr.py:191-200."""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from ._lib import lib, ptr_array

from .plan import SasrecPlan


class InputMixin:
    def load_batch(self, pl: SasrecPlan, i_node, neg_samples, seq_d1, seq_d2, labels=None, domain_id=None, ob_label=None) -> None:
        """Copy a batch into the plan's static input buffers (async on the engine stream)."""
        if self.input_pool(pl) is not None:
            raise RuntimeError("this plan reads its batches from an installed input pool (set_input_pool); drop the pool before loading single batches")
        with torch.cuda.stream(self.stream):
            pl.in_i_node.copy_(i_node.reshape(-1), non_blocking=True)
            pl.in_neg.copy_(neg_samples.reshape(pl.shape.B, -1), non_blocking=True)
            pl.in_seq_d1.copy_(seq_d1, non_blocking=True)
            pl.in_seq_d2.copy_(seq_d2, non_blocking=True)
            if labels is not None:
                pl.labels.copy_(labels.reshape(pl.shape.B, -1), non_blocking=True)
                pl.domain.copy_(domain_id.reshape(-1), non_blocking=True)
            if ob_label is not None:
                pl.in_ob.copy_(ob_label.reshape(-1), non_blocking=True)

    def pack_batch(self, pl: SasrecPlan, i_node, neg_samples, seq_d1, seq_d2, labels, domain_id) -> torch.Tensor:
        """Pre-pack a batch into the plan's input layout (one contiguous int64 tensor) for load_packed()."""
        B, NI = pl.shape.B, pl.shape.NI
        lab = torch.zeros(2 * ((B * NI + 1) // 2), dtype=torch.float32, device=labels.device)
        lab[: B * NI] = labels.reshape(-1).float()
        return torch.cat((i_node.reshape(-1).long(), neg_samples.reshape(-1).long(), seq_d1.reshape(-1).long(), seq_d2.reshape(-1).long(),
                          domain_id.reshape(-1).long(), lab.view(torch.int64))).contiguous()

    def pack_epoch(self, pl: SasrecPlan, i_node, neg_samples, seq_d1, seq_d2, labels, domain_id, ob_label=None) -> torch.Tensor:
        """pack_batch() for n batches at once: i_node [n, B], neg_samples [n, B, NI-1], seq_d* [n, B, T], domain_id [n, B] (ob_label
        [n, B] for isDR plans), labels [B, NI] shared by every batch (dataset_seq.py:191,199) -> [n, in_words] int64, the layout
        set_input_pool() takes.  A handful of device ops for a whole epoch instead of four copies per step."""
        B, NI = pl.shape.B, pl.shape.NI
        n = i_node.shape[0]
        lab = torch.zeros(2 * ((B * NI + 1) // 2), dtype=torch.float32, device=i_node.device)
        lab[: B * NI] = labels.reshape(-1).float()
        parts = [i_node.reshape(n, -1).long(), neg_samples.reshape(n, -1).long(), seq_d1.reshape(n, -1).long(), seq_d2.reshape(n, -1).long(),
                 domain_id.reshape(n, -1).long(), lab.view(torch.int64).unsqueeze(0).expand(n, -1)]
        if pl.in_ob is not None:
            parts.append((ob_label if ob_label is not None else torch.zeros_like(domain_id)).reshape(n, -1).long())
        out = torch.cat(parts, 1).contiguous()
        if out.shape[1] != pl.in_words:
            raise ValueError(f"packed row has {out.shape[1]} words, the plan expects {pl.in_words}")
        return out

    def set_input_pool(self, pl: SasrecPlan, pool: Optional[torch.Tensor]) -> None:
        """Make `pool` ([n_pool, in_words] int64, rows = pack_batch() images, resident in HBM) the plan's input: every following
        train step consumes the next row, chosen ON THE DEVICE by the step counter, so the replayed graph needs no per-step input
        copy (train_sr.py:185-199 moves each batch inside the loop).  None returns to load_batch()/load_packed().  Graphs of the
        plan are re-captured."""
        key = self._graph_key()                       # like the graphs, a pool belongs to (Adam state, objective): the DR trainer's
        if pool is not None:                          # two loops each keep their own (train_sr_dr.py:191-229 / :363-402)
            if pool.dtype != torch.int64 or pool.dim() != 2 or pool.shape[1] != pl.in_words or pool.stride(1) != 1 \
                    or pool.device != pl.in_pack.device:
                raise ValueError(f"input pool must be a device int64 [n, {pl.in_words}] tensor with contiguous rows")
            pl.pools[key] = [pool, (-self.step) % pool.shape[0]]
        else:
            pl.pools.pop(key, None)
        torch.cuda.synchronize(self.device)
        if getattr(pl, "graphs", None):               # the pool pointer is baked into captured launches
            pl.graphs.pop(key, None)
        if getattr(pl, "graphs_n", None):
            for k in [k for k in pl.graphs_n if k[0] == key]:
                pl.graphs_n.pop(k)
        if getattr(pl, "dp_graphs", None):
            pl.dp_graphs.clear()
        pl.graphs_local = {}

    def input_pool(self, pl: SasrecPlan):
        """[pool, phase] installed for the current (Adam state, objective), or None."""
        return pl.pools.get(self._graph_key()) if getattr(pl, "pools", None) else None

    def refill_input_pool(self, pl: SasrecPlan, pool: torch.Tensor) -> bool:
        """Overwrite the installed pool's contents with `pool` (next epoch) keeping every captured graph; possible when the shape is
        unchanged and the step counter sits on a pool boundary (the phase baked into the captured launch still holds).  False:
        the caller installs the new pool with set_input_pool()."""
        ent = self.input_pool(pl)
        if ent is None or ent[0].shape != pool.shape or (-self.step) % pool.shape[0] != ent[1]:
            return False
        with torch.cuda.stream(self.stream):
            ent[0].copy_(pool, non_blocking=True)
        return True

    def load_packed(self, pl: SasrecPlan, packed: torch.Tensor) -> None:
        if self.input_pool(pl) is not None:
            raise RuntimeError("this plan reads its batches from an installed input pool (set_input_pool); drop the pool before loading single batches")
        with torch.cuda.stream(self.stream):
            pl.in_pack.copy_(packed, non_blocking=True)
