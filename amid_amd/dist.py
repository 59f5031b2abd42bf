"""Data-parallel exchange for the SASRec step: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (``torch.nn.DataParallel`` is commented out, train_sr.py:473), so this
layer is new.  Every rank holds a full replica of the parameters (the table
is 458 MB -- trivial against 288 GB of HBM) and a shard of the global minibatch; per step there is
exactly one exchange, made of

  dense   one flat-buffer all-reduce(sum) of every non-table gradient (1.7 MB: latency-bound, so ONE
          call, never per-parameter buckets); averaging is folded into Adam's ``grad_scale``;
  sparse  ONE all-gather of each rank's segment-reduced (unique ids, gradient rows), packed into a single
          buffer (packed_rows): ids and rows are padded to the largest per-rank count with zero rows under a neutral id (the HIP backend uses a
          sentinel one past the table so that every list stays sorted; the torch backend repeats the first
          id, which adds exact zeros), then every rank merges the world's lists -- a stable rank-ordered
          merge of sorted lists (4 launches) + the segment-reduce kernel it used locally.  Same inputs,
          same fixed summation order => replicas stay bit-identical.

Device-specific work (merging) is delegated to a backend object so the protocol itself is covered
by world_size-2 gloo tests on CPU, with a test double in place of the HIP kernels.
"""
from __future__ import annotations

from typing import Optional, Protocol, Tuple

import torch
import torch.distributed as dist


def packed_rows(umax: int, D: int) -> Tuple[int, int]:
    """Layout of one rank's packed chunk, in rows of D floats: (id rows, total rows).  The chunk is [id rows | umax gradient
    rows]; the first umax int32 of the id rows are the ids (bit-cast), so ids and rows travel in ONE all-gather."""
    id_rows = (umax + D - 1) // D
    return id_rows, id_rows + umax


class MergeBackend(Protocol):
    def pad_packed(self, uniq_ids: torch.Tensor, uniq_rows: torch.Tensor, n_uniq: torch.Tensor, umax: int) -> torch.Tensor:
        """This rank's chunk [packed_rows(umax, D)[1] * D] float32: entries at and beyond n_uniq (device scalar) carry a neutral
        id (sentinel / repeated first id) and a zero row."""
        ...

    def gather_buffer(self, world: int, umax: int) -> torch.Tensor:
        """Receive buffer for the world's chunks, [world * chunk] float32."""
        ...

    def merge_packed(self, gathered: torch.Tensor, world: int, umax: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The world's chunks -> (uniq_ids [n] int32, uniq_rows [n, D], n_uniq [1] int32), n = world * umax; only the first
        n_uniq entries are meaningful; rows of equal ids summed in rank order."""
        ...


class SparseDenseExchange:
    def __init__(self, backend: MergeBackend, group=None, host_staging: bool = False, always: bool = False):
        """always: run the collectives and the merge even in a world of one (profiling the exchange path on a single GPU)."""
        self.always = always
        self.backend = backend
        self.group = group
        # host_staging: move collective payloads through host memory (for process groups without device
        # collectives, e.g. gloo with GPU tensors in the single-GPU two-process test); never used with RCCL
        self.host_staging = host_staging
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @property
    def grad_scale(self) -> float:
        """Each rank's loss is the mean over its own shard; the global-batch mean is the rank average."""
        return 1.0 / self.world

    @property
    def active(self) -> bool:
        return self.world > 1 or self.always

    def all_reduce_dense(self, flat_grad: torch.Tensor) -> None:
        if self.active:
            if self.host_staging and flat_grad.is_cuda:
                h = flat_grad.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat_grad.copy_(h)
            else:
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def all_reduce_max(self, t: torch.Tensor) -> None:
        """In-place MAX over the world (data-pipeline bookkeeping, e.g. the epoch's largest unique-row count)."""
        if self.active:
            if self.host_staging and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def all_gather_packed(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        if self.host_staging and send.is_cuda:
            h = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_gather_into_tensor(h, send.cpu(), group=self.group)
            recv.copy_(h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)

    def exchange_sparse(self, uniq_ids: torch.Tensor, uniq_rows: torch.Tensor, n_uniq: torch.Tensor, umax: Optional[int] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """uniq_ids [cap] int32, uniq_rows [cap, D], n_uniq [1] int32 (device) -> merged world lists.

        umax = the world's largest per-rank unique count for THIS step (or any bound on it) when the host already knows it (the
        data pipeline can count a batch's unique ids while packing it and max-reduce the counts ahead of time, as bench.py
        does): the exchange then needs no device -> host synchronisation.  With umax=None it costs one small host sync per step."""
        if not self.active:
            return uniq_ids, uniq_rows, n_uniq
        if umax is None:
            nmax = n_uniq.cpu() if self.host_staging else n_uniq.clone()
            dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=self.group)
            umax = int(nmax.item())
        umax = max(1, min(int(umax), uniq_ids.numel()))
        send = self.backend.pad_packed(uniq_ids, uniq_rows, n_uniq, umax)
        recv = self.backend.gather_buffer(self.world, umax)
        self.all_gather_packed(send, recv)                       # ids and rows of a rank travel together: ONE collective
        return self.backend.merge_packed(recv, self.world, umax)


class TorchMergeBackend:
    """Reference implementation of the backend protocol in plain torch (the test double of the gloo tests; also documents
    what the HIP backend must compute)."""

    def __init__(self, D: int, device="cpu"):
        self.D, self.device = D, device

    def pad_packed(self, uniq_ids, uniq_rows, n_uniq, umax):
        D = self.D
        id_rows, rows = packed_rows(umax, D)
        pad = torch.arange(umax, device=uniq_ids.device, dtype=torch.int32) >= n_uniq.to(torch.int32)
        ids = torch.where(pad, uniq_ids[0:1].expand(umax), uniq_ids[:umax])            # repeated first id: adds an exact 0.0
        out = torch.zeros(rows * D, dtype=torch.float32, device=uniq_rows.device)
        out[:umax] = ids.to(torch.int32).view(torch.float32)
        out[id_rows * D:] = (uniq_rows[:umax] * (~pad).unsqueeze(1).to(uniq_rows.dtype)).reshape(-1)
        return out

    def gather_buffer(self, world, umax):
        return torch.empty(world * packed_rows(umax, self.D)[1] * self.D, dtype=torch.float32, device=self.device)

    def merge_packed(self, gathered, world, umax):
        D = self.D
        id_rows, rows = packed_rows(umax, D)
        g = gathered.view(world, rows * D)
        ids = g[:, :umax].contiguous().view(torch.int32).reshape(-1)
        rws = g[:, id_rows * D:].reshape(world * umax, D)
        n = ids.numel()
        u, inv = torch.unique(ids.long(), return_inverse=True)        # sorted ascending, like the merge kernels
        out = torch.zeros(n, D, dtype=rws.dtype, device=rws.device)
        out.index_add_(0, inv, rws)
        uid = torch.zeros(n, dtype=torch.int32, device=ids.device)
        uid[: u.numel()] = u.to(torch.int32)
        return uid, out, torch.tensor([u.numel()], dtype=torch.int32, device=ids.device)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """DistributedSampler-style contiguous shard of a global batch (drop_last semantics of train_sr.py:452)."""
    B = next(iter(batch.values())).shape[0]
    per = B // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
