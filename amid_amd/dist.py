"""Data-parallel exchange for the SASRec step: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (``torch.nn.DataParallel`` is commented out, train_sr.py:473), so this
layer is new.  Every rank holds a full replica of the parameters (the table
is 458 MB -- trivial against 288 GB of HBM) and a shard of the global minibatch; per step there is
exactly one exchange, made of

  dense   one flat-buffer all-reduce(sum) of every non-table gradient (1.7 MB: latency-bound, so ONE
          call, never per-parameter buckets); averaging is folded into Adam's ``grad_scale``;
  sparse  all-gather of each rank's segment-reduced (unique ids, gradient rows): ids and rows are
          padded to the largest per-rank count with zero rows under a neutral id (the HIP backend uses a
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


class MergeBackend(Protocol):
    def pad(self, uniq_ids: torch.Tensor, uniq_rows: torch.Tensor, n_uniq: torch.Tensor, umax: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """First umax entries of a rank's lists; entries at and beyond n_uniq (device scalar) become (uniq_ids[0], zero row)."""
        ...

    def gather_buffers(self, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Receive buffers (ids [n] int32, rows [n, D]) for the all-gather."""
        ...

    def merge(self, ids: torch.Tensor, rows: torch.Tensor, world: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """ids [n] int32 (duplicates allowed), rows [n, D] -> (uniq_ids [n], uniq_rows [n, D], n_uniq [1] int32);
        only the first n_uniq entries of the outputs are meaningful."""
        ...


class SparseDenseExchange:
    def __init__(self, backend: MergeBackend, group=None, host_staging: bool = False, always: bool = False):
        """always: run the collectives and the merge even in a world of one (profiling the exchange path on a single GPU)."""
        self.always = always
        self.backend = backend
        self.group = group
        # host_staging: move collective payloads through pinned host memory (for process groups without device
        # collectives, e.g. gloo with GPU tensors in the single-GPU two-process test); never used with RCCL
        self.host_staging = host_staging
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    @property
    def grad_scale(self) -> float:
        """Each rank's loss is the mean over its own shard; the global-batch mean is the rank average."""
        return 1.0 / self.world

    def all_reduce_dense(self, flat_grad: torch.Tensor) -> None:
        if self.world > 1 or self.always:
            if self.host_staging and flat_grad.is_cuda:
                h = flat_grad.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat_grad.copy_(h)
            else:
                dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)

    def exchange_sparse(self, uniq_ids: torch.Tensor, uniq_rows: torch.Tensor, n_uniq: torch.Tensor, umax: Optional[int] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """uniq_ids [cap] int32, uniq_rows [cap, D], n_uniq [1] int32 (device) -> merged world lists.

        umax = the world's largest per-rank unique count for THIS step when the host already knows it (the data pipeline can
        count a batch's unique ids while packing it and max-reduce the counts ahead of time, as bench.py does): the exchange
        then needs no device -> host synchronisation.  With umax=None it costs one small host sync per step."""
        if self.world == 1 and not self.always:
            return uniq_ids, uniq_rows, n_uniq
        if umax is None:
            nmax = n_uniq.cpu() if self.host_staging else n_uniq.clone()
            dist.all_reduce(nmax, op=dist.ReduceOp.MAX, group=self.group)
            umax = int(nmax.item())
        umax = max(1, min(int(umax), uniq_ids.numel()))
        ids, rows = self.backend.pad(uniq_ids, uniq_rows, n_uniq, umax)      # neutral padding: zero rows under a repeated / sentinel id
        all_ids, all_rows = self.backend.gather_buffers(self.world * umax)
        if self.host_staging and ids.is_cuda:
            h_ids, h_rows = torch.empty(all_ids.shape, dtype=ids.dtype), torch.empty(all_rows.shape, dtype=rows.dtype)
            dist.all_gather_into_tensor(h_ids, ids.cpu(), group=self.group)
            dist.all_gather_into_tensor(h_rows, rows.cpu(), group=self.group)
            all_ids.copy_(h_ids)
            all_rows.copy_(h_rows)
        else:
            dist.all_gather_into_tensor(all_ids, ids, group=self.group)
            dist.all_gather_into_tensor(all_rows, rows, group=self.group)
        return self.backend.merge(all_ids, all_rows, self.world)


class TorchMergeBackend:
    """Reference implementation of the backend protocol in plain torch (the test double of the gloo tests; also documents
    what the HIP backend must compute)."""

    def __init__(self, D: int, device="cpu"):
        self.D, self.device = D, device

    def pad(self, uniq_ids, uniq_rows, n_uniq, umax):
        ids = uniq_ids[:umax].clone()
        rows = uniq_rows[:umax].clone()
        pad = torch.arange(umax, device=ids.device, dtype=torch.int32) >= n_uniq.to(torch.int32)
        ids = torch.where(pad, uniq_ids[0:1].expand(umax), ids)
        rows = rows * (~pad).unsqueeze(1).to(rows.dtype)
        return ids, rows

    def gather_buffers(self, n):
        return torch.empty(n, dtype=torch.int32, device=self.device), torch.empty(n, self.D, dtype=torch.float32, device=self.device)

    def merge(self, ids, rows, world=0):
        n = ids.numel()
        u, inv = torch.unique(ids.long(), return_inverse=True)        # sorted ascending, like the radix sort
        out = torch.zeros(n, rows.shape[1], dtype=rows.dtype, device=rows.device)
        out.index_add_(0, inv, rows)
        uid = torch.zeros(n, dtype=torch.int32, device=ids.device)
        uid[: u.numel()] = u.to(torch.int32)
        return uid, out, torch.tensor([u.numel()], dtype=torch.int32, device=ids.device)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    """DistributedSampler-style contiguous shard of a global batch (drop_last semantics of train_sr.py:452)."""
    B = next(iter(batch.values())).shape[0]
    per = B // world
    return {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
