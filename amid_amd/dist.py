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

  sparse, owner-bucketed (SURVEY.md section 8(e)'s alternative, chosen when the all-gather would move more than
          `owner_threshold` bytes per rank, i.e. world * umax * D * 4 -- the cfg 5 regime of ~400 k unique rows per rank):
          every rank splits its list by owner = id % world into `world` buckets, ONE all-to-all hands each owner its buckets, the
          owner merges them (same merge + segment-reduce kernels, rank order), and ONE all-gather of the reduced lists gives every
          rank the world's gradient rows -- each id once, however many ranks touched it.  Bucket and reduced-list capacities are
          sized from the actual counts (two scalar max-reductions per step), so this path runs eagerly, not from a captured graph:
          at the sizes that select it a step is milliseconds long.

Device-specific work (merging, bucketing) is delegated to a backend object so the protocol itself is covered
by world_size-2 / -4 gloo tests on CPU, with a test double in place of the HIP kernels.
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


    # optional, for the owner-bucketed exchange:
    #   bucket_counts(uniq_ids, n_uniq, world) -> [world] int tensor: entries per owner (id % world)
    #   fill_buckets(uniq_ids, uniq_rows, n_uniq, world, bmax) -> [world * chunk(bmax)] float32: the stable split, one packed chunk
    #       per owner (same layout as pad_packed's), unused slots neutral
    #   capacity -> int: the largest world * len the merge accepts


OWNER_THRESHOLD_BYTES = 64 << 20


def round_up(n: int, m: int) -> int:
    return (int(n) + m - 1) // m * m


class SparseDenseExchange:
    def __init__(self, backend: MergeBackend, group=None, host_staging: bool = False, always: bool = False,
                 owner_threshold: Optional[int] = OWNER_THRESHOLD_BYTES):
        """always: run the collectives and the merge even in a world of one (profiling the exchange path on a single GPU).
        owner_threshold: bytes per rank an all-gather of the sparse lists may move (world * umax * D * 4) before the owner-bucketed
        exchange takes over; None = never, 0 = always (tests)."""
        self.always = always
        self.owner_threshold = owner_threshold
        self.stats = {"collectives": 0, "bytes_out": 0, "bytes_in": 0, "owner_steps": 0, "gather_steps": 0}
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

    def _count(self, out_bytes: int, in_bytes: int) -> None:
        self.stats["collectives"] += 1
        self.stats["bytes_out"] += int(out_bytes)
        self.stats["bytes_in"] += int(in_bytes)

    def all_reduce_dense(self, flat_grad: torch.Tensor) -> None:
        if self.active:
            self._count(flat_grad.numel() * 4, flat_grad.numel() * 4)
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

    def all_to_all_packed(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        """send = world equal chunks, chunk r for rank r; recv = world chunks, chunk r from rank r."""
        self._count(send.numel() * 4, recv.numel() * 4)
        if self.host_staging and send.is_cuda:
            h = torch.empty(recv.shape, dtype=recv.dtype)
            dist.all_to_all_single(h, send.cpu(), group=self.group)
            recv.copy_(h)
        else:
            dist.all_to_all_single(recv, send, group=self.group)

    def _max_over_world(self, t: torch.Tensor) -> int:
        """max over the ranks of a device scalar; one small host synchronisation."""
        h = t.detach().reshape(1).to("cpu" if self.host_staging else t.device, torch.int64)
        if self.world > 1:
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
        return int(h.item())

    def use_owner(self, umax: Optional[int], D: int) -> bool:
        """Whether a step whose largest per-rank unique count is `umax` takes the owner-bucketed exchange."""
        return (self.active and self.owner_threshold is not None and umax is not None and hasattr(self.backend, "fill_buckets")
                and self.world * int(umax) * D * 4 > self.owner_threshold)

    def all_gather_packed(self, send: torch.Tensor, recv: torch.Tensor) -> None:
        self._count(send.numel() * 4, recv.numel() * 4)
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
        if self.use_owner(umax, uniq_rows.shape[1]):
            out = self._exchange_owner(uniq_ids, uniq_rows, n_uniq)
            if out is not None:
                self.stats["owner_steps"] += 1
                return out
        self.stats["gather_steps"] += 1
        send = self.backend.pad_packed(uniq_ids, uniq_rows, n_uniq, umax)
        recv = self.backend.gather_buffer(self.world, umax)
        self.all_gather_packed(send, recv)                       # ids and rows of a rank travel together: ONE collective
        return self.backend.merge_packed(recv, self.world, umax)

    def _exchange_owner(self, uniq_ids, uniq_rows, n_uniq):
        """Owner-bucketed exchange: split by id % world -> all-to-all -> owner merge -> all-gather of the reduced lists -> merge
        (the reduced lists are disjoint, so the last merge only interleaves them).  None when a reduced list is too long for the
        backend (ids crowding on one owner): the caller falls back to the plain all-gather; the decision is taken on world-reduced
        counts, so every rank takes it alike."""
        be, W = self.backend, self.world
        bmax = round_up(max(1, self._max_over_world(be.bucket_counts(uniq_ids, n_uniq, W).max())), 64)
        send = be.fill_buckets(uniq_ids, uniq_rows, n_uniq, W, bmax)
        recv = be.gather_buffer(W, bmax)
        self.all_to_all_packed(send, recv)
        ids_o, rows_o, n_o = be.merge_packed(recv, W, bmax)                     # this owner's ids, each once, rows summed in rank order
        rmax = min(round_up(max(1, self._max_over_world(n_o)), 64), ids_o.numel())
        if W * rmax > be.capacity:
            return None
        send2 = be.pad_packed(ids_o, rows_o, n_o, rmax)
        recv2 = be.gather_buffer(W, rmax)
        self.all_gather_packed(send2, recv2)
        return be.merge_packed(recv2, W, rmax)


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

    capacity = 1 << 30

    def bucket_counts(self, uniq_ids, n_uniq, world):
        n = int(n_uniq)
        return torch.bincount(uniq_ids[:n].long() % world, minlength=world)

    def fill_buckets(self, uniq_ids, uniq_rows, n_uniq, world, bmax):
        """One pad_packed() chunk per owner; the boolean selection keeps the ascending order (stable split)."""
        n = int(n_uniq)
        ids, rws = uniq_ids[:n], uniq_rows[:n]
        chunks = []
        for o in range(world):
            sel = (ids.long() % world) == o
            k = int(sel.sum())
            bi = torch.cat((ids[sel], uniq_ids[0:1].expand(max(bmax - k, 0))))            # neutral padding as in pad_packed
            br = torch.cat((rws[sel], torch.zeros(max(bmax - k, 0), self.D, dtype=rws.dtype, device=rws.device)))
            chunks.append(self.pad_packed(bi, br, torch.tensor([k], dtype=torch.int32), bmax))
        return torch.cat(chunks)

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
