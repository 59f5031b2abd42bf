"""world_size-2 / -4 gloo tests (CPU) of the data-parallel exchange protocol in amid_amd/dist.py.
The HIP merge kernels are replaced by a torch test double; the protocol (flat dense all-reduce,
padded sparse all-gather or the owner-bucketed all-to-all + all-gather, merge, 1/world scaling) is the code under test."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import amid_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, host_knows_umax, owner=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, TorchMergeBackend, shard_batch
        torch.manual_seed(0)
        n_items, D, T, hid, B = 120, 16, 10, 8, 8
        P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=3)
        batch = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=5)
        # owner: the owner-bucketed exchange (split by id % world, all-to-all, owner merge, all-gather) whatever the size
        ex = SparseDenseExchange(TorchMergeBackend(D), owner_threshold=0 if owner else None)
        assert ex.world == world and ex.rank == rank and ex.grad_scale == 1.0 / world
        local = shard_batch(batch, rank, world)
        _, _, g = orc.loss_and_grads("sasrec", P, local, None)
        names = [k for k in g if k != "item_emb_layer.emb_item.weight"]
        flat = torch.cat([g[k].reshape(-1) for k in names])
        tg = g["item_emb_layer.emb_item.weight"]
        touched = torch.unique(torch.cat([local[k].reshape(-1) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2")]))
        cap = 2 * local["seq_d1"].numel() + 2 * local["i_node"].numel()
        ids = torch.zeros(cap, dtype=torch.int32)
        rows = torch.zeros(cap, D)
        ids[: touched.numel()] = touched.to(torch.int32)
        rows[: touched.numel()] = tg[touched]
        rows[touched.numel():] = 777.0          # garbage beyond n_uniq must never leak into the merge
        nu = torch.tensor([touched.numel()], dtype=torch.int32)
        ex.all_reduce_dense(flat)
        umax = None
        if host_knows_umax:                      # the data pipeline counted the uniques and max-reduced them ahead of time
            cnt = torch.tensor([touched.numel()])
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            umax = int(cnt)
        mid, mrows, mnu = ex.exchange_sparse(ids, rows, nu, umax=umax)
        U = int(mnu.item())
        table_grad = torch.zeros(n_items, D)
        table_grad[mid[:U].long()] = mrows[:U]
        assert ex.stats["owner_steps" if owner else "gather_steps"] == 1 and ex.stats["gather_steps" if owner else "owner_steps"] == 0
        assert ex.stats["collectives"] == (3 if owner else 2)          # dense all-reduce + (all-to-all, all-gather | all-gather)
        q.put((rank, (flat * ex.grad_scale).numpy(), (table_grad * ex.grad_scale).numpy(), int(nu.item()), U))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,host_knows_umax,owner", [(2, False, False), (2, True, False), (2, False, True), (2, True, True),
                                                         (4, False, True), (4, True, False)])
@pytest.mark.timeout(300)
def test_exchange_matches_single_process_global_batch(world, host_knows_umax, owner):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, host_knows_umax, owner)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n_items, D, T, hid, B = 120, 16, 10, 8, 8
    P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=3)
    batch = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=5)
    _, _, g = orc.loss_and_grads("sasrec", P, batch, None)       # single process, global batch, global mean
    names = [k for k in g if k != "item_emb_layer.emb_item.weight"]
    want_flat = torch.cat([g[k].reshape(-1) for k in names])
    want_tab = g["item_emb_layer.emb_item.weight"]
    outs = [(r, torch.from_numpy(f), torch.from_numpy(t), nu, U) for r, f, t, nu, U in outs]
    for rank, flat, tab, nu, U in outs:
        assert float((flat - want_flat).abs().max()) < 1e-6
        assert float((tab - want_tab).abs().max()) < 1e-6
        assert U >= nu
    for o in outs[1:]:
        assert torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2], o[2])    # replicas identical


def test_exchange_world1_is_passthrough():
    from amid_amd.dist import SparseDenseExchange, TorchMergeBackend
    ex = SparseDenseExchange(TorchMergeBackend(8))
    assert ex.world == 1 and ex.grad_scale == 1.0
    ids, rows, nu = torch.arange(4, dtype=torch.int32), torch.randn(4, 8), torch.tensor([3], dtype=torch.int32)
    a, b, c = ex.exchange_sparse(ids, rows, nu)
    assert a is ids and b is rows and c is nu
    f = torch.randn(5)
    g = f.clone()
    ex.all_reduce_dense(g)
    assert torch.equal(f, g)


def test_torch_backend_buckets_are_a_stable_split():
    from amid_amd.dist import TorchMergeBackend, packed_rows
    D, W = 8, 4
    be = TorchMergeBackend(D)
    ids = torch.tensor([1, 2, 4, 5, 8, 9, 13, 21, 0, 0], dtype=torch.int32)
    rows = torch.arange(10 * D, dtype=torch.float32).reshape(10, D)
    nu = torch.tensor([8], dtype=torch.int32)
    cnt = be.bucket_counts(ids, nu, W)
    assert cnt.tolist() == [2, 5, 1, 0]
    bmax = 6
    out = be.fill_buckets(ids, rows, nu, W, bmax).view(W, -1)
    id_rows, tot = packed_rows(bmax, D)
    got = out[:, :bmax].contiguous().view(torch.int32)
    assert got[0, :2].tolist() == [4, 8] and got[1, :5].tolist() == [1, 5, 9, 13, 21] and got[2, :1].tolist() == [2]
    r = out[:, id_rows * D:].reshape(W, bmax, D)
    assert torch.equal(r[1, :5], rows[[0, 3, 5, 6, 7]]) and float(r[1, 5:].abs().sum()) == 0.0 and float(r[3].abs().sum()) == 0.0
