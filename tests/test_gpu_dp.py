"""Data-parallel step on the GPU engine: two processes share the one test GPU (gloo group, payloads staged through the
host because gloo has no device all-gather); parameters after K steps must equal the oracle's dense-Adam run on the
global batch.  Exercises SasrecEngine.train_step_dp, HipMergeBackend and the sparse merge with real kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import amid_oracle as orc

pytestmark = pytest.mark.gpu
CFG = dict(n_items=160, D=64, T=20, hid=16, B=8, K=3, seed=11, lr=1e-3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches():
    c = CFG
    return [orc.synthetic_batch(c["B"], c["T"], c["n_items"] - 1, pad_id=c["n_items"] - 1, neg=1, seed=500 + t) for t in range(c["K"])]


def _worker(rank, world, port, use_graph, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, shard_batch
        from amid_amd.engine import SasrecEngine
        c = CFG
        torch.cuda.set_device(0)
        P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"]), seed=7)
        eng = SasrecEngine(c["n_items"], c["D"], c["T"], c["hid"], device="cuda:0", lr=c["lr"], seed=c["seed"])
        eng.load_state_dict(P)
        Bl = c["B"] // world
        pl = eng.plan(Bl, c["T"], 2, need_grad=True)
        ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=True)
        first = True
        for batch in _batches():
            local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
            eng.load_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"])
            if use_graph and first:
                eng.capture_local_grads(pl)
                first = False
            eng.train_step_dp(pl, ex, use_graph=use_graph)
            eng.sync()
        eng.flush_table()
        eng.sync()
        q.put((rank, {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}, float(pl.loss.item())))   # by value
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.timeout(600)
def test_two_rank_dp_matches_global_batch_oracle(use_graph):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_graph, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=500) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = CFG
    P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"]), seed=7)
    opt = orc.DenseAdam(P, lr=c["lr"])
    Bl = c["B"] // world
    for t, batch in enumerate(_batches(), start=1):
        local_masks = orc.philox_masks_sasrec(Bl, c["T"], c["D"], seed=c["seed"], step=t)       # every rank: same seed, local row indices
        masks = {k: torch.cat([v] * world, 0) for k, v in local_masks.items()}
        orc.train_step("sasrec", P, opt, batch, masks)
    sd0, sd1 = ({k: torch.from_numpy(v) for k, v in o[1].items()} for o in outs)
    for k, v in P.items():
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged on {k}"
        d = (sd0[k] - v).abs()
        if k.endswith("in_proj_bias"):
            n = v.numel() // 3
            d = torch.cat((d[:n], d[2 * n:]))
        assert float(d.max()) < 1e-4, k
