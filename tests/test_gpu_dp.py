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


def _get(q, procs, timeout=500):
    """q.get() that gives up as soon as a worker has died instead of waiting out the whole timeout."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                raise AssertionError("a worker process failed: " + str([p.exitcode for p in procs]))
            if time.time() - t0 > timeout:
                raise


def _batches():
    c = CFG
    return [orc.synthetic_batch(c["B"], c["T"], c["n_items"] - 1, pad_id=c["n_items"] - 1, neg=1, seed=500 + t) for t in range(c["K"])]


def _worker(rank, world, port, use_graph, q, host_knows_umax=False, pool=False, owner=False, backend="gloo", dense="gather"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    devi = rank if backend == "nccl" else 0               # RCCL refuses two ranks on one device ("Duplicate GPU detected")
    if backend == "nccl":
        torch.cuda.set_device(devi)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", devi))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, shard_batch
        from amid_amd.engine import SasrecEngine
        c = CFG
        torch.cuda.set_device(devi)
        P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"]), seed=7)
        eng = SasrecEngine(c["n_items"], c["D"], c["T"], c["hid"], device=f"cuda:{devi}", lr=c["lr"], seed=SasrecEngine.rank_seed(c["seed"], rank))
        eng.load_state_dict(P)
        Bl = c["B"] // world
        pl = eng.plan(Bl, c["T"], 2, need_grad=True)
        ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=backend != "nccl", owner_threshold=0 if owner else None)
        first = True
        if pool:                 # the rank's shards of all K batches resident in HBM; the step picks its batch by the device step counter
            packed = []
            for batch in _batches():
                local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
                packed.append(eng.pack_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"],
                                             local["domain_id"]))
            eng.set_input_pool(pl, torch.stack(packed))
        for batch in _batches():
            local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
            if not pool:
                eng.load_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"])
            if use_graph and first:
                eng.capture_local_grads(pl)
                first = False
            umax = None
            if host_knows_umax:       # what a data pipeline does while packing: count the uniques, max-reduce ahead of the step
                cnt = torch.tensor([int(torch.unique(torch.cat([local[k].reshape(-1) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2")])).numel())])
                cnt = cnt.cuda() if backend == "nccl" else cnt
                dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
                umax = (int(cnt) + 63) // 64 * 64             # a bucketed bound: the graph pair of the exchange is reused across steps
            eng.train_step_dp(pl, ex, use_graph=use_graph, umax=umax, dense=dense)
            eng.sync()
        eng.flush_table()
        eng.sync()
        assert ex.stats["owner_steps"] == (c["K"] if owner else 0) and (not owner or ex.stats["gather_steps"] == 0)
        assert int(ex.backend.owner_counts[world].item()) == 0          # no bucket overflowed
        q.put((rank, {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}, float(pl.loss.item())))   # by value
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_graph,host_knows_umax,pool,owner,dense", [(False, False, False, False, "gather"), (True, False, False, False, "gather"),
                                                                         (True, True, False, False, "gather"), (True, True, True, False, "gather"),
                                                                         (False, False, False, True, "gather"), (True, True, True, True, "gather"),
                                                                         (True, True, True, False, "allreduce"), (True, True, False, False, "allreduce")])
@pytest.mark.timeout(600)
def test_two_rank_dp_matches_global_batch_oracle(use_graph, host_knows_umax, pool, owner, dense):
    """owner: the owner-bucketed sparse exchange (amid_owner_count_i32 / amid_owner_buckets_f32 + all-to-all + all-gather).
    dense: how the flat dense gradient crosses the ranks in the graph-pair step (SasrecEngine.DENSE_EXCHANGE): behind the sparse rows
    in the one all-gather, or as its own all-reduce."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, use_graph, q, host_knows_umax, pool, owner, "gloo", dense)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = CFG
    P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"]), seed=7)
    opt = orc.DenseAdam(P, lr=c["lr"])
    Bl = c["B"] // world
    for t, batch in enumerate(_batches(), start=1):
        # every rank draws its own dropout stream (SasrecEngine.rank_seed) over its local row indices
        from amid_amd.engine import SasrecEngine
        per_rank = [orc.philox_masks_sasrec(Bl, c["T"], c["D"], seed=SasrecEngine.rank_seed(c["seed"], r), step=t) for r in range(world)]
        masks = {k: torch.cat([m[k] for m in per_rank], 0) for k in per_rank[0]}
        orc.train_step("sasrec", P, opt, batch, masks)
    sd0, sd1 = ({k: torch.from_numpy(v) for k, v in o[1].items()} for o in outs)
    for k, v in P.items():
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged on {k}"
        d = (sd0[k] - v).abs()
        if k.endswith("in_proj_bias"):
            n = v.numel() // 3
            d = torch.cat((d[:n], d[2 * n:]))
        assert float(d.max()) < 1e-4, k


@pytest.mark.parametrize("owner,dense", [(False, "gather"), (True, "gather"), (False, "allreduce")])
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL refuses two ranks on one device (ncclInvalidUsage: Duplicate GPU detected; "
                    "profiles/tools/probe/nccl_two_ranks_one_gpu.py), so the nccl leg needs two GPUs; the single-GPU box runs the gloo legs")
@pytest.mark.timeout(600)
def test_two_rank_dp_over_rccl_two_gpus(owner, dense):
    """The same two-rank step with the production backend: RCCL ("nccl"), one GPU per rank, device collectives, no host staging."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, True, q, True, True, owner, "nccl", dense)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in outs[0][1]:
        assert (outs[0][1][k] == outs[1][1][k]).all(), f"replicas diverged on {k}"


# ---- the folded step under data parallel (round 6): D 128, T 50, an input pool -- the shape family the single-GPU eleven-launch step covers ----
FOLD = dict(n_items=600, D=128, T=50, hid=32, B=24, K=4, seed=23, lr=1e-3)


def _fold_batches(T=50):
    c = dict(FOLD, T=T)
    out = []
    for t in range(3):            # a pool of three batches walked K = 4 times round: rows lag and come back
        b = orc.synthetic_batch(c["B"], c["T"], c["n_items"] - 1, pad_id=c["n_items"] - 1, neg=1, seed=800 + t)
        out.append(b)
    return out


def _fold_worker(rank, world, port, q, fused, mode, T=50):
    """mode: "pair" = graph A | all-gather | graph B (bench.py's path), "local" = the local-gradients graph + eager exchange, "eager"."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, shard_batch
        from amid_amd.engine import SasrecEngine
        c = dict(FOLD, T=T)
        torch.cuda.set_device(0)
        P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"]), seed=9)
        eng = SasrecEngine(c["n_items"], c["D"], c["T"], c["hid"], device="cuda:0", lr=c["lr"], seed=SasrecEngine.rank_seed(c["seed"], rank))
        eng.FUSED_TAIL_DP = fused
        eng.load_state_dict(P)
        Bl = c["B"] // world
        pl = eng.plan(Bl, c["T"], 2, need_grad=True)
        ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=True)
        packed = []
        for batch in _fold_batches(T):
            local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
            packed.append(eng.pack_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"]))
        eng.set_input_pool(pl, torch.stack(packed))
        if mode != "eager":           # (the graph pair's first step runs the local-gradients graph, then captures the pair)
            eng.capture_local_grads(pl)
        umax = eng.n_sparse_train(pl, dp=True) if mode == "pair" else None
        losses, first = [], None
        for t in range(c["K"]):
            eng.train_step_dp(pl, ex, use_graph=mode != "eager", umax=umax)
            eng.sync()
            eng.check_index_error(pl)
            assert bool(pl.tail2) == fused, "the data-parallel step did not take the path the test names"
            losses.append(float(pl.loss.item()))
            if t == 0:
                first = {name: eng.dense.view(name, eng.dense.grad).cpu().numpy().copy() for name in eng.dense.slots}
        if mode == "pair":
            assert len(getattr(pl, "dp_graphs", {})) == 1, "the graph pair was not captured"
        eng.flush_table()
        eng.sync()
        q.put((rank, {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}, losses, first))
    finally:
        dist.destroy_process_group()


def _fold_run(fused, mode, T=50):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fold_worker, args=(r, world, port, q, fused, mode, T)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return outs


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode,T", [("pair", 50), ("local", 50), ("eager", 50), ("pair", 20), ("eager", 20)])      # (T <= 32: folded since round 6, FOLD_SHORT)
def test_two_rank_dp_folded_step_matches_fifteen_launch_step_and_oracle(mode, T):
    """SasrecEngine.FUSED_TAIL_DP: every rank's local half of the data-parallel step in the single-GPU step's folded form (step head, head on
    the forward's tail, embedding backward on the last strip, position rows in the tail; phase B of the segment reduce + the chunk's packing
    as one launch, amid_grad_tail_live_dp_f32).  Held: the two replicas bit-identical; the first step's loss on every rank bit-identical
    to the fifteen-launch data-parallel step's and its summed dense gradients to rounding; the parameters after K steps (rows that lag
    and come back: a three-batch pool) to rounding of the fifteen-launch step's and within 1e-4 of ONE process stepping the oracle
    (dense Adam) over the global batches."""
    c = dict(FOLD, T=T)
    world = 2
    res = {fused: _fold_run(fused, mode, T) for fused in (False, True)}
    for fused, outs in res.items():
        for k in outs[0][1]:
            assert (outs[0][1][k] == outs[1][1][k]).all(), f"replicas diverged on {k} (folded={fused})"
    for r in range(world):
        assert res[True][r][2][0] == res[False][r][2][0], (res[True][r][2], res[False][r][2])
        for name, want in res[False][r][3].items():
            got = res[True][r][3][name]
            scale = max(float(abs(want).max()), 1e-30)
            assert float(abs(got - want).max()) / scale < 2e-6, name
    P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"]), seed=9)
    opt = orc.DenseAdam(P, lr=c["lr"])
    Bl = c["B"] // world
    from amid_amd.engine import SasrecEngine
    batches = _fold_batches(T)
    for t in range(1, c["K"] + 1):
        per_rank = [orc.philox_masks_sasrec(Bl, c["T"], c["D"], seed=SasrecEngine.rank_seed(c["seed"], r), step=t) for r in range(world)]
        masks = {k: torch.cat([m[k] for m in per_rank], 0) for k in per_rank[0]}
        orc.train_step("sasrec", P, opt, batches[(t - 1) % 3], masks)
    for fused in (False, True):
        sd = {k: torch.from_numpy(v) for k, v in res[fused][0][1].items()}
        for k, v in P.items():
            d = (sd[k] - v).abs()
            if k.endswith("in_proj_bias"):
                n = v.numel() // 3
                d = torch.cat((d[:n], d[2 * n:]))
            assert float(d.max()) < 1e-4, (fused, k, float(d.max()))
    a, b = res[False][0][1], res[True][0][1]
    for k in a:
        ta, tb = torch.from_numpy(a[k]).double(), torch.from_numpy(b[k]).double()
        assert float((ta - tb).norm() / ta.norm().clamp(min=1e-30)) < 1e-3, k


ITC = dict(n_items=300, D=64, T=20, hid=16, B=8, K=3, seed=13, lr=1e-3, ts2=0.15)


def _itc_params_and_batches():
    c = ITC
    P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"], itc_bs=c["B"]), seed=47)
    for d in (1, 2):
        P[f"sac{d}.last_layernorm.weight"] *= 0.3          # keeps the batch softmax of the pair-max scores away from one-hot
    batches = []
    for t in range(c["K"]):
        g = torch.Generator().manual_seed(70 + t)
        b = orc.synthetic_batch(c["B"], c["T"], c["n_items"] - 1, pad_id=c["n_items"] - 1, neg=1, seed=600 + t)
        b["seq_d1"] = torch.randint(1, c["n_items"] - 1, (c["B"], c["T"]), generator=g)      # no shared pad positions: distinct pair-max scores
        b["seq_d2"] = torch.randint(1, c["n_items"] - 1, (c["B"], c["T"]), generator=g)
        batches.append(b)
    return P, batches


def _itc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, shard_batch
        from amid_amd.engine import SasrecEngine
        c = ITC
        torch.cuda.set_device(0)
        P, batches = _itc_params_and_batches()
        # bs = the GLOBAL batch: InterComp's Linear(bs, 1) and its softmax span every rank's rows
        eng = SasrecEngine(c["n_items"], c["D"], c["T"], c["hid"], device="cuda:0", lr=c["lr"], seed=SasrecEngine.rank_seed(c["seed"], rank),
                           itc_bs=c["B"], itc_threshold=c["ts2"])
        eng.load_state_dict(P)
        Bl = c["B"] // world
        pl = eng.plan(Bl, c["T"], 2, need_grad=True)
        assert pl.itc_world == world
        ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=True)
        gates = []
        for batch in batches:
            local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
            eng.load_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"])
            # a bound the host knows (here the largest possible): from the second step on the step replays graph segments cut at InterComp's
            # mid-step collectives (engine._coll); the first step runs eagerly and captures them
            eng.train_step_dp(pl, ex, use_graph=True, umax=eng.n_sparse_train(pl))
            eng.sync()
            assert any(isinstance(v[0], list) and len(v[0]) >= 5 for v in getattr(pl, "dp_graphs", {}).values()), "the step was not captured in segments"
            gates.append(pl.itc_gate.cpu().clone())
        eng.flush_table()
        eng.sync()
        q.put((rank, {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}, [g.numpy().copy() for g in gates], float(pl.loss.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_dp_with_intercomp_matches_global_batch_oracle():
    """isItC (what run.sh trains, /root/reference/run.sh:1) under data parallel: InterComp's softmax over the batch and Linear(bs, 1)
    (model_seq.py:490-495) span the GLOBAL batch -- each rank holds half of its rows, the ranks all-gather pair-max scalars, user
    vectors and their gradients in the middle of the step.  Both replicas end bit-identical and equal to ONE process stepping the
    oracle (isItC=True, dense Adam) over the global batches; the gates are those of the global softmax."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_itc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = ITC
    P, batches = _itc_params_and_batches()
    opt = orc.DenseAdam(P, lr=c["lr"])
    Bl = c["B"] // world
    from amid_amd.engine import SasrecEngine
    for t, batch in enumerate(batches, start=1):
        per_rank = [orc.philox_masks_sasrec(Bl, c["T"], c["D"], seed=SasrecEngine.rank_seed(c["seed"], r), step=t) for r in range(world)]
        masks = {k: torch.cat([m[k] for m in per_rank], 0) for k in per_rank[0]}
        taps = {}
        orc.sasrec_forward(P, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], masks, taps, isItC=True, threshold2=c["ts2"])
        gate = taps["itc_d1"]["gate"]
        assert 0 < int(gate.sum()) < c["B"] and taps["itc_d1"]["margin"] > 1e-4, (gate, taps["itc_d1"]["margin"])
        for o in outs:
            assert torch.equal(torch.from_numpy(o[2][t - 1]), gate), (t, o[0])
        orc.train_step("sasrec", P, opt, batch, masks, isItC=True, threshold2=c["ts2"])
    sd0, sd1 = ({k: torch.from_numpy(v) for k, v in o[1].items()} for o in outs)
    for k, v in P.items():
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged on {k}"
        d = (sd0[k] - v).abs()
        if k.endswith("in_proj_bias"):
            n = v.numel() // 3
            d = torch.cat((d[:n], d[2 * n:]))
        assert float(d.max()) < 1e-4, (k, float(d.max()))


INC = dict(n_items=300, D=64, T=20, hid=16, B=8, K=3, seed=17, lr=1e-3, ts1=0.13)


def _inc_params_and_batches():
    c = INC
    P = orc.random_params(orc.sasrec_param_shapes(c["n_items"], c["D"], c["T"], c["hid"], inc_bs=c["B"]), seed=50 + c["D"] + c["T"])
    P["item_emb_layer.emb_item.weight"] *= 3.0          # self pair-max scores a few units apart: the batch softmax is not flat
    batches = []
    for t in range(c["K"]):
        g = torch.Generator().manual_seed(90 + t)
        b = orc.synthetic_batch(c["B"], c["T"], c["n_items"] - 1, pad_id=c["n_items"] - 1, neg=1, seed=700 + t)
        b["seq_d1"] = torch.randint(1, c["n_items"] - 1, (c["B"], c["T"]), generator=g)
        b["seq_d2"] = torch.randint(1, c["n_items"] - 1, (c["B"], c["T"]), generator=g)
        batches.append(b)
    return P, batches


def _inc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, shard_batch
        from amid_amd.engine import SasrecEngine
        c = INC
        torch.cuda.set_device(0)
        P, batches = _inc_params_and_batches()
        # bs = the GLOBAL batch: InnerComp's Linear(bs, 1) and its softmax span every rank's rows
        eng = SasrecEngine(c["n_items"], c["D"], c["T"], c["hid"], device="cuda:0", lr=c["lr"], seed=SasrecEngine.rank_seed(c["seed"], rank),
                           inc_bs=c["B"], inc_threshold=c["ts1"])
        eng.load_state_dict(P)
        Bl = c["B"] // world
        pl = eng.plan(Bl, c["T"], 2, need_grad=True)
        assert pl.inc_world == world
        ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=True)
        gates = []
        for batch in batches:
            local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
            eng.load_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"])
            # (as the InterComp worker: eager first step, then graph segments cut at InnerComp's gather and its two all-reduces)
            eng.train_step_dp(pl, ex, use_graph=True, umax=eng.n_sparse_train(pl))
            eng.sync()
            assert any(isinstance(v[0], list) and len(v[0]) >= 7 for v in getattr(pl, "dp_graphs", {}).values()), "the step was not captured in segments"
            gates.append(pl.inc_gate.cpu().clone())
        eng.flush_table()
        eng.sync()
        sd = {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}
        try:                                                   # a shard outside a data-parallel step is refused, loudly
            eng.enqueue_train_step(pl)
            refused = False
        except ValueError:
            refused = True
        eng.sync()
        q.put((rank, sd, [g.numpy().copy() for g in gates], refused))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_dp_with_innercomp_matches_global_batch_oracle():
    """isInC under data parallel: InnerComp's softmax over the batch and Linear(bs, 1) (model_seq.py:459-472) span the GLOBAL batch IN
    FRONT of the encoders -- each rank holds half of the rows; the ranks all-gather their self pair-max scores, all-reduce the partial
    token sums S forward and the group's gradient dZ backward (amid_inc_*_shard_f32).  Both replicas end bit-identical and equal to
    ONE process stepping the oracle (isInC=True, dense Adam) over the global batches; every rank's gates are its rows of the global
    softmax's."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_inc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = INC
    P, batches = _inc_params_and_batches()
    opt = orc.DenseAdam(P, lr=c["lr"])
    Bl = c["B"] // world
    from amid_amd.engine import SasrecEngine
    for t, batch in enumerate(batches, start=1):
        per_rank = [orc.philox_masks_sasrec(Bl, 2 * c["T"], c["D"], seed=SasrecEngine.rank_seed(c["seed"], r), step=t) for r in range(world)]
        masks = {k: torch.cat([m[k] for m in per_rank], 0) for k in per_rank[0]}
        taps = {}
        orc.sasrec_forward(P, batch["i_node"], batch["neg_samples"], batch["seq_d1"], batch["seq_d2"], masks, taps, isInC=True, threshold1=c["ts1"])
        for d in (1, 2):
            gate = taps[f"inc_d{d}"]["gate"]
            assert 0 < int(gate.sum()) < c["B"] and taps[f"inc_d{d}"]["margin"] > 1e-4, (gate, taps[f"inc_d{d}"]["margin"])
            for o in outs:
                r = o[0]
                assert torch.equal(torch.from_numpy(o[2][t - 1][d - 1]), gate[r * Bl:(r + 1) * Bl]), (t, d, r)
        orc.train_step("sasrec", P, opt, batch, masks, isInC=True, threshold1=c["ts1"])
    assert all(o[3] for o in outs)
    sd0, sd1 = ({k: torch.from_numpy(v) for k, v in o[1].items()} for o in outs)
    for k, v in P.items():
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged on {k}"
        d = (sd0[k] - v).abs()
        if k.endswith("in_proj_bias"):
            n = v.numel() // 3
            d = torch.cat((d[:n], d[2 * n:]))
        assert float(d.max()) < 1e-4, (k, float(d.max()))


BCOMP = dict(n_items=400, T=12, hid=16, B=8, K=3, seed=23, lr=1e-3)


def _bcomp_setup(kind):
    """Parameters, global batches and a threshold in the widest gap of the first batch's global batch softmax (tests/test_gpu_bert4rec.py)."""
    from tests.test_gpu_bert4rec import batch_with_masked_keys, comp_fwd_kw
    c = BCOMP
    P = orc.random_params(orc.bert4rec_param_shapes(c["n_items"], c["hid"], inc_bs=c["B"] if kind == "inc" else 0,
                                                    itc_bs=c["B"] if kind == "itc" else 0), seed=31)
    P["item_emb_layer.emb_item.weight"] = P["item_emb_layer.emb_item.weight"] * 0.1
    batches = [batch_with_masked_keys(c["B"], c["T"], c["n_items"], 900 + t) for t in range(c["K"])]
    taps = {}
    orc.bert4rec_forward(P, batches[0]["i_node"], batches[0]["neg_samples"], batches[0]["seq_d1"], batches[0]["seq_d2"], None, taps=taps,
                         **comp_fwd_kw(kind, 0.5))
    sm = torch.sort(torch.cat([taps[f"{kind}_d{d}"]["softmax"] for d in (1, 2)])).values
    i = int(torch.argmax(sm[1:] - sm[:-1]))
    return P, batches, float((sm[i] + sm[i + 1]) / 2)


def _bcomp_worker(rank, world, port, q, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amid_amd.dist import SparseDenseExchange, shard_batch
        from amid_amd.engine import SasrecEngine
        from amid_amd.engine_bert import Bert4recEngine
        c = BCOMP
        torch.cuda.set_device(0)
        P, batches, thr = _bcomp_setup(kind)
        # comp_bs = the GLOBAL batch: the module's Linear(bs, 1) and its softmax span every rank's rows
        eng = Bert4recEngine(c["n_items"], 128, c["T"], c["hid"], device="cuda:0", lr=c["lr"], seed=SasrecEngine.rank_seed(c["seed"], rank),
                             comp=kind, comp_bs=c["B"], comp_threshold=thr)
        eng.load_state_dict(P)
        pl = eng.plan(c["B"] // world, c["T"], 2, need_grad=True)
        assert pl.inc_world == world
        ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=True)
        gates = []
        for batch in batches:
            local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
            eng.load_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"])
            eng.train_step_dp(pl, ex, use_graph=True, umax=eng.n_sparse_train(pl))      # (eager first step, then graph segments)
            eng.sync()
            gates.append(pl.inc_gate.cpu().clone())
        eng.flush_table()
        eng.sync()
        q.put((rank, {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}, [g.numpy().copy() for g in gates]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("kind", ["inc", "itc"])
def test_two_rank_dp_with_bert4rec_comp_matches_global_batch_oracle(kind):
    """BERT4Rec(isInC) / (isItC) under data parallel (round 5): the token group in front of BERT4Rec's encoders (model_seq.py:283-294) spans the
    GLOBAL batch -- each rank holds half of the rows, the ranks all-gather their scores and all-reduce the token sums S forward and
    their gradient dZ backward (amid_bert_comp_*_shard_f32).  Both replicas end bit-identical and track ONE process stepping the oracle
    over the global batches; every rank's gates are its rows of the global softmax's."""
    from tests.test_gpu_bert4rec import comp_fwd_kw
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcomp_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = BCOMP
    P, batches, thr = _bcomp_setup(kind)
    opt = orc.DenseAdam(P, lr=c["lr"])
    Bl = c["B"] // world
    from amid_amd.engine import SasrecEngine
    for t, batch in enumerate(batches, start=1):
        per_rank = [orc.philox_masks_bert4rec(Bl, 2 * c["T"], seed=SasrecEngine.rank_seed(c["seed"], r), step=t) for r in range(world)]
        masks = {k: torch.cat([m[k] for m in per_rank], 0) for k in per_rank[0]}
        taps = {}
        loss, _, grads = orc.loss_and_grads("bert4rec", P, batch, masks, taps=taps, **comp_fwd_kw(kind, thr))
        for d in (1, 2):
            gate, margin = taps[f"{kind}_d{d}"]["gate"], taps[f"{kind}_d{d}"]["margin"]
            if margin < 1e-5:
                pytest.skip(f"a batch-softmax value sits within {margin:.2e} of the threshold: the gate is rounding-dependent")
            for o in outs:
                r = o[0]
                assert torch.equal(torch.from_numpy(o[2][t - 1][d - 1]).bool(), gate[r * Bl:(r + 1) * Bl].bool()), (t, d, r)
        opt.step(P, grads)
    sd0, sd1 = ({k: torch.from_numpy(v) for k, v in o[1].items()} for o in outs)
    for k, v in P.items():
        assert torch.equal(sd0[k], sd1[k]), f"replicas diverged on {k}"
        if k.endswith("linear_layers.1.bias"):
            continue
        d = (sd0[k] - v).abs()
        assert float((d > 3e-4).float().mean()) < 2e-3 and float(d.max()) < 3.5e-3, (k, float(d.max()), float((d > 3e-4).float().mean()))


def _dr_cli_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(rank), AMID_DIST_BACKEND="gloo")
    import amid_amd.train_sr_dr as tdr
    captured = {}
    real_train = tdr.train

    def spy(model, *a, **k):
        out = real_train(model, *a, **k)
        model.engine.flush_table(); model.engine.sync()
        captured["sd"] = {n: v.detach().cpu().numpy().copy() for n, v in model.state_dict().items()}
        return out

    tdr.train = spy
    summary = tdr.main(["--data_root", root, "-ds", "amazon", "-dm", "toy", "--overlap_ratio", "0.75", "--model", "sasrec", "--bs", "8",
                        "--seq_len", "20", "--emb_dim", "64", "--hid_dim", "16", "--epoch", "1", "--neg_nums", "19", "--seeds", "1",
                        "--device", "cuda:0", "-md", os.path.join(root, "model"), "--isItC", "True", "--ts2", "0.05", "--lr2", "0.01",
                        "--dr_e_w", "0.01", "--max_steps", "5"])
    q.put((rank, captured["sd"], {f"{k[0]}/{k[1]}": float(v) for k, v in summary[0].items()}))


@pytest.mark.timeout(900)
def test_train_sr_dr_cli_isitc_data_parallel_two_ranks(tmp_path):
    """run.sh's own configuration -- train_sr_dr.py --isItC True (the doubly-robust trainer: two objectives, two Adam states) -- under a
    two-process launch: InterComp is built for the global batch of 2 x --bs rows, both loops shard their batches and exchange
    gradients every step; the replicas must end bit-identical with identical metrics."""
    import numpy as np
    from tests.test_gpu_module import _write_csv
    rng = np.random.default_rng(3)
    root = tmp_path / "amazon_dataset"
    root.mkdir()
    _write_csv(root / "toy_train75.csv", 120, rng, 1, 400, 400, 900)
    _write_csv(root / "toy_train75_DR.csv", 120, rng, 1, 400, 400, 900, ob_label=True)
    _write_csv(root / "toy_test.csv", 48, rng, 1, 400, 400, 900)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dr_cli_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs, 800) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, m0), (_, sd1, m1) = outs
    assert any(k.startswith("itc_d1.trans_bs") and v.shape[-1] == 16 for k, v in sd0.items())       # Linear(bs, 1) over the GLOBAL batch
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    assert m0 == m1 and all(0.0 <= v <= 1.0 for v in m0.values())


def _cli_worker(rank, world, port, root, q, dm="toy", extra=()):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", WORLD_SIZE=str(world),
                      RANK=str(rank), LOCAL_RANK=str(rank), AMID_DIST_BACKEND="gloo")
    import amid_amd.train_sr as tsr
    captured = {}
    real_train = tsr.train

    def spy(model, *a, **k):               # keep the trained module to compare the replicas afterwards
        out = real_train(model, *a, **k)
        captured["sd"] = {n: v.detach().cpu().numpy().copy() for n, v in model.state_dict().items()}
        return out

    tsr.train = spy
    summary = tsr.main(["--data_root", root, "-ds", "amazon", "-dm", dm, "--overlap_ratio", "0.75", "--model", "sasrec", "--bs", "16",
                        "--seq_len", "20", "--emb_dim", "64", "--hid_dim", "16", "--epoch", "1", "--neg_nums", "19", "--seeds", "1",
                        "--device", "cuda:0", "-md", os.path.join(root, "model")] + list(extra))
    q.put((rank, captured["sd"], {f"{k[0]}/{k[1]}": float(v) for k, v in summary[0].items()}))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("extra", [(), ("--isInC", "True", "--ts1", "0.02"), ("--isItC", "True", "--ts2", "0.02")], ids=["plain", "isInC", "isItC"])
def test_train_sr_cli_data_parallel_two_ranks(tmp_path, extra):
    """The CLI under a two-process launch (what torch.distributed.run sets up): ranks shard every global batch, exchange
    gradients each step and must end with bit-identical parameters and identical metrics -- plain, with --isInC and with --isItC
    (the comp module then spans the GLOBAL batch of world x --bs rows; the evaluation steps through batches of that size)."""
    import numpy as np
    from tests.test_gpu_module import _write_csv
    rng = np.random.default_rng(1)
    root = tmp_path / "amazon_dataset"
    root.mkdir()
    _write_csv(root / "toy_train75.csv", 200, rng, 1, 400, 400, 900)
    _write_csv(root / "toy_test.csv", 64, rng, 1, 400, 400, 900)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_worker, args=(r, world, port, str(tmp_path), q, "toy", extra)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, m0), (_, sd1, m1) = outs
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    if extra and extra[0] == "--isInC":
        assert sd0["inc_d1.trans_bs.weight"].shape[-1] == 2 * 16 and sd0["sac1.pos_emb.weight"].shape[0] == 2 * 20
    assert m0 == m1 and all(0.0 <= v <= 1.0 for v in m0.values())


JOINT_STEPS = 6


@pytest.mark.timeout(900)
def test_train_sr_cli_joint_mode_two_ranks_equals_single_process_oracle(tmp_path):
    """BASELINE.json configs[3]'s joint mode through the CLI (-dm toy+toyb: shared table, the second dataset's items item_length + 2
    rows behind the first's, alternating batches) under a two-process launch: both replicas end bit-identical, and equal to ONE
    process stepping the oracle (dense Adam, the reference's arithmetic) over the same global batches -- each global batch being
    rank 0's rows followed by rank 1's, with each rank's own dropout stream."""
    import numpy as np
    from tests.test_gpu_module import _write_csv
    from amid_amd.dataset_seq import DeviceBatches, DualDomainSeqDataset, JointBatches
    from amid_amd.engine import SasrecEngine
    from amid_amd.model_seq import SASRec
    rng = np.random.default_rng(2)
    root = tmp_path / "amazon_dataset"
    root.mkdir()
    _write_csv(root / "toy_train75.csv", 140, rng, 1, 400, 400, 900)
    _write_csv(root / "toy_test.csv", 40, rng, 1, 400, 400, 900)
    _write_csv(root / "toyb_train75.csv", 100, rng, 1, 300, 300, 800)
    _write_csv(root / "toyb_test.csv", 40, rng, 1, 300, 300, 800)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cli_worker, args=(r, world, port, str(tmp_path), q, "toy+toyb", ("--max_steps", str(JOINT_STEPS))))
             for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([_get(q, procs, 800) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, m0), (_, sd1, m1) = outs
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k
    assert m0 == m1

    # one process, the oracle, the same global batches (world x bs rows each)
    item_length, T, D, hid, bs = 447410, 20, 64, 16, 16
    dss = []
    for j, dm in enumerate(("toy", "toyb")):
        ds = DualDomainSeqDataset(seq_len=T, isTrain=True, neg_nums=19, long_length=7, pad_id=item_length + 1, seed=0,
                                  csv_path=str(root / f"{dm}_train75.csv"))
        if j:
            ds.shift_items(item_length + 2)
            assert int(ds.i_node.min()) > item_length + 1
        dss.append(ds)
    loader = JointBatches(*[DeviceBatches(d, world * bs, shuffle=True, device="cuda:0", seed=0) for d in dss])
    assert [w for w, _ in loader.order()][:4] == [0, 1, 0, 1]
    model = SASRec(user_length=2 * 895510, user_emb_dim=D, item_length=2 * item_length, item_emb_dim=D, seq_len=T, hid_dim=hid, bs=bs,
                   isInC=False, isItC=False, threshold1=0.5, threshold2=0.5, lr=5e-4, seed=0)
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    del model
    opt = orc.DenseAdam(P, lr=5e-4)
    for t, b in zip(range(1, JOINT_STEPS + 1), loader):
        batch = {k: b[k].cpu() for k in ("i_node", "neg_samples", "seq_d1", "seq_d2", "label", "domain_id")}
        per_rank = [orc.philox_masks_sasrec(bs, T, D, seed=SasrecEngine.rank_seed(0, r), step=t) for r in range(world)]
        orc.train_step("sasrec", P, opt, batch, {k: torch.cat([m[k] for m in per_rank], 0) for k in per_rank[0]})
    for k, v in P.items():
        d = (torch.from_numpy(sd0[k]) - v).abs()
        if k.endswith("in_proj_bias"):
            n = v.numel() // 3
            d = torch.cat((d[:n], d[2 * n:]))          # the key bias has a zero gradient (softmax shift invariance): Adam noise only
        assert float(d.max()) < 2e-4, (k, float(d.max()))
