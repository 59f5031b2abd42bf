"""Two ranks (gloo, one GPU): isItC data-parallel steps eager against graph segments -- per-key difference of the final parameters.
    python profiles/tools/probe/itc_dp_seg_check2.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, q, seg):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_gpu_dp import ITC, _itc_params_and_batches
    from amid_amd.dist import SparseDenseExchange, shard_batch
    from amid_amd.engine import SasrecEngine
    c = ITC
    torch.cuda.set_device(0)
    P, batches = _itc_params_and_batches()
    eng = SasrecEngine(c["n_items"], c["D"], c["T"], c["hid"], device="cuda:0", lr=c["lr"], seed=SasrecEngine.rank_seed(c["seed"], rank),
                       itc_bs=c["B"], itc_threshold=c["ts2"])
    eng.load_state_dict(P)
    pl = eng.plan(c["B"] // world, c["T"], 2, need_grad=True)
    ex = SparseDenseExchange(eng.merge_backend(world * pl.shape.n_idx), host_staging=True)
    snaps = []
    for batch in batches:
        local = {k: v.cuda() for k, v in shard_batch(batch, rank, world).items()}
        eng.load_batch(pl, local["i_node"], local["neg_samples"], local["seq_d1"], local["seq_d2"], local["label"], local["domain_id"])
        eng.train_step_dp(pl, ex, use_graph=seg, umax=eng.n_sparse_train(pl))
        eng.sync()
        snaps.append(dict(loss=float(pl.loss.item()), gate=pl.itc_gate.cpu().clone(), n_uniq=int(pl.n_uniq.item()),
                          dense=eng.dense.data.cpu().clone(),
                          mids={k: getattr(pl, k).float().cpu().numpy().copy() for k in ("u_raw", "u_raw_g", "itc_s", "itc_s_g", "u_g", "u", "du", "du_g", "xg")}))
    eng.flush_table(); eng.sync()
    q.put((rank, seg, {k: v.cpu().numpy().copy() for k, v in eng.state_dict().items()}, [(s["loss"], s["n_uniq"], s["dense"].numpy().copy(), s["mids"]) for s in snaps]))
    dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    res = {}
    for seg in (False, True):
        q = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, 2, 29600 + int(seg), q, seg)) for r in range(2)]
        [p.start() for p in procs]
        outs = [q.get(timeout=300) for _ in range(2)]
        [p.join(60) for p in procs]
        res[seg] = sorted(outs, key=lambda t: t[0])
    import numpy as np
    for r in (0, 1):
        a, b = res[False][r], res[True][r]
        print("rank", r, "per-step (loss, n_uniq) eager", [(round(s[0], 6), s[1]) for s in a[3]], "segments", [(round(s[0], 6), s[1]) for s in b[3]])
        for t in range(len(a[3])):
            print("   step", t + 1, "dense params max diff", float(np.abs(a[3][t][2] - b[3][t][2]).max()),
                  {k: float(np.abs(a[3][t][3][k] - b[3][t][3][k]).max()) for k in a[3][t][3]})
        bad = {k: float(np.abs(a[2][k] - b[2][k]).max()) for k in a[2] if not np.array_equal(a[2][k], b[2][k])}
        print("   final differ:", {k: v for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:6]})
