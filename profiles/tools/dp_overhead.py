"""Cost of the data-parallel code path on ONE GPU: a world of one RCCL rank runs the full per-step exchange (dense all-reduce,
sparse pad + all-gather, sorted-list merge, Adam on the merged lists) after the local-gradients graph, against the single-GPU
whole-step graph.  The collectives of a 1-rank world cost their launch latency only, so the difference is the host + merge
overhead a multi-GPU step pays on top of the wire time.   python profiles/tools/dp_overhead.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch, torch.distributed as dist
import bench
from amid_amd.dist import SparseDenseExchange
from amid_amd.engine import SasrecEngine

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
eng = SasrecEngine(bench.N_ROWS, bench.D, bench.T, bench.HID, lr=5e-4, seed=1)
bench.init_params(eng, 0)
pl = eng.plan(bench.B, bench.T, 2, True)
gen = torch.Generator().manual_seed(1)
pool, cnt = [], []
for _ in range(30):
    b = bench.synth_batch(gen, "cuda")
    pool.append(eng.pack_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"]))
    cnt.append(int(torch.unique(torch.cat([b[k].reshape(-1) for k in ("i_node", "neg_samples", "seq_d1", "seq_d2")])).numel()))
ex = SparseDenseExchange(eng.merge_backend(pl.shape.n_idx), always=True)
eng.set_input_pool(pl, torch.stack(pool))          # as bench.py: the batches stay in HBM, the step picks its batch on the device
eng.capture_local_grads(pl)
def single(n):
    eng.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        eng.replay_train_step(pl)
    eng.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
cap = (max(cnt) + 255) // 256 * 256          # one bound for the whole run, as bench.py's (a per-step bound must follow the pool's phase)
def run(n, umax_known, dense="gather"):
    eng.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        eng.train_step_dp(pl, ex, use_graph=True, umax=cap if umax_known else None, dense=dense)
    eng.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
eng.capture_train_step(pl)
single(30)
print("single-GPU whole-step graph : %.4f ms/step" % single(300))
for dense in ("gather", "allreduce"):
    run(5, True, dense)
    print("dp path, fixed bound %d, graph pair, dense exchange = %s : %.4f ms/step" % (cap, dense, run(200, True, dense)))
print("dp path, host sync / step: %.4f ms/step" % run(200, False))
# round 5: isItC under the data-parallel path -- the step used to be enqueued eagerly (collectives in the middle of forward and backward);
# with a known bound it is now captured in segments cut at those collectives (engine._coll) and replayed
eng2 = SasrecEngine(bench.N_ROWS, bench.D, bench.T, bench.HID, lr=5e-4, seed=1, itc_bs=bench.B, itc_threshold=0.2)
bench.init_params(eng2, 0)
pl2 = eng2.plan(bench.B, bench.T, 2, True)
ex2 = SparseDenseExchange(eng2.merge_backend(pl2.shape.n_idx), always=True)
eng2.set_input_pool(pl2, torch.stack([eng2.pack_batch(pl2, *[bench.synth_batch(gen, "cuda")[k] for k in ("i_node", "neg_samples", "seq_d1", "seq_d2", "label", "domain_id")]) for _ in range(30)]))
def run2(n, use_graph):
    eng2.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        eng2.train_step_dp(pl2, ex2, use_graph=use_graph, umax=eng2.n_sparse_train(pl2))
    eng2.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run2(5, False)
print("isItC dp path, eager launches : %.4f ms/step" % run2(100, False))
run2(5, True)
print("isItC dp path, graph segments : %.4f ms/step" % run2(100, True))
eng.sync(); torch.cuda.synchronize(); dist.barrier(); dist.destroy_process_group()
