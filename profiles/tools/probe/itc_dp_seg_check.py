"""isItC / isInC under the data-parallel path on a 1-rank world: eager launches against graph segments (engine._coll), same inputs --
the parameters after K steps must agree bit for bit.   python profiles/tools/probe/itc_dp_seg_check.py [itc|inc]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
import torch, torch.distributed as dist
from oracle import amid_oracle as orc
from amid_amd.dist import SparseDenseExchange
from amid_amd.engine import SasrecEngine

kind = sys.argv[1] if len(sys.argv) > 1 else "itc"
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
n_items, D, T, hid, B, K = 300, 128, 20, 16, 8, 4
kw = dict(itc_bs=B) if kind == "itc" else dict(inc_bs=B)
P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid, **kw), seed=47)
batches = [orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=600 + t) for t in range(K)]
out = {}
for mode in ("eager", "segments"):
    ekw = dict(itc_bs=B, itc_threshold=0.2) if kind == "itc" else dict(inc_bs=B, inc_threshold=0.13)
    eng = SasrecEngine(n_items, D, T, hid, device="cuda:0", lr=1e-3, seed=5, **ekw)
    eng.load_state_dict(P)
    pl = eng.plan(B, T, 2, need_grad=True)
    ex = SparseDenseExchange(eng.merge_backend(pl.shape.n_idx), always=True, host_staging=True)
    for b in batches:
        cu = {k: v.cuda() for k, v in b.items()}
        eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
        eng.train_step_dp(pl, ex, use_graph=(mode == "segments"), umax=eng.n_sparse_train(pl))
        eng.sync()
    eng.flush_table(); eng.sync()
    out[mode] = {k: v.cpu().clone() for k, v in eng.state_dict().items()}
    print(mode, "loss", float(pl.loss.item()), "graphs", {k: (len(v[0]) if isinstance(v[0], list) else 1) for k, v in getattr(pl, "dp_graphs", {}).items()})
bad = {k: float((out["segments"][k] - v).abs().max()) for k, v in out["eager"].items() if not torch.equal(out["segments"][k], v)}
print("differ:", bad if bad else "nothing")
dist.destroy_process_group()
