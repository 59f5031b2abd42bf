"""Which conditions of the folded step (SasrecEngine._folded_step_shape / _tail2_ok) a workload fails.
    python profiles/tools/probe/fold_conditions.py cfg5-uniform [bf16]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, R)
import torch
import bench
from amid_amd._lib import lib
from amid_amd.engine import SasrecEngine
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg5-uniform"]
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda", 0)
eng = SasrecEngine(wl["n_rows"], wl.get("D", bench.D), wl.get("T", bench.T), bench.HID, device=dev, lr=5e-4, seed=1234, compute=dt)
bench.init_params(eng, seed=0)
B, T = wl["B"], wl.get("T", bench.T)
pl = eng.plan(B, T, 1 + bench.NEG, need_grad=True)
gen = torch.Generator().manual_seed(1000)
pool = []
for _ in range(4):
    b = bench.synth_batch(gen, dev, wl)
    pool.append(eng.pack_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"]))
eng.set_input_pool(pl, torch.stack(pool))
eng._in_train_step = True
c = dict(SORT_RIDERS=eng.SORT_RIDERS, D128=eng.D == 128, f32=eng.compute == "f32", NI=pl.shape.NI > 1, FUSED_HEAD=eng.FUSED_HEAD, BWD_SPLIT=eng.BWD_SPLIT,
         strip=bool(pl.strip), pool=eng.input_pool(pl) is not None, live_forward_ok=bool(eng.live_forward_ok(pl)), wgrad_mode=eng._wgrad_mode(eng.D),
         fold_catchup=bool(eng._fold_catchup(pl)), sort_plan=eng._sort_plan_c(pl) is not None, n_compact=pl.n_compact, splits=pl.splits,
         tiles_ok=(pl.n_compact + 2047) // 2048 <= 12 * pl.splits, p3_bwd=bool(eng._p3_bwd_for(pl)),
         folded=bool(eng._folded_step_shape(pl)), tail2=bool(eng._tail2_ok(pl)), seq_supported=lib().value("amid_sas_seq_supported", B, T, eng.D, eng.H))
for k, v in c.items():
    print(f"{k:18s} {v}")
