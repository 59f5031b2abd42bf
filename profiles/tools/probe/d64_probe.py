import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from amid_amd.engine import SasrecEngine
from amid_amd._lib import KernelTimer, lib
D = int(os.environ.get("PROBE_D", "64"))
bench.D = D
B, T = int(os.environ.get("PROBE_B", "256")), int(os.environ.get("PROBE_T", "50"))
eng = SasrecEngine(bench.N_ROWS, D, T, bench.HID, lr=5e-4, seed=1)
bench.D = D
import types
# init params like bench (generic helper reads eng.D)
bench.init_params(eng, seed=0)
pl = eng.plan(B, T, 2, need_grad=True)
g = torch.Generator().manual_seed(0)
b = bench.synth_batch(g, "cuda", dict(B=B, T=T, pad_id=bench.PAD_ID, max_id=bench.MAX_REAL_ID, kind="real"))
eng.load_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"])
L = lib()
eng.enqueue_train_step(pl); eng.sync()
L.timer = KernelTimer()
for _ in range(5):
    eng.enqueue_train_step(pl); eng.sync()
durs = L.timer.collect(L); L.timer = None
print(f"D={D}: " + ", ".join(f"{k} {1e3 * sum(v) / 5:.1f}us/{len(v) // 5}" for k, v in durs.items()))
eng.capture_train_step(pl)
for _ in range(20): eng.replay_train_step(pl)
eng.sync(); t0 = time.perf_counter()
for _ in range(200): eng.replay_train_step(pl)
eng.sync(); ms = (time.perf_counter() - t0) / 200 * 1e3
print(f"D={D}: {ms:.4f} ms/step")
