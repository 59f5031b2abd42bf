"""grad_tail with the two reduce tables of a plan (full-row / live-row LayerNorm partial slots): python profiles/tools/grad_tail_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import bench
from amid_amd._lib import lib
from amid_amd.engine import SasrecEngine
L = lib()
eng = SasrecEngine(bench.N_ROWS, bench.D, bench.T, bench.HID, lr=5e-4, seed=1)
bench.init_params(eng, 0)
pl = eng.plan(bench.B, bench.T, 2, True)
gen = torch.Generator().manual_seed(1)
b = bench.synth_batch(gen, "cuda")
eng.load_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"])
for _ in range(3):
    eng.enqueue_train_step(pl)
eng.sync()
def run(live, tag):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(eng.stream):
        for _ in range(5): eng._enqueue_grad_tail(pl, live)
        ev[0].record(eng.stream)
        for _ in range(50): eng._enqueue_grad_tail(pl, live)
        ev[1].record(eng.stream)
    torch.cuda.synchronize()
    print(f"{tag}: {ev[0].elapsed_time(ev[1]) / 50 * 1e3:.1f} us (grad_tail + spans), entries {pl.red_n_v if live else pl.red_n}")
run(False, "full-row table")
run(True, "live-row table")
