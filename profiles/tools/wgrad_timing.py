import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from amid_amd.engine import SasrecEngine
eng = SasrecEngine(bench.N_ROWS, bench.D, bench.T, bench.HID, lr=5e-4, seed=1)
bench.init_params(eng, 0)
pl = eng.plan(bench.B, bench.T, 2, True)
gen = torch.Generator().manual_seed(1)
for it in range(3):
    b = bench.synth_batch(gen, "cuda")
    eng.load_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"])
    eng.enqueue_train_step(pl); eng.sync()
bp = pl.b_part[0].cpu()      # [2, 6, splits, D]
for off, nm in ((0, "wave0"), (8, "wave4")):
    t = bp[..., off:off + 6].reshape(-1, 6)
    print(nm, "per-WG cycles: prologue(first chunk), loop+epilogue, mma, stage, sync -- mean / min / max")
    for j, n in enumerate(("prologue", "loop+epi", "mma", "stage", "sync")):
        print("  ", n, float(t[:, j].mean()), float(t[:, j].min()), float(t[:, j].max()))
