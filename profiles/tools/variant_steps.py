"""Step time of the widened SASRec variants on the cfg 2 shape (B 256, T 50, D 128, hid 32, neg 1), graph-replayed, against the plain
model: isItC, isItC + isDR (what run.sh trains; both objectives), isInC.   python profiles/tools/variant_steps.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import bench
from amid_amd.engine import SasrecEngine

if os.environ.get("VARIANT_T"):          # e.g. VARIANT_T=20: the mybank shape run.sh trains on (train_sr_dr.py:550 default seq_len)
    bench.T = int(os.environ["VARIANT_T"])
    bench.WORKLOADS["cfg2"]["T"] = bench.T


def run(tag, steps=200, objective=0, **kw):
    eng = SasrecEngine(bench.N_ROWS, bench.D, bench.T, bench.HID, lr=5e-4, seed=1, **kw)
    bench.init_params(eng, 0)
    pl = eng.plan(bench.B, bench.T, 2, True)
    gen = torch.Generator().manual_seed(1)
    b = bench.synth_batch(gen, "cuda")
    ob = torch.randint(0, 2, (bench.B,), device="cuda") if kw.get("dr") else None
    eng.load_batch(pl, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], b["label"], b["domain_id"], ob)
    eng.dr_mode = objective
    if os.environ.get("VARIANT_KERNELS"):        # one eager step under the C-ABI timer: which entry points this variant's step is made of
        from amid_amd._lib import KernelTimer, lib
        L = lib()
        eng.enqueue_train_step(pl); eng.sync()
        L.timer = KernelTimer()
        for _ in range(5):
            eng.enqueue_train_step(pl); eng.sync()
        durs = L.timer.collect(L)
        L.timer = None
        print(f"  {tag}: " + ", ".join(f"{k} {1e3 * sum(v) / 5:.1f}us/{len(v) // 5}" for k, v in durs.items()), flush=True)
    eng.capture_train_step(pl)
    for _ in range(20):
        eng.replay_train_step(pl)
    eng.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        eng.replay_train_step(pl)
    eng.sync(); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag:34s} {ms:.4f} ms/step  {bench.B / ms * 1e3:,.0f} samples/s", flush=True)
    del eng, pl
    torch.cuda.empty_cache()


VARIANTS = {
    "plain": lambda: run("plain"),
    "itc": lambda: run("isItC", itc_bs=bench.B, itc_threshold=0.4),
    "itc-dr-e": lambda: run("isItC + isDR, loss_cls + w loss_dr_e", itc_bs=bench.B, itc_threshold=0.4, dr=True, dr_e_w=0.01),
    "itc-dr-r": lambda: run("isItC + isDR, loss_dr_r", objective=1, itc_bs=bench.B, itc_threshold=0.4, dr=True, dr_e_w=0.01),
    "inc": lambda: run("isInC (2T = 100 tokens)", inc_bs=bench.B, inc_threshold=0.5),
}
for name in (sys.argv[1:] or list(VARIANTS)):
    VARIANTS[name]()
