#!/usr/bin/env python3
"""Duration of the fused encoder-forward launch against the number of live sequences (one workgroup per sequence at T = 50):
a step at one-workgroup-per-CU occupancy shows at 256."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from amid_amd._lib import KernelTimer, lib  # noqa: E402
from amid_amd.engine import SasrecEngine  # noqa: E402
from oracle import amid_oracle as orc  # noqa: E402

T, D, hid, n_items = 50, 128, 32, 3000
P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=1)
for B in [int(x) for x in (sys.argv[1:] or "64 128 192 224 256 288 384 512".split())]:
    eng = SasrecEngine(n_items, D, T, hid, seed=3)
    eng.load_state_dict(P)
    pl = eng.plan(B, T, 2, need_grad=True)
    batch = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=2)
    cu = {k: v.cuda() for k, v in batch.items()}
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
    for _ in range(3):
        eng.enqueue_train_step(pl)
    eng.sync()
    L = lib()
    L.timer = KernelTimer()
    for _ in range(10):
        eng.enqueue_train_step(pl)
        eng.sync()
    d = L.timer.collect(L)
    L.timer = None
    print(B, {k: round(1e3 * sum(v) / len(v), 1) for k, v in d.items() if "seq_fwd" in k or "strip" in k or "attn" in k})
