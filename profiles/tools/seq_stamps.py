#!/usr/bin/env python3
"""Where the fused encoder-forward kernel's cycles go (csrc/sasrec_seq.hip): runs a train step of the headline shape on the DIAGNOSTIC
library built by profiles/tools/build_diag.sh (s_memtime stamps of workgroup 0) and prints the deltas per wave."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import amid_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "profiles", "tools", "_diag", "libamid_hip_diag.so")
import torch  # noqa: E402
from amid_amd.engine import SasrecEngine  # noqa: E402
from oracle import amid_oracle as orc  # noqa: E402

B, T, D, hid, n_items = 256, 50, 128, 32, 3000
P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=1)
eng = SasrecEngine(n_items, D, T, hid, seed=3)
eng.load_state_dict(P)
pl = eng.plan(B, T, 2, need_grad=True)
batch = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=2)
cu = {k: v.cuda() for k, v in batch.items()}
for _ in range(int(os.environ.get("STAMP_STEPS", "4"))):
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
    eng.enqueue_train_step(pl)
    eng.sync()
L = _lib.lib()._dll
host = (ctypes.c_ulonglong * (4 * 32))()
assert L.amid_seq_stamps_read(host) == 0
names = {0: "entry", 1: "prologue"}
for l in (0, 1):
    for i, n in enumerate(("LN1", "k", "v", "q", "attention", "o+LN2", "c1", "c2")):
        names[2 + 10 * l + i] = f"L{l}.{n}"
for i, n in zip(range(22, 31), ("rd1 begin", "barrier A", "K/V written", "barrier B", "reads + keep words issued", "S MFMAs", "mask + max", "exp", "PV MFMAs")):
    names[i] = n
order = sorted(k for k in names if k < 20)
fine = list(range(22, 31))
for w in range(1):
    dc, dr = host[w * 32 + 21] - host[w * 32 + 0], host[w * 32 + 31] - host[w * 32 + 20]
    print(f"in-kernel clock: {dc} shader cycles in {dr} ticks of the 100 MHz real-time counter = {dc / dr * 0.1:.3f} GHz")
for w in range(4):
    t = {i: host[w * 32 + i] for i in order}
    tf = [host[w * 32 + i] for i in fine]
    print(f"wave {w} attention round 1 (last layer): " + ", ".join(f"{names[fine[k]]} +{tf[k] - tf[k - 1]}" for k in range(1, len(fine))))
    print(f"wave {w}: total {t[order[-1]] - t[0]}: " + ", ".join(f"{names[i]} +{t[i] - t[order[k - 1]]}" for k, i in enumerate(order) if k))

fine = (ctypes.c_ulonglong * (8 * 64))()
if hasattr(L, "amid_seq_fine_read") and L.amid_seq_fine_read(fine) == 0:
    fn = ["k:wait", "k:mma", "k:epi", "v:wait", "v:mma", "v:epi", "q:wait", "q:mma", "q:epi", "q store", "attention", "(bias loads)", "o:wait", "o:mma",
          "o:epi+LN2", "c1:wait", "c1:mma", "c1:epi", "c2:wait", "c2:mma", "c2:epi"]
    for w in range(4):
        t = [fine[w * 64 + i] for i in range(22)]
        print(f"wave {w} last layer, fine: " + ", ".join(f"{fn[i]} +{t[i + 1] - t[i]}" for i in range(21)))

sched = (ctypes.c_ulonglong * (1024 * 4))()
assert L.amid_seq_sched_read(sched) == 0
rows = [(i, sched[i * 4], sched[i * 4 + 1], sched[i * 4 + 2]) for i in range(512) if sched[i * 4]]
t0 = min(r[1] for r in rows)
live = [r for r in rows if r[2] - r[1] > 2000]
print(f"{len(rows)} workgroups stamped, {len(live)} long ones; kernel span {(max(r[2] for r in rows) - t0) / 100:.1f} us")
starts = sorted((r[1] - t0) / 100 for r in live)
print("start times of the long workgroups (us), deciles:", [round(starts[int(k * (len(starts) - 1) / 10)], 1) for k in range(11)])
durs = sorted((r[2] - r[1]) / 100 for r in live)
print("durations (us), deciles:", [round(durs[int(k * (len(durs) - 1) / 10)], 1) for k in range(11)])
