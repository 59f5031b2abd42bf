#!/usr/bin/env python3
"""Where the N-split fused forward's cycles go (csrc/sasrec_seqn.hip): a train step of the headline shape on the DIAGNOSTIC library
(profiles/tools/build_diag.sh: s_memtime stamps of workgroup 0, last layer) -- per wave: ring wait, LayerNorm, every product, the
attention core.  STAMP_VARIANT picks the build (default 42; set on the class, SasrecEngine.SEQ_FWD_VARIANT)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
VARIANT = int(os.environ.get("STAMP_VARIANT", "42"))
import amid_amd._lib as _lib  # noqa: E402

_lib.LIB_PATH = os.environ.get("AMID_DIAG_LIB") or os.path.join(ROOT, "profiles", "tools", "_diag", "libamid_hip_diag.so")
import torch  # noqa: E402
from amid_amd.engine import SasrecEngine  # noqa: E402

SasrecEngine.SEQ_FWD_VARIANT = VARIANT
from oracle import amid_oracle as orc  # noqa: E402

B, T, D, hid, n_items = int(os.environ.get("STAMP_B", "256")), int(os.environ.get("STAMP_T", "50")), 128, 32, 3000
P = orc.random_params(orc.sasrec_param_shapes(n_items, D, T, hid), seed=1)
eng = SasrecEngine(n_items, D, T, hid, seed=3, compute=os.environ.get("STAMP_DTYPE", "f32"))
eng.load_state_dict(P)
pl = eng.plan(B, T, 2, need_grad=True)
batch = orc.synthetic_batch(B, T, n_items - 1, pad_id=n_items - 1, neg=1, seed=2)
cu = {k: v.cuda() for k, v in batch.items()}
for _ in range(4):
    eng.load_batch(pl, cu["i_node"], cu["neg_samples"], cu["seq_d1"], cu["seq_d2"], cu["label"], cu["domain_id"])
    eng.enqueue_train_step(pl)
    eng.sync()
L = _lib.lib()._dll
host = (ctypes.c_ulonglong * (8 * 64))()
assert L.amid_seqn_stamps_read(host) == 0
names = ["k:ring wait", "x read + LN1", "k product", "v: ring wait", "v product", "q: ring wait", "q product", "q store + K/V images",
         "attention core", "stats + o parts out + ring wait", "o read + product + residual", "r out + ring wait", "r read + LN2",
         "c1 product + dropout + relu", "h out + ring wait", "h read + c2 product + epilogue"]
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
nw = {42: 8, 22: 4, 24: 8, 14: 4, 18: 8}[VARIANT]
for w in range(nw):
    t = [host[w * 64 + i] for i in order]
    tot = host[w * 64 + 63] - host[w * 64 + 62]
    rt = host[w * 64 + 61] - host[w * 64 + 60]            # the same interval on the constant 100 MHz clock
    if w == 0 and rt > 0:
        print(f"workgroup 0: {tot} cycles in {rt / 100:.2f} us = {tot / rt / 10:.3f} GHz; the last workgroup starts "
              f"{(host[58] - host[60]) / 100:+.2f} us after workgroup 0 and ends {(host[59] - host[61]) / 100:+.2f} us after it")
    print(f"wave {w}: kernel {tot} cycles; last layer {t[-1] - t[0]}: " + ", ".join(f"{names[i]} +{t[i + 1] - t[i]}" for i in range(16)))
