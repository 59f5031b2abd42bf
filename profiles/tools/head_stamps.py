#!/usr/bin/env python3
"""Where the fused head's time goes: builds csrc/head_fused.hip with -DAMID_HEAD_STAMPS into a DIAGNOSTIC library
(gpurun_out/libhead_diag.so; the product library carries no stamps), runs amid_head_fwd_bwd_own_vec_f32 (the folded step's head) at the headline shape (B 256, T 50,
D 128, hid 32, 2 items per row) and prints the real-time-counter (100 MHz) deltas between the
phase boundaries of workgroup 0."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libhead_diag.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAMID_HEAD_STAMPS",
                "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "amid_amd/csrc/head_fused.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
B, T, D, hid, NI = 256, 50, 128, 32, 2
g = torch.Generator().manual_seed(0)
dev = "cuda"
r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)      # noqa: E731
x = r(2 * B * T, D)
lnw, lnb = [1 + r(D), 1 + r(D)], [r(D), r(D)]
items = r(B * NI, D)
w1, b1, w2, b2 = r(hid, 2 * D), r(hid), r(hid), r(1)
labels = (torch.rand(B, NI, generator=g) < 0.5).float().to(dev)
dom = (torch.rand(B, generator=g) < 0.5).long().to(dev)
u = torch.empty(2, B, D, device=dev)
p1, p2, dp1, dp2 = (torch.zeros(B, NI, device=dev) for _ in range(4))
loss_part = torch.zeros(B, device=dev)
dx = torch.empty(2 * B * T, D, device=dev)
ditems = torch.empty(B * NI, D, device=dev)
ln_part = torch.empty(2 * B, 2, D, device=dev)
P = (hid * 2 * D + 2 * hid + 1 + 3) & ~3
sc_part = torch.empty(B, P, device=dev)
src = [r(D, D) for _ in range(24)]
dst = [torch.empty(D, D, device=dev) for _ in range(24)]
pa = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])     # noqa: E731
f = L.amid_head_fwd_bwd_own_vec_f32
L.amid_scorer_vec_floats.restype = ctypes.c_longlong
hidg = torch.empty(B, L.amid_scorer_vec_floats(NI, hid), device=dev)
vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
f.argtypes = [vp] * 10 + [ci] * 5 + [cf] + [vp] * 12 + [ci, vp]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(6):
    if it == 5:
        ev0.record()
    rc = f(x.data_ptr(), pa(lnw), pa(lnb), items.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), labels.data_ptr(),
           dom.data_ptr(), B, T, NI, D, hid, 1e-8, u.data_ptr(), p1.data_ptr(), p2.data_ptr(), dp1.data_ptr(), dp2.data_ptr(),
           loss_part.data_ptr(), dx.data_ptr(), ditems.data_ptr(), ln_part.data_ptr(), hidg.data_ptr(), None, None, 0, None)
    assert rc == 0, rc
    if it == 5:
        ev1.record()
    torch.cuda.synchronize()
print(f"launch (events, null stream): {ev0.elapsed_time(ev1) * 1e3:.1f} us")
host = (ctypes.c_ulonglong * 32)()
assert L.amid_head_stamps_read(host) == 0
names = ["entry", "W1^T staged", "LayerNorm + mean (2 barriers)", "user half", "item half (2 barriers)", "logits, loss, dLoss/dp", "loss sum (2 barriers)",
         "fence + barrier", "bwd: zero, hidden gradients", "bwd: d items, dW1 item half", "bwd: du, partial sums", "LayerNorm backward, dx"]
t = [host[i] for i in range(12)]
print(f"workgroup 0: total {(t[11] - t[0]) / 100:.2f} us; " + ", ".join(f"{names[i]} +{(t[i] - t[i - 1]) / 100:.2f}" for i in range(1, 12)))
