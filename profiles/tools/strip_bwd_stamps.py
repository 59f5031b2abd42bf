#!/usr/bin/env python3
"""Where the feed-forward backward strip kernel's cycles go: loads the DIAGNOSTIC library built with -DAMID_STRIP_STAMPS
(`bash profiles/tools/build_diag.sh` first; the product library carries no stamps), runs amid_sas_strip_ffn_bwd_f32 at the headline
shape over a live list with fp32 products (mma_bf16 = 0) and on bf16 pieces (3) and prints the s_memtime deltas between the stamp
points of workgroup 0's four waves."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = ctypes.CDLL(os.path.join(ROOT, "profiles/tools/_diag/libamid_hip_diag.so"))
B, T, D = 256, 50, 128
M = B * T
g = torch.Generator().manual_seed(0)
dev = "cuda"
act = lambda: torch.randn(2 * M, D, generator=g).to(dev)          # noqa: E731
mk = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)      # noqa: E731
dxo, h, r = act(), act().relu(), act()
tmq = torch.zeros(2 * M, D // 4, dtype=torch.uint8, device=dev)
lnw = [mk(D), mk(D)]
mats = [[mk(D, D), mk(D, D)] for _ in range(3)]
img = torch.empty(6, 3, D * D, dtype=torch.bfloat16, device=dev)
outs = [torch.empty(2 * M, D, device=dev) for _ in range(4)]
dom = (torch.rand(B, generator=g) < 0.5).long()
d0, d1 = torch.nonzero(dom == 0).flatten(), torch.nonzero(dom != 0).flatten()
live = torch.cat((d0, d1, torch.tensor([d0.numel()]))).int().to(dev)
part = torch.empty(4 * B, 2, D, device=dev)
st = torch.ones(16, dtype=torch.int64, device=dev)          # StepState: seed 1, step 1
pa = lambda t: (ctypes.c_void_p * 2)(t[0].data_ptr(), t[1].data_ptr())     # noqa: E731
vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
fi = L.amid_sas_weights_bf16_planes
fi.argtypes = [vp, ci, ci, ci, ci, vp, vp]
src = (vp * 6)(*[m[gg].data_ptr() for m in mats for gg in (0, 1)])
assert fi(src, 6, D, 0, 3, img.data_ptr(), None) == 0
f = L.amid_sas_strip_ffn_bwd_f32
f.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, ci, vp, ci, vp, ci, cf, vp, vp, vp, vp, vp, ci, vp]
names = {17: "entry, first weight request, operand loads issued", 18: "mask + dropout of dz", 19: "ring wait 1 (w2T)", 20: "product 1 (dh)",
         21: "relu' + ring wait 2 (w1T)", 22: "product 2 (dy)", 23: "LN2 backward", 24: "ring wait 3 (woT)", 25: "product 3 (d_o)",
         26: "d_o store + LN partials", 27: "barrier + partials out"}
for bf in (0, 3):
    W = (lambda k: pa(mats[k])) if bf == 0 else (lambda k: (vp * 2)(img[2 * k].data_ptr(), img[2 * k + 1].data_ptr()))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(6):
        if it == 5:
            ev[0].record()
        rc = f(dxo.data_ptr(), tmq.data_ptr(), h.data_ptr(), r.data_ptr(), pa(lnw), W(0), W(1), W(2), 1e-8, B, T, D, live.data_ptr(), 1,
               st.data_ptr(), 1, 0.5, *[o.data_ptr() for o in outs], part.data_ptr(), bf, None)
        assert rc == 0, rc
    ev[1].record()
    torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (4 * 32))()
    assert L.amid_strip_stamps_read(host) == 0
    print(f"mma_bf16 = {bf}: launch {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us")
    for w in range(4):
        t = [host[w * 32 + i] for i in range(28)]
        print(f"  wave {w}: total {t[27] - t[16]} cycles; " + ", ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(17, 28)))
        if bf == 3:          # inside the last product (strip_mma16x6's own stamps)
            u = [host[w * 32 + i] for i in (24, 28, 29, 30, 31, 25)]
            print("          product 3: " + ", ".join(f"{n} +{u[i + 1] - u[i]}" for i, n in enumerate(
                ("deferred stores + split", "pass 1 (mid, lo planes)", "barrier + next planes requested", "pass 2 (hi plane)", "drain"))))
