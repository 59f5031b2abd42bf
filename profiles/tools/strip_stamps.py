#!/usr/bin/env python3
"""Where a strip kernel's cycles go: builds csrc/sasrec_strip.hip with -DAMID_STRIP_STAMPS into a DIAGNOSTIC library of its own
(gpurun_out/libstrip_diag.so; the product library carries no stamps), runs amid_sas_strip_qkv_fwd_f32 at the headline shape over a
live list and prints the s_memtime deltas between the stamp points of workgroup 0's four waves."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libstrip_diag.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAMID_STRIP_STAMPS",
                "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "amid_amd/csrc/sasrec_strip.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
B, T, D = 256, 50, 128
M = B * T
g = torch.Generator().manual_seed(0)
dev = "cuda"
x = torch.randn(2 * M, D, generator=g).to(dev)
mk = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)      # noqa: E731
lnw, lnb, w_in, b_in = [mk(D), mk(D)], [mk(D), mk(D)], [mk(3 * D, D), mk(3 * D, D)], [mk(3 * D), mk(3 * D)]
outs = [torch.empty(2 * M, D, device=dev) for _ in range(4)]
dom = (torch.rand(B, generator=g) < 0.5).long()
d0, d1 = torch.nonzero(dom == 0).flatten(), torch.nonzero(dom != 0).flatten()
live = torch.cat((d0, d1, torch.tensor([d0.numel()]))).int().to(dev)
pa = lambda t: (ctypes.c_void_p * 2)(t[0].data_ptr(), t[1].data_ptr())     # noqa: E731
vp = ctypes.c_void_p
f = L.amid_sas_strip_qkv_fwd_f32
f.argtypes = [vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]
for it in range(5):
    rc = f(x.data_ptr(), pa(lnw), pa(lnb), pa(w_in), pa(b_in), 1e-8, B, T, D, live.data_ptr(), *[o.data_ptr() for o in outs], None)
    assert rc == 0, rc
    torch.cuda.synchronize()
host = (ctypes.c_ulonglong * (4 * 32))()
assert L.amid_strip_stamps_read(host) == 0
names = ["entry", "row", "loads issued", "LN done", "ring.next 1", "mma 1", "epi 1", "ring.next 2", "mma 2", "ring.next 3", "mma 3", "end"]
for w in range(4):
    t = [host[w * 32 + i] for i in range(12)]
    print(f"wave {w}: total {t[11] - t[0]} cycles; " + ", ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 12)))
