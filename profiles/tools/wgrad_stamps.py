#!/usr/bin/env python3
"""Where the weight-gradient launch's time goes: builds csrc/sasrec_bwd.hip with -DAMID_WGRAD_STAMPS into a DIAGNOSTIC library
(gpurun_out/libwgrad_diag.so), runs amid_sas_wgrad_rows_f32 at the headline shape (2 layers, M = 256 x 50 rows per domain, 21 splits,
the domain hint) and prints the real-time-counter (100 MHz) deltas of workgroup (0, 0, 0)."""
import ctypes
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libwgrad_diag.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAMID_WGRAD_STAMPS",
                "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "amid_amd/csrc/sasrec_bwd.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
B, T, D, S = 256, 50, 128, 21
M = B * T
g = torch.Generator().manual_seed(0)
dev = "cuda"
dy = [(torch.randn(2 * M, D, generator=g) * 0.1).to(dev) for _ in range(12)]
xx = [(torch.randn(2 * M, D, generator=g) * 0.1).to(dev) for _ in range(12)]
wpart = [torch.empty(2, 6, S, D * D, device=dev) for _ in range(2)]
bpart = [torch.empty(2, 6, S, D, device=dev) for _ in range(2)]
dom = (torch.rand(B, generator=g) < 0.5).long().to(dev)
pa = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])     # noqa: E731
f = L.amid_sas_wgrad_rows_f32
vp, ci = ctypes.c_void_p, ctypes.c_int
f.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, ci, ci, ci, vp]
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(6):
    if it == 5:
        ev0.record()
    rc = f(pa(dy), pa(xx), 2, M, D, S, pa(wpart), pa(bpart), dom.data_ptr(), B, T, 0, None)
    assert rc == 0, rc
    if it == 5:
        ev1.record()
    torch.cuda.synchronize()
print(f"launch (events, null stream): {ev0.elapsed_time(ev1) * 1e3:.1f} us")
host = (ctypes.c_ulonglong * 64)()
assert L.amid_wgrad_stamps_read(host) == 0
t = list(host)
print(f"workgroup (0,0,0): hint prologue +{(t[1] - t[0]) / 100:.2f}, first chunk requested +{(t[2] - t[1]) / 100:.2f}, "
      f"chunk loop +{(t[40] - t[2]) / 100:.2f}, accumulators stored + bias sums +{(t[41] - t[40]) / 100:.2f}; total {(t[41] - t[0]) / 100:.2f} us")
sched = (ctypes.c_ulonglong * 2048)()
assert L.amid_wgrad_sched_read(sched) == 0
n = S * 12 * 2
wgs = list(range(n))
t00 = min(sched[2 * i] for i in wgs)
st = sorted((sched[2 * i] - t00) / 100 for i in wgs)
du = sorted((sched[2 * i + 1] - sched[2 * i]) / 100 for i in wgs)
en = max(sched[2 * i + 1] for i in wgs) - t00
dec = lambda v: [round(v[min(len(v) - 1, k * len(v) // 10)], 1) for k in range(11)]     # noqa: E731
print(f"{n} workgroups: kernel span {en / 100:.1f} us; start times (us), deciles: {dec(st)}; durations (us), deciles: {dec(du)}")
for z in (0, 1):
    v = [((sched[2 * i + 1] - sched[2 * i]) / 100, (sched[2 * i + 1] - t00) / 100) for i in wgs if i // (S * 12) == z]
    print(f"domain {z} (dispatched {'first' if z == 0 else 'second'}): mean duration {sum(a for a, _ in v) / len(v):.1f} us, mean end {sum(b for _, b in v) / len(v):.1f} us")
