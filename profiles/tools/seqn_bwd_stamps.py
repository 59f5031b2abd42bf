#!/usr/bin/env python3
"""Where the N-split build of the fused per-sequence backward spends its time: builds csrc/sasrec_strip.hip + csrc/sasrec_seqn_bwd.hip with
-DAMID_STRIP_STAMPS into a DIAGNOSTIC library (gpurun_out/libseqnb_diag.so; the product library carries no stamps), runs
amid_sas_seq_bwd_f32 (variant 2, then variant 1 for the launch time beside it) at the headline shape (B 256, T 50, D 128, eval-mode
dropout) and prints the real-time-counter (100 MHz) deltas between the phase boundaries of workgroup 0's eight waves."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
so = os.path.join(out, "libseqnb_diag.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAMID_STRIP_STAMPS",
                "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "amid_amd/csrc/sasrec_strip.hip"),
                os.path.join(ROOT, "amid_amd/csrc/sasrec_seqn_bwd.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
B, T, D, H = 256, 50, 128, 8
M = B * T
g = torch.Generator().manual_seed(0)
dev = "cuda"
act = lambda: (torch.randn(2 * M, D, generator=g) * 0.5).to(dev)      # noqa: E731
wt = lambda: (torch.randn(D, D, generator=g) * 0.05).to(dev)           # noqa: E731
vec = lambda: (1.0 + 0.1 * torch.randn(D, generator=g)).to(dev)        # noqa: E731
keep = []


def arr(ts):
    keep.append(ts)
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


dxo = act()
tmq = torch.zeros(2 * M, D // 4, dtype=torch.uint8, device=dev)
per_layer = {n: [act(), act()] for n in ("h", "r", "x", "q", "k", "v", "o", "dpre2", "dpre1", "dr", "dq", "dk", "dv")}
stats = [torch.stack((torch.full((2 * M, H), 4.0), torch.full((2 * M, H), 0.05)), -1).contiguous().to(dev) for _ in range(2)]
per_dom = {n: [wt() for _ in range(4)] for n in ("wq", "wk", "wv", "wo", "w1", "w2")}
lnw = {n: [vec() for _ in range(4)] for n in ("ln1", "ln2")}
ln1p = [torch.empty(2 * B, 2, D, device=dev) for _ in range(2)]
ln2p = [torch.empty(2 * B, 2, D, device=dev) for _ in range(2)]
d_o, dx = act(), act()
dom = (torch.rand(B, generator=g) < 0.5).long()
d0, d1 = torch.nonzero(dom == 0).flatten(), torch.nonzero(dom != 0).flatten()
live = torch.cat((d0, d1, torch.tensor([d0.numel()]))).int().to(dev)
f = L.amid_sas_seq_bwd_f32
vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
f.argtypes = [ci] + [vp] * 18 + [cf, ci, ci, ci, ci, vp, vp, ci, cf] + [vp] * 10 + [ci, vp]
P = per_layer
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def run():
  for it in range(6):
      if it == 5:
          ev0.record()
      rc = f(2, dxo.data_ptr(), tmq.data_ptr(), arr(P["h"]), arr(P["r"]), arr(P["x"]), arr(P["q"]), arr(P["k"]), arr(P["v"]), arr(P["o"]),
             arr(stats), arr(lnw["ln1"]), arr(lnw["ln2"]), arr(per_dom["wq"]), arr(per_dom["wk"]), arr(per_dom["wv"]), arr(per_dom["wo"]),
             arr(per_dom["w1"]), arr(per_dom["w2"]), 1e-8, B, T, D, H, live.data_ptr(), None, 0, 0.5, arr(P["dpre2"]), arr(P["dpre1"]),
             arr(P["dr"]), d_o.data_ptr(), arr(P["dq"]), arr(P["dk"]), arr(P["dv"]), dx.data_ptr(), arr(ln1p), arr(ln2p), 0, None)
      assert rc == 0, rc
      if it == 5:
          ev1.record()
      torch.cuda.synchronize()

  return ev0.elapsed_time(ev1) * 1e3


L.amid_sas_seq_bwd_variant.argtypes = [ci]
L.amid_sas_seq_bwd_variant(1)
print(f"strip build, launch (events, null stream): {run():.1f} us")
L.amid_sas_seq_bwd_variant(2)
print(f"N-split build, launch (events, null stream): {run():.1f} us")
host = (ctypes.c_ulonglong * (8 * 64))()
assert L.amid_seqnb_stamps_read(host) == 0
names = {0: "entry"}
for k, l in enumerate((1, 0)):
    sb = 1 + 8 * k
    for j, n in enumerate(("ffn product 1", "product 2", "LN2' + product 3", "barrier + attention head", "barrier + LN sums out", "qkv product 1", "product 2",
                           "product 3 + LN1'")):
        names[sb + j] = f"L{l} {n}"
names[63] = "end"
idx = sorted(names)
for w in range(8):
    t = {i: host[w * 64 + i] for i in idx}
    print(f"wave {w} (strip {w % 4}, part {w // 4}): total {(t[63] - t[0]) / 100:.2f} us; " +
          ", ".join(f"{names[i]} +{(t[i] - t[j]) / 100:.2f}" for j, i in zip(idx[:-1], idx[1:])))

fine = {10: "L0 product 2 done", 32: "barrier + exchange + ring wait", 33: "LN2' arithmetic", 34: "attention operands requested", 11: "product 3 + d_o stored", 35: "vmcnt(0) + barrier",
        36: "LN partial sums to LDS", 12: "d_o loaded + attention core", 37: "vmcnt(0)", 13: "barrier + sums out", 14: "qkv product 1", 15: "product 2", 38: "ring wait + loads requested",
        39: "product 3", 40: "exchange of dqn", 16: "LN1' arithmetic"}
order = list(fine)
for w in (0, 4):
    t = [host[w * 64 + i] for i in order]
    print(f"wave {w}, layer 0 in detail: " + ", ".join(f"{fine[order[k]]} +{(t[k] - t[k - 1]) / 100:.2f}" for k in range(1, len(order))))
