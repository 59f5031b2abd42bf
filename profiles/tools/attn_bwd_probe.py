"""Attention backward (SASRec shape) under different active-sequence patterns: python profiles/tools/attn_bwd_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from amid_amd._lib import lib
L = lib()
T, D, H = 50, 128, 8
st = torch.zeros(64, dtype=torch.uint8, device="cuda")
L.call("amid_step_state_pack", st.cpu().numpy().ctypes.data, 1, 7, 5e-4, 0.9, 0.999, 1e-8) if False else None
from amid_amd.engine import SasrecEngine
eng = SasrecEngine(1000, D, T, 32, lr=5e-4, seed=1)
stp = eng.step_state.data_ptr()
s = eng.s
def run(B, dom, tag):
    M = 2 * B * T
    f = lambda: torch.randn(M, D, device="cuda")
    q, k, v, o, do = f(), f(), f(), f(), f()
    stats = torch.rand(M, H, 2, device="cuda") + 0.5
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    dptr = dom.data_ptr() if dom is not None else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    with torch.cuda.stream(eng.stream):
        for _ in range(5):
            L.call("amid_attn_bwd_rows_f32", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), stats.data_ptr(), do.data_ptr(), None, B, T, D, H, 1, 0,
                   stp, 1, 0.5, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dptr, s)
        ev[0].record(eng.stream)
        for _ in range(50):
            L.call("amid_attn_bwd_rows_f32", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), stats.data_ptr(), do.data_ptr(), None, B, T, D, H, 1, 0,
                   stp, 1, 0.5, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dptr, s)
        ev[1].record(eng.stream)
    torch.cuda.synchronize()
    print(f"{tag:40s} B {B:4d}: {ev[0].elapsed_time(ev[1]) / 50 * 1e3:7.1f} us")
g = torch.Generator().manual_seed(0)
run(256, None, "all sequences")
run(128, None, "all sequences")
run(64, None, "all sequences")
run(256, (torch.rand(256, generator=g) < 0.5).long().cuda(), "random half")
run(256, (torch.arange(256) < 128).long().cuda(), "first half domain 1")
run(256, (torch.arange(256) % 2).long().cuda(), "alternating")
run(256, torch.zeros(256, dtype=torch.long).cuda(), "all domain 0")
