"""The weight-gradient launch (amid_sas_wgrad_rows_f32, 2 layers, B 256 x T 50 rows per domain, 21 splits) in its four product modes:
0 = fp32 matrix instructions, 2 / 3 = fp32 operands as three bf16 pieces each (nine / six piece pairs), 1 = operands rounded to bf16.
Error of the summed partials against the fp64 product, for unit-normal operands and for operands spread over six decades; then time.
    python profiles/tools/probe/wgrad_split_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))))
import torch
from amid_amd._lib import lib, ptr_array
L = lib()
B, T, D, splits = 256, 50, 128, 21
M = B * T
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(5)
dom = torch.randint(0, 2, (B,), device=dev, generator=g)
live = torch.cat((dom == 0, dom == 1)).float().repeat_interleave(T)
s = torch.cuda.current_stream().cuda_stream
def operands(spread):
    def one(mask):
        t = torch.randn(2 * M, D, device=dev, generator=g)
        if spread: t = t * torch.pow(10.0, -6.0 * torch.rand(2 * M, D, device=dev, generator=g))
        return t * live[:, None] if mask else t
    return [one(True) for _ in range(12)], [one(False) for _ in range(12)]
wp = [torch.empty(2, 6, splits, D * D, device=dev) for _ in range(2)]
bp = [torch.empty(2, 6, splits, D, device=dev) for _ in range(2)]
def run(dy, xx, mode):
    L.call("amid_sas_wgrad_rows_f32", ptr_array([t.data_ptr() for t in dy]), ptr_array([t.data_ptr() for t in xx]), 2, M, D, splits,
           ptr_array([t.data_ptr() for t in wp]), ptr_array([t.data_ptr() for t in bp]), dom.data_ptr(), B, T, mode, s)
for spread in (False, True):
    dy, xx = operands(spread)
    want = []
    for wsel in range(12):
        for gd in range(2):
            want.append(dy[wsel][gd * M:(gd + 1) * M].double().t() @ xx[wsel][gd * M:(gd + 1) * M].double())
    for mode in (0, 2, 3, 1):
        run(dy, xx, mode); torch.cuda.synchronize()
        emax, el2, k = 0.0, 0.0, 0
        for wsel in range(12):
            for gd in range(2):
                got = wp[wsel // 6][gd, wsel % 6].double().sum(0).view(D, D)
                w = want[k]; k += 1
                emax = max(emax, float((got - w).abs().max() / w.abs().max()))
                el2 = max(el2, float((got - w).norm() / w.norm()))
        print(f"spread={int(spread)} mode {mode}: max |err| / max |dW| = {emax:.3e}   L2 relative = {el2:.3e}")
dy, xx = operands(False)
pa = (ptr_array([t.data_ptr() for t in dy]), ptr_array([t.data_ptr() for t in xx]), ptr_array([t.data_ptr() for t in wp]), ptr_array([t.data_ptr() for t in bp]))
def launch(mode):
    L.call("amid_sas_wgrad_rows_f32", pa[0], pa[1], 2, M, D, splits, pa[2], pa[3], dom.data_ptr(), B, T, mode, s)
for mode in (0, 2, 3, 1):
    gr = torch.cuda.CUDAGraph()
    launch(mode); torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        for _ in range(20): L.call("amid_sas_wgrad_rows_f32", pa[0], pa[1], 2, M, D, splits, pa[2], pa[3], dom.data_ptr(), B, T, mode, torch.cuda.current_stream().cuda_stream)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"mode {mode}: {e0.elapsed_time(e1) / 100 * 1e3:.1f} us per launch (graph of 20)")
