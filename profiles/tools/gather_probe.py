"""K1 at the cfg 5 S-uniform shape (417 792 random rows of a 10 M x 128 fp32 table): the fused gather with and without the dropout
counters, and the plain gather, timed with HIP events.   python profiles/tools/gather_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import ctypes
import torch
from amid_amd._lib import lib

L = lib()
B, T, D, NI, n_rows = 4096, 50, 128, 2, 10_000_002
dev = torch.device("cuda:0")
table = torch.empty(n_rows, D, device=dev).normal_()
pos = torch.randn(2, T, D, device=dev)
n_idx = 2 * B * T + B * NI
idx = torch.randint(0, n_rows - 2, (n_idx,), device=dev, dtype=torch.int32)
xg = torch.empty(n_idx, D, device=dev)
tmq = torch.zeros(2 * B * T, D // 4, dtype=torch.uint8, device=dev)
host = (ctypes.c_ubyte * L.value("amid_step_state_bytes"))()
L.call("amid_step_state_pack", ctypes.addressof(host), 7, 3, 5e-4, 0.9, 0.999, 1e-8)
st = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
s = torch.cuda.current_stream().cuda_stream


def timed(tag, fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    alg = n_idx * (4 + 2 * D * 4)
    print(f"{tag:46s} {us:7.1f} us  {alg / us / 1e6:.2f} TB/s algorithmic")


timed("fused gather, dropout on", lambda: L.call("amid_embed_fwd_f32", table.data_ptr(), idx.data_ptr(), pos[0].data_ptr(), pos[1].data_ptr(), B, T, D, B * NI,
                                                 xg.data_ptr(), tmq.data_ptr(), st.data_ptr(), 1, 0.5, s))
timed("fused gather, dropout off", lambda: L.call("amid_embed_fwd_f32", table.data_ptr(), idx.data_ptr(), pos[0].data_ptr(), pos[1].data_ptr(), B, T, D, B * NI,
                                                  xg.data_ptr(), tmq.data_ptr(), st.data_ptr(), 0, 0.5, s))
timed("plain gather (amid_gather_rows_f32)", lambda: L.call("amid_gather_rows_f32", table.data_ptr(), n_rows, D, idx.data_ptr(), 0, n_idx, xg.data_ptr(), None, s))
timed("torch index_select (for scale)", lambda: torch.index_select(table, 0, idx.long(), out=xg))
