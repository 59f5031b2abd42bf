"""Standalone timing of the BERT4Rec attention core launches (csrc/attention_mfma_bert.hip) at the headline shape: forward / backward,
train (dropout counters drawn) / eval, over a live list of B sequences; HIP events around 50 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from amid_amd._lib import lib
L = lib()
B, T, D, H = int(os.environ.get("PB", "256")), int(os.environ.get("PT", "50")), 128, 4
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
M = B * T
q, k, v, d_o = (torch.randn(2 * M, D, device=dev, generator=g) for _ in range(4))
o, dq, dk, dv = (torch.zeros(2 * M, D, device=dev) for _ in range(4))
stats = torch.zeros(2 * M, H, 2, device=dev)
keep = torch.ones(B, T, dtype=torch.uint8, device=dev)
dom = (torch.rand(B, device=dev, generator=g) < 0.5).long()
live = torch.zeros(B + 1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
L.call("amid_live_list_i32", dom.data_ptr(), B, live.data_ptr(), s)
import ctypes
host = (ctypes.c_ubyte * L.value("amid_step_state_bytes"))()
L.call("amid_step_state_pack", ctypes.addressof(host), 5, 3, 5e-4, 0.9, 0.999, 1e-8)
st = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(dev)
def run(name, fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {1e3 * e0.elapsed_time(e1) / n:.1f} us per launch")
for tr in (1, 0):
    run(f"fwd live train={tr}", lambda: L.call("amid_attn_bert_fwd_live_f32", q.data_ptr(), k.data_ptr(), v.data_ptr(), keep.data_ptr(), B, T, D, H, 0,
                                               st.data_ptr(), tr, 0.1, o.data_ptr(), stats.data_ptr(), live.data_ptr(), s))
    run(f"bwd live train={tr}", lambda: L.call("amid_attn_bert_bwd_live_f32", q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), stats.data_ptr(),
                                               d_o.data_ptr(), keep.data_ptr(), B, T, D, H, 0, st.data_ptr(), tr, 0.1, dq.data_ptr(), dk.data_ptr(),
                                               dv.data_ptr(), live.data_ptr(), s))
