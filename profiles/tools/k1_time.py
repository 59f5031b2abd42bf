"""K1 (fused embedding forward) alone at the cfg 5 shape: batch 4096 x seq 50 over a 10 M x 128 fp32 table, uniform ids.
    python profiles/tools/k1_time.py"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from amid_amd._lib import lib
L = lib()
B, T, D, NI, N = 4096, 50, 128, 2, 10_000_002
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
table = torch.empty(N, D, device=dev).normal_(generator=g)
M = B * T
idx = torch.randint(0, N, (2 * M + B * NI,), device=dev, generator=g, dtype=torch.int32)
dom = torch.randint(0, 2, (B,), device=dev, generator=g)
live = torch.zeros(B + 1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
L.call("amid_live_list_i32", dom.data_ptr(), B, live.data_ptr(), s)
pos = torch.randn(2, T, D, device=dev)
xg = torch.empty(2 * M + B * NI, D, device=dev)
tmq = torch.empty(2 * M, D // 4, dtype=torch.uint8, device=dev)
ic, rc = torch.zeros(M + B * NI, dtype=torch.int32, device=dev), torch.zeros(M + B * NI, dtype=torch.int32, device=dev)
st = torch.zeros(L.value("amid_step_state_bytes"), dtype=torch.uint8, device=dev)
host = (ctypes.c_ubyte * L.value("amid_step_state_bytes"))()
L.call("amid_step_state_pack", ctypes.addressof(host), 7, 3, 5e-4, 0.9, 0.999, 1e-8)
st.copy_(torch.frombuffer(bytearray(host), dtype=torch.uint8))
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
rows_live, rows_full = M + B * NI, 2 * M + B * NI
for train in (1, 0):
    t = timeit(lambda: L.call("amid_embed_fwd_live_compact_f32", table.data_ptr(), idx.data_ptr(), pos[0].data_ptr(), pos[1].data_ptr(), B, T, D, B * NI,
                              xg.data_ptr(), tmq.data_ptr(), st.data_ptr(), train, 0.5, live.data_ptr(), ic.data_ptr(), rc.data_ptr(), s))
    print(f"live+compact train={train}: {t:.1f} us  {rows_live * 1040 / t / 1e6:.2f} TB/s (read + write)")
    t = timeit(lambda: L.call("amid_embed_fwd_live_f32", table.data_ptr(), idx.data_ptr(), pos[0].data_ptr(), pos[1].data_ptr(), B, T, D, B * NI,
                              xg.data_ptr(), tmq.data_ptr(), st.data_ptr(), train, 0.5, live.data_ptr(), s))
    print(f"live         train={train}: {t:.1f} us  {rows_live * 1032 / t / 1e6:.2f} TB/s")
    t = timeit(lambda: L.call("amid_embed_fwd_f32", table.data_ptr(), idx.data_ptr(), pos[0].data_ptr(), pos[1].data_ptr(), B, T, D, B * NI,
                              xg.data_ptr(), tmq.data_ptr(), st.data_ptr(), train, 0.5, s))
    print(f"full         train={train}: {t:.1f} us  {rows_full * 1032 / t / 1e6:.2f} TB/s")
out = torch.empty(rows_full, D, device=dev)
t = timeit(lambda: L.call("amid_gather_rows_f32", table.data_ptr(), N, D, idx.data_ptr(), 0, rows_full, out.data_ptr(), None, s))
print(f"plain gather (no pos / dropout / mask): {t:.1f} us  {rows_full * 1028 / t / 1e6:.2f} TB/s")
