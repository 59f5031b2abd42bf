#!/usr/bin/env python3
"""Cost of the lazy-Adam catch-up (amid_lazy_adam_catchup_positions_f32) against the gap of the lagging rows: cfg 4's list shape (10 752
positions, 89 % of them the pad row, ~1 500 distinct lagging rows), every lagging row `gap` steps behind.  Prints microseconds per launch."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from amid_amd._lib import lib  # noqa: E402

L = lib()
D, n_rows, n_idx, n_lag = 128, 200_000, 10_752, 1_500
g = torch.Generator().manual_seed(0)
table = torch.randn(n_rows, D, generator=g).cuda()
m0 = (torch.randn(n_rows, D, generator=g) * 1e-3).cuda()
v0 = (torch.rand(n_rows, D, generator=g) * 1e-5 + 1e-8).cuda()
rows = torch.randperm(n_rows - 1, generator=g)[:n_lag]
pos = torch.full((n_idx,), n_rows - 1, dtype=torch.int32)
pos[torch.randperm(n_idx, generator=g)[:n_lag]] = rows.int()
pos = pos.cuda()
host = (ctypes.c_ubyte * L.value("amid_step_state_bytes"))()
s = torch.cuda.current_stream().cuda_stream
for gap in (0, 8, 30, 60, 120, 200, 300, 480, 1000):
    t = 5000
    L.call("amid_step_state_pack", ctypes.addressof(host), 0, t, 5e-4, 0.9, 0.999, 1e-8)
    st = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    ts = []
    for it in range(6):
        tab, m, v = table.clone(), m0.clone(), v0.clone()
        last = torch.full((n_rows,), t - 1, dtype=torch.int32, device="cuda")
        last[rows.cuda()] = t - 1 - gap
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("amid_lazy_adam_catchup_positions_f32", tab.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(), pos.data_ptr(), n_idx, D,
               st.data_ptr(), s)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"gap {gap:5d}: {sorted(ts)[len(ts) // 2]:7.1f} us")
