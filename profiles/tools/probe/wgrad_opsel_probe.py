#!/usr/bin/env python3
"""Which launches of the weight-gradient kernel built with -DAMID_WGS_ZERO_BY_MUL (rows past a split's end zeroed by a 0 / 1 multiplication;
profiles/tools/probe/wgrad_opsel_repro.sh builds the library) come out wrong?  For a range of split counts the 24 summed gradients of
the train step's launch shape (B 256, T 50, live-row hint, mode 3 = six piece pairs) against the fp64 product, with the parity of the rows
per split beside them: a thread stages a PAIR of rows (2 rr, 2 rr + 1), so the two rows of a pair get different factors exactly when a
split ends on an odd row.    AMID_LIB_PATH=profiles/tools/_diag/libamid_hip_zeromul.so python profiles/tools/probe/wgrad_opsel_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from amid_amd._lib import lib, ptr_array      # noqa: E402

L = lib()
B, T, D = 256, 50, 128
M = B * T
g = torch.Generator().manual_seed(99)
dom = (torch.rand(B, generator=g) < 0.5).long()
live = torch.cat((dom == 0, dom == 1)).float().repeat_interleave(T)
dy = [(torch.randn(2 * M, D, generator=g) * live[:, None]).cuda() for _ in range(12)]
xx = [torch.randn(2 * M, D, generator=g).cuda() for _ in range(12)]
want = [[dy[wi][gd * M:(gd + 1) * M].double().t() @ xx[wi][gd * M:(gd + 1) * M].double() for gd in range(2)] for wi in range(12)]
n_live = [int((dom == 0).sum()), int((dom == 1).sum())]
print(f"library {os.environ.get('AMID_LIB_PATH', '(product)')}; live sequences per domain {n_live}")
for splits in (8, 10, 16, 20, 21, 25, 32, 40):
    wp = [torch.full((2, 6, splits, D * D), float("nan"), device="cuda") for _ in range(2)]
    bp = [torch.full((2, 6, splits, D), float("nan"), device="cuda") for _ in range(2)]
    L.call("amid_sas_wgrad_rows_f32", ptr_array([t.data_ptr() for t in dy]), ptr_array([t.data_ptr() for t in xx]), 2, M, D, splits,
           ptr_array([t.data_ptr() for t in wp]), ptr_array([t.data_ptr() for t in bp]), dom.cuda().data_ptr(), B, T, 3,
           torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    rps = [-(-n * T // splits) for n in n_live]
    errs = [[float((wp[wi // 6][gd, wi % 6].sum(0).view(D, D).double() - want[wi][gd]).abs().max() / want[wi][gd].abs().max()) for gd in range(2)]
            for wi in range(12)]
    worst = [max(errs[wi][gd] for wi in range(12)) for gd in range(2)]
    bad = [sum(errs[wi][gd] > 2e-6 for wi in range(12)) for gd in range(2)]
    print(f"splits {splits:3d}: rows per split (domain 0, 1) = {rps} ({'odd' if rps[0] % 2 else 'even'}, {'odd' if rps[1] % 2 else 'even'}); "
          f"worst error {worst[0]:.2e}, {worst[1]:.2e}; wrong tiles {bad[0]} / 12, {bad[1]} / 12")
