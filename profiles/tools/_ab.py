#!/usr/bin/env python3
"""Prints the headline numbers of the bench lines gpurun_out/b_*.json side by side (A/B runs of one gpurun call)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "b_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = {n.replace("amid_", "").replace("_f32", ""): round(v["avg_launch_us"], 1) for n, v in d["kernels"].items()
             if any(t in n for t in ("attn", "seq_bwd", "strip", "seq_fwd", "head", "wgrad"))}
        print(os.path.basename(f), d["ms_per_step"], d["value"], d.get("loss_last"), k)
    except Exception as e:      # noqa: BLE001
        print(os.path.basename(f), "ERR", e)
