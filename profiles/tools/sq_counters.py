#!/usr/bin/env python3
"""Summarise one rocprofv3 --pmc pass of SQ counters (profiles/tools/sq_counters.sh) per kernel, as markdown.

Units (MI355X_MICROARCH.md, 'rocprofv3 PMC slots' + constants table; calibrated against kernels whose MFMA count is known by
construction): SQ_VALU_MFMA_BUSY_CYCLES here sums, over the launch, 4 x the cycles a SIMD's matrix pipe is held (one
v_mfma_f32_16x16x4_f32 = 8 passes = 32 cycles of its SIMD adds 128); SQ_BUSY_CU_CYCLES sums the cycles each CU has a wave resident;
SQ_INSTS_VALU_MFMA_MOPS_F32 counts executed fp32 MFMA work in units of 512 FLOP.  So
    matrix-pipe busy fraction of the resident time = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)
    executed matrix FLOP                          = 512 x SQ_INSTS_VALU_MFMA_MOPS_F32   (against the algorithmic FLOP of bench.py)
SQ_LDS_BANK_CONFLICT = extra LDS-array cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles; SQ_WAIT_ANY / SQ_WAVE_CYCLES = share of the
waves' resident time parked at s_waitcnt / barriers."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
files = glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for row in csv.DictReader(open(files[0])):
    k = row["Kernel_Name"]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
bench = None
try:
    bench = json.loads([ln for ln in open(os.path.join(out, "sq_bench.json")) if ln.startswith("{")][-1])
except Exception:
    pass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
algo = {}
if bench:
    import bench as B
    cfg = bench["config"]
    work = B.algorithmic_work(cfg["batch_per_gpu"], cfg.get("unique_rows_last_step"), cfg["seq_len"])
    for entry, (kind, amount) in work.items():
        sym = B.KERNEL_SYMBOL.get(entry)
        if sym and kind.startswith("mfma") and entry in bench.get("kernels", {}):
            for one in (sym if isinstance(sym, tuple) else (sym,)):
                algo[one] = (entry, amount)
print("| kernel | launches | MFMA busy / (4 x CU busy) | executed matrix GFLOP / launch (512 x (MOPS_F32 + MOPS_BF16)) | algorithmic GFLOP / launch | executed / algorithmic | "
      "LDS conflict cycles / LDS active cycles | WAIT_ANY / WAVE_CYCLES |")
print("|---|---|---|---|---|---|---|---|")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
    if not (k.startswith("void amid") or k.startswith("amid")):
        continue
    m = max(n[k], 1)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(4.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0), 1.0)
    gf = 512.0 * (c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0)) / m / 1e9
    al = next(((e, a) for s, (e, a) in algo.items() if s in k), None)
    ratio = f"{gf / (al[1] / 1e9):.2f}" if al and gf > 0 else "-"
    algs = f"{al[1] / 1e9:.3f} (`{al[0]}`)" if al else "-"
    lds = f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / m:.3g} / {c.get('SQ_LDS_IDX_ACTIVE', 0) / m:.3g}"
    wait = c.get("SQ_WAIT_ANY", 0.0) / max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print(f"| `{k[:90]}` | {m} | {busy:.3f} | {gf:.3f} | {algs} | {ratio} | {lds} | {wait:.2f} |")
