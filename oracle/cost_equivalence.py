#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): is the oracle a fair stand-in for the reference as bench.py's `cpu_baseline`?

Times, on this host's CPU cores and at the headline configuration (BASELINE.json configs[1]: batch 256 x seq 50 x dim 128, hid 32,
894 820-row table, 1 negative, dropout on), one train step of

  * the REFERENCE itself: /root/reference/model_seq.py's SASRec, the loop body of train_sr.py:185-215 (forward, masked BCE,
    loss.backward(), torch.optim.Adam over every parameter including the dense table), imported with the three-line .cuda()
    no-op shim of SURVEY.md section 8(c) -- nothing of it is copied;
  * the ORACLE: oracle/amid_oracle.py train_step (the restatement bench.py times as cpu_baseline, kind "port").

Both run the same arithmetic (dense embedding gradient + dense Adam over the whole table dominate the CPU step), so the two medians
should agree within the run-to-run spread; the numbers are recorded in BASELINE.md section "Cost equivalence".

    python oracle/cost_equivalence.py [--steps 8] [--threads 8]

Needs /root/reference; never runs on the GPU box and is imported by nothing.
"""
import argparse
import json
import os
import sys
import time

import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, T, D, HID, NEG = 256, 50, 128, 32, 1
ITEM_LENGTH = 447410                      # train_sr.py:450
N_ROWS = 2 * ITEM_LENGTH                  # train_sr.py:456
PAD_ID = ITEM_LENGTH + 1


def batches(n, seed=99):
    """cloth_sport-shaped batches (short sequences, left-padded): the generator bench.py's cpu_baseline uses."""
    import bench
    gen = torch.Generator().manual_seed(seed)
    return [bench.synth_batch(gen, "cpu") for _ in range(n)]


def time_reference(bs, threads):
    torch.Tensor.cuda = lambda s, *a, **k: s
    torch.nn.Module.cuda = lambda s, *a, **k: s
    _ones = torch.ones

    def ones_nodev(*a, **k):
        k.pop("device", None)
        return _ones(*a, **k)

    torch.ones = ones_nodev
    sys.path.insert(0, REF)
    import model_seq                      # the reference's own module
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    m = model_seq.SASRec(2 * 895510, D, N_ROWS, D, T, HID, B, False, False, 0.5, 0.5)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    m.train()
    bce = torch.nn.functional.binary_cross_entropy
    ts = []
    for b in bs:
        t0 = time.perf_counter()
        opt.zero_grad()
        p1, p2 = m(None, b["i_node"], b["neg_samples"], b["seq_d1"], b["seq_d2"], None, None, True)[:2]
        p1, p2 = p1.reshape(B, -1), p2.reshape(B, -1)
        m2 = b["domain_id"].float().unsqueeze(1)
        loss = (bce(p1, b["label"], reduction="none") * (1 - m2) + bce(p2, b["label"], reduction="none") * m2).mean()   # train_sr.py:205-211
        loss.backward()
        opt.step()
        ts.append(time.perf_counter() - t0)
    torch.ones = _ones
    return ts


def time_oracle(bs, threads):
    from oracle import amid_oracle as orc
    torch.set_num_threads(threads)
    P = orc.random_params(orc.sasrec_param_shapes(N_ROWS, D, T, HID), seed=0)
    opt = orc.DenseAdam(P, lr=5e-4)
    shapes = orc.philox_masks_sasrec(1, T, D, 0, 1)
    masks = {k: (torch.rand((B,) + tuple(v.shape[1:])) >= 0.5).float() for k, v in shapes.items()}
    ts = []
    for b in bs:
        t0 = time.perf_counter()
        orc.train_step("sasrec", P, opt, b, masks)
        ts.append(time.perf_counter() - t0)
    return ts


WARMUP = 4                                # allocator, thread pool, first touch of the 1.4 GB of table + Adam state (the first
                                          # reference steps take 10-50 s here)


def median(ts):
    s = sorted(ts[WARMUP:])
    return s[len(s) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--threads", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--only", choices=("reference", "oracle"), help="internal: time one side and print its median seconds")
    a = ap.parse_args()
    if a.only:
        bs = batches(a.steps + WARMUP)
        print(median(time_reference(bs, a.threads) if a.only == "reference" else time_oracle(bs, a.threads)))
        return
    import subprocess                     # one fresh process per side: neither inherits the other's 3 GB of resident state
    side = lambda w: float(subprocess.run([sys.executable, os.path.abspath(__file__), "--only", w, "--steps", str(a.steps), "--threads",   # noqa: E731
                                           str(a.threads)], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
    t_ref, t_orc = side("reference"), side("oracle")
    print(json.dumps({"config": f"SASRec train step, batch {B} x seq {T} x dim {D}, table {N_ROWS} rows, dense Adam", "threads": a.threads,
                      "timed_steps": a.steps, "reference_ms_per_step": round(1e3 * t_ref, 1), "oracle_ms_per_step": round(1e3 * t_orc, 1),
                      "reference_samples_per_s": round(B / t_ref, 1), "oracle_samples_per_s": round(B / t_orc, 1),
                      "oracle_over_reference_time": round(t_orc / t_ref, 3)}))


if __name__ == "__main__":
    main()
