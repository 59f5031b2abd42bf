"""Randomised sweep of the timed path against the oracle (tests/test_gpu_timed_path.py::_timed_vs_oracle): FUZZ_N random (B, T, D, domain
split, fused backward forced on / off / auto) combinations with T from 1 to 64 and D in {64, 128}; prints the failures and a count.  Not part of
the test suite (minutes of oracle time): run by hand on the GPU box after kernel changes -- FUZZ_SEED / FUZZ_N select the draw."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_timed_path as tp
random.seed(int(os.environ.get("FUZZ_SEED", "1")))
n_ok = 0
cases = []
for _ in range(int(os.environ.get("FUZZ_N", "40"))):
    D = random.choice([64, 128])
    T = random.choice([1, 2, 7, 15, 16, 17, 20, 31, 32, 33, 40, 47, 48, 49, 50, 63, 64])
    B = random.choice([1, 3, 37, 64, 130, 256, 300])
    split = random.choice(["mixed", "mixed", "all0", "all1", "one0"])
    force = random.choice([None, "1", "0"])
    cases.append((B, T, D, split, force))
for (B, T, D, split, force) in cases:
    try:
        tp._timed_vs_oracle(B, T, D, None, split, compact_min=None, seq_backward=force)
        n_ok += 1
    except Exception as e:
        print("FAIL", (B, T, D, split, force), type(e).__name__, str(e)[:300], flush=True)
print(f"fuzz: {n_ok} / {len(cases)} ok", flush=True)
