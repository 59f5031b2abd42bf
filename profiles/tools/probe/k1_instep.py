"""K1 inside the cfg 5 step (bench.gather_stress) for the library named by AMID_LIB_PATH."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
out = bench.gather_stress(torch.device("cuda", 0))
for k, v in out["kernels"].items():
    print(f"{k:45s} {v['avg_launch_us']:7.1f} us  {v['achieved']:7.1f} GB/s  {v['frac']}")
