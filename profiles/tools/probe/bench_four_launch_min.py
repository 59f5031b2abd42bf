"""Probe: bench.py with the index sort's four-launch implementation taken from FOUR_MIN indices on (default 65536):
    FOUR_MIN=0 python profiles/tools/probe/bench_four_launch_min.py --workload cfg3 --no-stress --no-cpu-baseline"""
import os, sys, runpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
torch.cuda.init()
torch.zeros(1, device="cuda")
from amid_amd._lib import lib
v = int(os.environ.get("FOUR_MIN", "65536"))
print("prev four-launch min:", lib().value("amid_sort_set_four_launch_min", v), "->", v, file=sys.stderr)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
