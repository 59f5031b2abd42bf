"""The evaluation loop alone (bench.py's `eval` object: test() at run.sh's 999 negatives, train_sr.py:31-128) -- for a rocprofv3 pass of its own:
    rocprofv3 --kernel-trace --stats -d DIR -o e -- python3 profiles/tools/eval_bench.py [n_batches]"""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import bench
from amid_amd.engine import SasrecEngine

torch.cuda.set_device(0)
eng = SasrecEngine(bench.N_ROWS, bench.D, bench.T, bench.HID, lr=5e-4, seed=1)
bench.init_params(eng, 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print(json.dumps(bench.eval_throughput(eng, "cuda:0", bench.B, bench.T, n_batches=n)))
