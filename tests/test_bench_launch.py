"""bench.py's N > 1 entry: started WITHOUT a launcher it must start its own ranks as child processes (never exec, never from a process
that has initialised the GPU), relay rank 0's JSON line and exit with the launcher's code; started under a launcher it is one rank.
CPU part: the command it builds.  GPU part (-m gpu): the whole thing with two ranks folded onto the one test GPU over gloo, both
dense exchanges -- the only way to execute the N > 1 control flow of the bench on a single-GPU box (RCCL refuses two ranks on one
device).  The reference has no multi-GPU path to mirror (train_sr.py:473: DataParallel commented out)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_without_launcher_starts_its_own_ranks(monkeypatch):
    import torch
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        seen["cuda_initialised"] = torch.cuda.is_initialized()

        class R:
            returncode = 7
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "8", "--warmup", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the launcher's exit code is this process's
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "8", "--warmup", "4"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert not seen["cuda_initialised"]                        # the parent has not touched the GPU when it starts the children


def test_bench_refuses_a_world_size_that_contradicts_gpus(monkeypatch):
    import bench
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE" in str(e.value.code)


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("dense", ["gather", "allreduce"])
def test_bench_two_ranks_self_launched_on_one_gpu(dense):
    env = dict(os.environ, AMID_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "4", "--no-cpu-baseline",
                        "--dense-exchange", dense], env=env, cwd=ROOT, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 8 and out["warmup"] == 4
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"] and out["config"]["parallelism"] == "dp2"
    d = out["dist"]
    assert d["world_size"] == 2 and d["dense_exchange"] == dense
    assert d["collectives_per_step"] == (1 if dense == "gather" else 2)
    assert d["bytes_sent_per_step_per_rank"] > 0 and d["bytes_received_per_step_per_rank"] > 0
    # gather: every rank receives the world's dense copies behind the sparse rows; allreduce: one dense gradient's worth
    want = d["sparse_chunk_bytes_per_rank"] * 2 + d["dense_bytes_received_per_step"][dense]
    assert abs(d["bytes_received_per_step_per_rank"] - want) <= 2 * 4 * 128 * 2, (d, want)
    assert out["value"] > 0 and out["roofline"]["frac"] > 0


def test_driver_arguments_keep_four_step_graphs():
    """The driver runs `bench.py --steps 20 --warmup 5`: the timed region must replay the graphs the CLI's train loop replays
    (amid_amd/train_sr.py STEPS_PER_GRAPH = 4), the warm-up is rounded up to whole graphs instead of shrinking the graph."""
    import bench
    from amid_amd.train_sr import STEPS_PER_GRAPH
    assert STEPS_PER_GRAPH == 4
    assert bench.steps_per_graph_for(4, 20, True) == 4          # the driver's arguments
    assert bench.steps_per_graph_for(4, 200, True) == 4         # the defaults
    assert bench.steps_per_graph_for(4, 10, True) == 2          # K not a multiple: the largest divisor below
    assert bench.steps_per_graph_for(4, 7, True) == 1
    assert bench.steps_per_graph_for(4, 20, False) == 1         # eager / copy input / N > 1: single steps
