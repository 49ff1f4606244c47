"""The N > 1 path of bench.py end to end on a 1-GPU box: the driver's launch line (python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py
--gpus 2) with the two replicas SHARING the device (TGX_BENCH_SHARE_GPU=1).  What it exercises is everything the 8-GPU scaling run will execute first: the
rendezvous on 127.0.0.1, one context per rank, the barrier + synchronize bracket of the timed region, MAX over ranks, the aggregate on rank 0 and the one JSON line.
The number itself is a shared-GPU figure, not a scaling result (replicas only, no data-path collective)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_two_replicas_through_the_launcher_on_a_shared_gpu():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TGX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "8", "--model", "qwen2.5-0.5b", "--prompt", "64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 32 and line["warmup"] == 8 and line["scaling"] == "weak"
    assert line["config"]["replicas"] == 2
    # whole-job aggregate: both replicas' tokens over the slowest replica's time
    assert abs(line["value"] - 2 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-3
    assert line["cpu_baseline"] is None                # the CPU leg runs at N = 1 only
    assert line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1


def test_bench_two_replicas_self_spawned_on_a_shared_gpu():
    """the plain `python3 bench.py --gpus 2` shape (no launcher, no WORLD_SIZE): bench.py spawns its replicas and a file rendezvous carries barrier + MAX"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TGX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "32", "--warmup", "8", "--model", "qwen2.5-0.5b", "--prompt", "64"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 32 and line["warmup"] == 8 and line["scaling"] == "weak"
    assert line["config"]["replicas"] == 2
    assert abs(line["value"] - 2 * 1e3 / line["ms_per_step"]) / line["value"] < 1e-3
    assert line["cpu_baseline"] is None
