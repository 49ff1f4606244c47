"""bench.py's N > 1 path is 'replicas only' (DESIGN.md §5): no data-path collective, gloo carries the barriers around the timed
region and the MAX-over-ranks of the per-rank elapsed times.  This runs that control plane with world_size 2 on CPU and checks the
aggregation arithmetic (value = N * K / max_r elapsed_r) by calling bench.py's own timed_region() / aggregate_tokens_per_s(); only the
per-rank work is a stand-in since no GPU is present."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    # runs bench.py's OWN timed_region / aggregate_tokens_per_s (the control plane of the N > 1 run) with a stand-in workload
    import json, os, sys, time
    sys.path.insert(0, os.environ["TGX_ROOT"])
    import torch, torch.distributed as dist
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    K = 8
    syncs = []
    job, mine = bench.timed_region(lambda: time.sleep(0.05 * (rank + 1)), lambda: syncs.append(time.perf_counter()), dist, torch)   # rank 1 is the slow replica
    if rank == 0:
        print(json.dumps({"value": bench.aggregate_tokens_per_s(world, K, job), "max": job, "mine": mine, "n_gpus": world, "syncs": len(syncs)}))
    dist.barrier(); dist.destroy_process_group()
""")


def test_two_rank_gloo_barrier_and_max(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TGX_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2
    assert line["max"] >= 0.1 - 1e-3                 # MAX over ranks: the job is as slow as its slowest replica
    assert line["mine"] < 0.09                       # rank 0's own region does not include waiting for rank 1
    assert abs(line["value"] - 2 * 8 / line["max"]) < 1e-9
    assert line["syncs"] == 3                       # synchronize before the barrier, after it, and after the K steps


def test_bench_refuses_gpus_without_launcher():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "torch.distributed.run" in (out.stderr + out.stdout)
