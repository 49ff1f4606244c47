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
    rdv = os.environ.get("TGX_BENCH_RDV")                   # set by bench.spawn_replicas; absent under torch.distributed.run
    group = bench.FileGroup(rdv, rank, world) if rdv else bench.GlooGroup(rank, world)
    K = 8
    syncs = []
    job, mine = bench.timed_region(lambda: time.sleep(0.05 * (rank + 1)), lambda: syncs.append(time.perf_counter()), group)   # the last rank is the slow replica
    if os.environ.get("TGX_TEST_FAIL_RANK") == str(rank):
        sys.exit(3)                                          # a replica that dies after the timed region: the others must not hang in close()
    if rank == 0:
        print(json.dumps({"value": bench.aggregate_tokens_per_s(world, K, job), "max": job, "mine": mine, "n_gpus": world, "syncs": len(syncs),
                          "group": type(group).__name__}))
    group.close()
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
    assert line["n_gpus"] == 2 and line["group"] == "GlooGroup"
    assert line["max"] >= 0.1 - 1e-3                 # MAX over ranks: the job is as slow as its slowest replica
    assert line["mine"] < 0.09                       # rank 0's own region does not include waiting for rank 1
    assert abs(line["value"] - 2 * 8 / line["max"]) < 1e-9
    assert line["syncs"] == 3                       # synchronize before the barrier, after it, and after the K steps


SPAWN = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.environ["TGX_ROOT"])
    import bench
    sys.exit(bench.spawn_replicas([sys.executable, sys.argv[1]], int(sys.argv[2]), timeout_s=120.0))
""")


def _spawn(tmp_path, n, **extra_env):
    (tmp_path / "worker.py").write_text(WORKER)
    (tmp_path / "spawn.py").write_text(SPAWN)
    env = dict(os.environ, TGX_ROOT=ROOT, **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TGX_BENCH_RDV"):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(tmp_path / "spawn.py"), str(tmp_path / "worker.py"), str(n)], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_eight_self_spawned_replicas_file_barrier_and_max(tmp_path):
    """`python bench.py --gpus 8` without a launcher: bench.spawn_replicas starts the ranks, bench.FileGroup carries barrier + MAX"""
    out = _spawn(tmp_path, 8)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                           # rank 0's line only reaches stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["group"] == "FileGroup"
    assert line["max"] >= 0.4 - 1e-3                 # rank 7 slept 8 x 0.05 s
    assert line["mine"] < 0.2                        # rank 0's own region does not include waiting for the others
    assert abs(line["value"] - 8 * 8 / line["max"]) < 1e-9
    assert line["syncs"] == 3


def test_a_dead_replica_fails_the_job_instead_of_hanging_it(tmp_path):
    out = _spawn(tmp_path, 4, TGX_TEST_FAIL_RANK="2")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")] or "another replica failed" in out.stderr


def test_bench_gpus_flag_without_launcher_spawns_its_own_replicas():
    """no GPU here: each replica must fail loudly on the missing device (not fall back to anything), and the parent must report that failure"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "qwen2.5-0.5b", "--prompt", "16", "--steps", "4", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, env=env)
    import torch
    if not torch.cuda.is_available():
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert "needs a launcher" not in (out.stderr + out.stdout)
