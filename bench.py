#!/usr/bin/env python3
"""bench.py — decode tokens/s of the MI355X path on BASELINE.json's headline configuration.

Workload (BASELINE.json configs[2], the one `metric` is quoted on): Llama-3.2-1B, bf16 weights + bf16 KV
cache, batch 1, 2048-token prefill, then greedy decode.  A *step* is one decoded token: all 16 decoder
layers + final norm + lm_head over the whole vocabulary + argmax + the next token's embedding gather,
with token id and position resident on the GPU (one hipGraph replay per step).  Weights are synthetic
(tinygpt_amd.synth, seed 1234; no checkpoint exists offline) and prompt ids are uniform — timing is
data-independent; parity is established by tests/.

  python bench.py [--gpus N] [--steps K] [--warmup W]                       (N > 1: spawns its own N replica processes)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W                            (the launcher's N processes are the replicas)

N > 1 runs N independent replicas (one process per GPU, distinct prompt seeds, no data-path collective —
the streams never exchange data).  Both invocation shapes work: under a launcher torch.distributed
(gloo) carries the barrier and the MAX-over-ranks of the timed region; without one (no WORLD_SIZE in the
environment) `spawn_replicas` starts the N processes itself and a directory of small files carries the same two
things (`FileGroup`) — replicas need no rendezvous service.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     the dominant kernel class (gate_up GEMV: 55 % of a layer's bytes): algorithmic bytes per launch
               (2*N*K) / its average duration measured with HIP events on the launch stream
               (tgx_profile_decode: the class launched back-to-back over all layers); HBM peak 8.0 TB/s.
               Also carries the whole-step figure (`step_*`): bytes_per_token(T) * tokens/s.
  cpu_baseline the CPU oracle (oracle/liboracle.so, a restatement of the reference path — kind "port"), timed on this box's host
               cores with pinned threads: median of 3 bounded samples at the GPU run's own context (after the same 2048-token
               prompt) and at a short context; `legs` carries both with min / max.
"""
import argparse
import json
import os
import statistics
import sys
import time



def _host_cpu_budget():
    """(hardware threads this process may run on, CPUs' worth of time its cgroup grants) — read BEFORE any OpenMP runtime starts: with OMP_PROC_BIND set
    libgomp binds the initial thread to its place, after which sched_getaffinity reports that one core"""
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                 # cgroup v2: "<quota us> <period us>" or "max <period>"
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    return usable, quota


HOST_USABLE, HOST_CPU_QUOTA = _host_cpu_budget()

# the cpu_baseline leg is OpenMP code: pin its threads, one per core, SPREAD over the places (a team of 64 on a 2 x 64-core host takes every other core
# of both sockets: all memory channels) — set before anything loads libgomp.  The oracle places every weight row on the NUMA node of the thread that
# streams it (first touch in the product loop's own partition, oracle/tgx_oracle.c mat_store_rows).
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="llama-3.2-1b", help="key of tinygpt_amd.desc.KNOWN_CONFIGS")
    ap.add_argument("--model-dir", default=None, help="a real HF directory (config.json + model.safetensors or its sharded form, ModelLoader.cpp:18-89): "
                    "its hyper-parameters and weights replace the synthetic ones; `data` then reads \"checkpoint\"")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"], help="storage dtype of parameters and KV cache (the headline is bf16)")
    ap.add_argument("--prompt", type=int, default=2048, help="prefill length before the timed decode")
    ap.add_argument("--profile-reps", type=int, default=8)
    ap.add_argument("--no-graph", action="store_true", help="eager launches (needed under rocprofv3 kernel tracing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(desc, tensors, seconds, prompt_len, prompt_seed, max_prefill_s=90.0):
    """Oracle decode tok/s on the host cores (rank 0, N = 1 only).  Two legs, each the median of 3 samples:
      short    16-token prompt (context ~20..) — cheap, always measured;
      same     the GPU run's own prompt length (2048 for the headline), i.e. the same workload; skipped (and said so) when the
               oracle's prefill of that prompt would take longer than `max_prefill_s` on this host.
    `value` is the same-context figure when it was measured, else the short one; `sample` says which."""
    from oracle.oracle_ffi import OracleModel, build_oracle, oracle_backend
    from tinygpt_amd import synth
    from tinygpt_amd.ffi import GREEDY
    build_oracle()
    cores = os.cpu_count() or 1
    usable, quota = HOST_USABLE, HOST_CPU_QUOTA
    be = oracle_backend()
    ids = synth.synth_prompt(desc.vocab, 16, prompt_seed)[None, :]

    def load(n_thr):
        """a context whose weights were uploaded BY a team of n_thr threads: every page sits on the node of the thread that will stream it"""
        be.set_threads(n_thr)
        mm = OracleModel(desc)
        for name, bits in tensors:
            mm.upload(name, bits)
        return mm.finalize()

    # Team size: the port is memory-bound; what a team is worth is decided by where its pages are, so every candidate gets its own upload.  Up to round 4
    # the weights were first-touched by ONE thread (a serial memcpy) and every team streamed them through one socket's quadrant: 78 GB/s at best, and wide
    # teams collapsed (16 threads 40 tok/s, 64 threads 16).  The probe table stays on the line (`legs.probe`).
    # The GPU boxes of this pool grant the container a CFS quota (cpu.max 1600000 100000 = 16 CPUs' worth of time on a 2 x 64-core host): a team wider than
    # the quota is throttled, not faster — that, not libgomp, is why "all 256 host threads" collapses.  The probe brackets the quota.
    budget = usable if quota is None else max(1, min(usable, int(round(quota))))
    cands = sorted({t for t in (max(1, budget // 2), budget, budget + budget // 2, 2 * budget, 64, 128) if t <= usable} or {usable})
    probe, best_n, best_rate, m = {}, 1, 0.0, None
    for n_thr in cands:
        mm = load(n_thr)
        t0 = time.perf_counter(); mm.forward(ids); p16 = time.perf_counter() - t0
        mm.sample(GREEDY); mm.decode(1, GREEDY)
        t0 = time.perf_counter(); mm.decode(3, GREEDY); r = 3 / (time.perf_counter() - t0)
        probe[str(n_thr)] = {"tok_s": round(r, 2), "prefill16_s": round(p16, 3)}
        if r > best_rate:
            if m is not None:
                m.close()
            best_n, best_rate, m, prefill16_s = n_thr, r, mm, p16
        else:
            mm.close()
        if r < 0.6 * best_rate or (quota is not None and n_thr >= 2 * budget):          # wider teams only lose from here on
            break
    be.set_threads(best_n)

    def three_samples(budget_s, rate_guess):
        n = int(max(4, min(96, budget_s / 3.0 * rate_guess)))
        rates = []
        for _ in range(3):
            t0 = time.perf_counter(); m.decode(n, GREEDY); rates.append(n / (time.perf_counter() - t0))
        return n, rates

    n_s, r_short = three_samples(seconds / 2.0, best_rate)
    ctx_short = (20, 20 + 16 + 3 * n_s)
    legs = {"short": {"prompt_tokens": 16, "context": list(ctx_short), "tokens_per_sample": n_s, "tok_s_median": round(statistics.median(r_short), 3),
                      "tok_s_min": round(min(r_short), 3), "tok_s_max": round(max(r_short), 3)}}
    value, which = statistics.median(r_short), "short"
    est_prefill = prefill16_s / 16.0 * prompt_len          # measured on 16 tokens with the chosen team
    if prompt_len > 16 and prompt_len + 3 * 96 + 8 <= desc.max_ctx:
        if est_prefill <= max_prefill_s:
            m.reset_cache()
            t0 = time.perf_counter(); m.forward(synth.synth_prompt(desc.vocab, prompt_len, prompt_seed)[None, :]); pre_s = time.perf_counter() - t0
            m.sample(GREEDY)
            m.decode(1, GREEDY)
            n_l, r_same = three_samples(seconds / 2.0, statistics.median(r_short))
            legs["same"] = {"prompt_tokens": prompt_len, "context": [prompt_len + 2, prompt_len + 2 + 3 * n_l], "tokens_per_sample": n_l,
                            "tok_s_median": round(statistics.median(r_same), 3), "tok_s_min": round(min(r_same), 3), "tok_s_max": round(max(r_same), 3),
                            "oracle_prefill_s": round(pre_s, 1)}
            value, which = statistics.median(r_same), "same"
        else:
            legs["same"] = {"skipped": f"oracle prefill of {prompt_len} tokens estimated at {est_prefill:.0f} s > {max_prefill_s:.0f} s on this host"}
    m.close()
    ctx = legs[which]["context"]
    legs["probe"] = probe
    return {"value": round(value, 3), "unit": "tokens/s", "cores": best_n, "host_cores": cores, "host_cores_usable": usable, "cgroup_cpu_quota": quota, "kind": "port",
            "omp": {"OMP_PROC_BIND": os.environ.get("OMP_PROC_BIND"), "OMP_PLACES": os.environ.get("OMP_PLACES"), "threads": best_n},
            "legs": legs,
            "sample": f"oracle/liboracle.so (C+OpenMP restatement: an UNTUNED loop nest — plain fp32 loops, no blocking, no SIMD intrinsics; a stated baseline, not a tuned CPU "
                      f"implementation, so the GPU/CPU ratio says nothing about kernel quality), same synthetic {desc.name or 'model'} {desc.compute_dtype}; value = median of 3 samples of "
                      f"{legs[which]['tokens_per_sample']} greedy decode tokens after a {legs[which]['prompt_tokens']}-token prompt (context {ctx[0]}..{ctx[1]}: "
                      + ("the GPU run's own context" if which == "same" else "SHORTER than the GPU run's context — the oracle's prefill of the full prompt was over budget")
                      + f"), {best_n} of {cores} host threads (the container's CFS quota grants {quota if quota is not None else 'all'} CPUs; best team size of a probe over {cands[0]}..{cands[-1]}, each with its own NUMA-local upload), threads pinned (OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, OMP_PLACES=cores); "
                      f"spread min/max in legs"}


def kernel_source_sha256():
    """sha256 over the sources of the roofline kernel (the GEMV and the helpers it is built from): what ties profiles/pmc_traffic.json to a build"""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemv.h", "common.h"):
        with open(os.path.join(ROOT, "tinygpt_amd", "csrc", "kernels", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


class GlooGroup:
    """barrier + MAX over the launcher's ranks through torch.distributed (gloo): control plane only"""

    def __init__(self, rank, world):
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        self.rank, self.world, self._dist, self._torch = rank, world, dist, torch

    def barrier(self):
        self._dist.barrier()

    def max(self, v):
        t = self._torch.tensor([v], dtype=self._torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self._dist.barrier()
        self._dist.destroy_process_group()


class FileGroup:
    """The same two operations for replicas that `spawn_replicas` started: a directory the parent created, one small file per (generation, rank).
    A barrier = write my file of this generation, poll until all `world` files of it exist (200 us poll: the skew it leaves between ranks is
    three orders below the timed region).  No sockets, no hostname lookups, no torch.distributed."""

    def __init__(self, root, rank, world, timeout_s=1800.0):
        self.root, self.rank, self.world, self.gen, self.timeout_s = root, rank, world, 0, timeout_s

    def _exchange(self, payload):
        g = self.gen
        self.gen += 1
        mine = os.path.join(self.root, f"g{g}.r{self.rank}")
        with open(mine + ".tmp", "w") as f:
            f.write(payload)
        os.rename(mine + ".tmp", mine)                       # atomic: a reader never sees a half-written value
        paths = [os.path.join(self.root, f"g{g}.r{r}") for r in range(self.world)]
        deadline = time.monotonic() + self.timeout_s
        while not all(os.path.exists(q) for q in paths):
            if os.path.exists(os.path.join(self.root, "abort")) or time.monotonic() > deadline:
                raise RuntimeError(f"replica {self.rank}: another replica failed or timed out at barrier {g}")
            time.sleep(2e-4)
        return [open(q).read() for q in paths]

    def barrier(self):
        self._exchange("")

    def max(self, v):
        return max(float(x) for x in self._exchange(repr(float(v))))

    def close(self):
        self.barrier()


def timed_region(run_steps, sync, group=None):
    """The contract's timed region, used by the real run below and (with a stand-in `run_steps`) by tests/test_replicas_gloo.py:
    barrier + synchronize | EXACTLY the K steps of this rank | synchronize; then barrier + MAX over ranks.  No collective sits inside the
    timed region — the replicas never exchange data.  `group` is a GlooGroup / FileGroup (None at N = 1).
    Returns (elapsed of the job = slowest replica, this rank's own)."""
    sync()
    if group:
        group.barrier()
    sync()
    t0 = time.perf_counter()
    run_steps()
    sync()
    mine = time.perf_counter() - t0
    job = mine
    if group:
        group.barrier()
        job = group.max(mine)                                       # MAX over ranks: the job is as slow as its slowest replica
    return job, mine


def spawn_replicas(cmd, n, env=None, timeout_s=3600.0):
    """`python bench.py --gpus N` without a launcher: start the N replica processes (RANK = LOCAL_RANK = r, WORLD_SIZE = N, TGX_BENCH_RDV = a fresh
    directory for FileGroup), pass rank 0's stdout through (its one JSON line), send the other ranks' stdout to stderr, and return the worst exit code.
    A replica that dies drops an `abort` file so the others leave their barrier instead of waiting out the timeout."""
    import shutil
    import subprocess
    import tempfile
    rdv = tempfile.mkdtemp(prefix="tgx_bench_rdv_")
    procs = []
    try:
        for r in range(n):
            e = dict(os.environ if env is None else env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), TGX_BENCH_RDV=rdv)
            procs.append(subprocess.Popen(cmd, env=e, stdout=None if r == 0 else sys.stderr))
        deadline, rc, live = time.monotonic() + timeout_s, 0, set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = rc or code
                        open(os.path.join(rdv, "abort"), "w").close()
            if time.monotonic() > deadline:
                open(os.path.join(rdv, "abort"), "w").close()
                rc = rc or 124
                break
            time.sleep(0.02)
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=10)
                except Exception:
                    p.kill()                                        # the exact child this function started
        shutil.rmtree(rdv, ignore_errors=True)


def aggregate_tokens_per_s(world, steps, job_elapsed):
    """whole-job throughput of N replicas that each decoded `steps` tokens"""
    return world * steps / job_elapsed


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:           # no launcher: be the launcher (replicas need no rendezvous service)
        sys.exit(spawn_replicas([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world                                             # a launcher's process count wins over the flag

    import torch
    group = None
    if world > 1:
        rdv = os.environ.get("TGX_BENCH_RDV")
        group = FileGroup(rdv, rank, world) if rdv else GlooGroup(rank, world)      # control plane only: barrier + MAX
    if os.environ.get("TGX_BENCH_SHARE_GPU") == "1" and torch.cuda.is_available():   # testing the N > 1 flow on a 1-GPU box: ranks share devices
        local_rank %= torch.cuda.device_count()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    from tinygpt_amd import known_desc, synth
    from tinygpt_amd.ffi import GREEDY, Model, product_backend

    if args.model_dir:
        from tinygpt_amd.checkpoint import iter_checkpoint
        from tinygpt_amd.desc import load_desc
        desc = load_desc(args.model_dir, args.dtype)
        if not desc.name:
            desc.name = os.path.basename(os.path.normpath(args.model_dir))
    else:
        desc = known_desc(args.model, args.dtype)
    need_ctx = args.prompt + args.warmup + args.steps + 8
    if need_ctx > desc.max_ctx:
        sys.exit(f"prompt+warmup+steps = {need_ctx} exceeds contextSize {desc.max_ctx}")
    want_cpu = world == 1 and not args.no_cpu_baseline              # only then is the checkpoint needed twice (GPU upload, then the CPU leg)
    tensors = iter_checkpoint(args.model_dir) if args.model_dir else synth.synth_checkpoint(desc, 1234, 0.02)
    if want_cpu:
        tensors = list(tensors)                                       # otherwise streamed: one tensor in host memory at a time (6.4 GB x 8 ranks for config #5)
    model = Model(desc, product_backend(), device=local_rank)       # raises if the HIP library is missing
    if args.no_graph:
        model.set_option("graph", 0)
    for name, bits in tensors:
        model.upload(name, bits, strict=not args.model_dir)          # a real checkpoint may hold keys the path does not use (non-strict load, GPTModel.h:96)
    model.finalize()

    prompt = synth.synth_prompt(desc.vocab, args.prompt, 1234 + rank)[None, :]
    model.forward(prompt)                                           # untimed: allocates the prefill workspace, warms the code objects
    model.synchronize()
    prefill_ms = float("inf")
    for _ in range(3):                                              # best of three (the clocks are still ramping in the first timed pass); config info,
        model.reset_cache()                                         # not part of the timed decode region
        t0 = time.perf_counter()
        model.forward(prompt)
        model.synchronize()
        prefill_ms = min(prefill_ms, (time.perf_counter() - t0) * 1e3)
    model.sample(GREEDY)
    # the library switches attention forms at a context limit and re-captures its graphs when a decode call crosses it: keep the warm-up and
    # the timed region on the same form (the one beyond the limit) when the timed region would cross
    for key, off in (("attn.direct_limit", "attn.direct_max"), ("attn.nw4_limit", "attn.direct_nw4")):
        limit = model.get_option(key)                                  # what tgx_create chose for this geometry and batch
        if limit and args.prompt + 1 + args.warmup <= limit < args.prompt + 1 + args.warmup + args.steps:
            model.set_option(off, 0)

    def sync():
        model.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    model.decode(args.warmup, GREEDY, fetch=False)                  # untimed warm-up (instantiates the graph)
    elapsed, _mine = timed_region(lambda: model.decode(args.steps, GREEDY, fetch=False), sync, group)   # EXACTLY K steps per rank

    if rank != 0:
        if group: group.close()
        return

    T0 = args.prompt + 1 + args.warmup                              # tokens in the cache at the first timed step
    T_mean = T0 + (args.steps - 1) / 2.0
    tok_s = aggregate_tokens_per_s(world, args.steps, elapsed)
    bytes_tok = model.bytes_per_token(int(round(T_mean)))

    prof = model.profile_decode(args.profile_reps)
    n_gu, ms_gu = prof["gateup"]
    gu_bytes = (4 if args.dtype == "fp32" else 2) * ((1 if desc.family == "gpt2" else 2) * desc.inter) * desc.hidden   # GPT-2: c_fc alone
    gu_us = ms_gu / n_gu * 1e3
    achieved = gu_bytes / (gu_us * 1e-6) / 1e9
    classes = {k: round(ms / n * 1e3, 3) for k, (n, ms) in prof.items() if n}
    # HBM traffic of the dominant kernel cannot be read from inside the run (PMC counters need rocprofv3 around the process): it is the
    # figure a separate `rocprofv3 --pmc FETCH_SIZE` pass of this command gave (x2: the gfx950 correction of MI355X_MICROARCH.md §HBM),
    # stored per (model, dtype) in profiles/pmc_traffic.json — null for a configuration that has no such pass.
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic, traffic_source = None, "not measured for this model/dtype (no rocprofv3 --pmc pass on record)"
    if os.path.exists(pmc_path):
        try:
            rec = json.load(open(pmc_path))
            ent = rec.get("by_config", {}).get(f"{args.model}:{args.dtype}") if not args.model_dir else None
            if ent is None and args.model == "llama-3.2-1b" and args.dtype == "bf16" and "gateup_bytes_per_launch" in rec:
                ent = {"gateup_bytes_per_launch": rec["gateup_bytes_per_launch"], "source": rec.get("source", "profiles/r01_bench_pmc.txt")}
            if ent:
                have = kernel_source_sha256()
                if ent.get("kernel_src_sha256") == have:
                    traffic = ent["gateup_bytes_per_launch"]
                    traffic_source = (f"static: {ent.get('source', 'profiles/pmc_traffic.json')} (separate rocprofv3 --pmc FETCH_SIZE pass, x2 gfx950 correction), not collected in this run; "
                                      f"taken with the same kernel source (sha256 {have[:12]} of csrc/kernels/gemv.h + common.h)")
                else:
                    traffic_source = (f"a rocprofv3 --pmc FETCH_SIZE figure is on record ({ent['gateup_bytes_per_launch']} B per launch) but it was taken with another build of the "
                                      f"kernel (source sha256 {str(ent.get('kernel_src_sha256'))[:12]} vs {have[:12]} now): re-run tools/bench_configs.sh")
        except Exception as e:
            traffic_source = f"profiles/pmc_traffic.json unreadable: {e}"
    kname = "gemv_kernel<PRO_LAYERNORM,EPI_GELU> (ln_2 + c_fc + gelu)" if desc.family == "gpt2" else "gemv_kernel<PRO_RMSNORM,EPI_SILU_MUL> (gate_up)"
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": gu_bytes, "avg_launch_us": round(gu_us, 3),
                # where `achieved` comes from: no per-kernel clock exists inside a hipGraph replay, so the class is launched back to back over all
                # layers between two HIP events on the launch stream right after the timed region (tgx_profile_decode); the in-situ figure is rocprofv3's
                "achieved_source": "HIP events around the kernel class launched back-to-back over all layers on the launch stream, right after the timed graph replay "
                                   "(tgx_profile_decode); rocprofv3 --kernel-trace of the headline command with eager launches: profiles/r06_bench_kernel_stats.txt (12.53 us avg for the gate_up kernel against 12.8-13.1 here)",
                "kernel_classes_avg_us": classes,
                "step_bytes_per_token": bytes_tok, "step_achieved": round(bytes_tok * tok_s / world / 1e9, 1),
                "step_frac": round(bytes_tok * tok_s / world / 1e9 / HBM_PEAK_GBS, 4)}

    cpu = None
    if want_cpu:
        try:
            cpu = cpu_baseline(desc, tensors, args.cpu_seconds, args.prompt, 1234)
        except Exception as e:     # the GPU number stands on its own; say why the baseline is absent
            cpu = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}

    # batched prefill (MFMA): algorithmic flops of the S x H·W^T products (one bf16 pass; the kernel issues two, hi and lo)
    L, H, I = desc.layers, desc.hidden, desc.inter
    gemm_flops = 2.0 * args.prompt * L * ((desc.q_dim + 2 * desc.kv_dim) * H + H * desc.q_dim + (2 if desc.family == "gpt2" else 3) * I * H)
    attn_flops = 4.0 * L * desc.heads * desc.head_dim * args.prompt * (args.prompt + 1) / 2.0
    prefill_tflops = (gemm_flops + attn_flops) / (prefill_ms * 1e-3) / 1e12
    # what the matrix cores execute: fp32 activations enter as exact sums of 16-bit terms — 3 for the K / V columns of the bf16 QKV product
    # (their results are rounded into the cache), 2 elsewhere; the attention products (QK^T, PV) take 2 passes each
    kv_flops = 2.0 * args.prompt * L * 2 * desc.kv_dim * H
    executed_flops = (3.0 if args.dtype == "bf16" else 2.0) * kv_flops + 2.0 * (gemm_flops - kv_flops) + 2.0 * attn_flops
    if args.dtype == "fp32":
        executed_flops = gemm_flops + attn_flops          # f32-input MFMA: one pass
    prefill_exec_tflops = executed_flops / (prefill_ms * 1e-3) / 1e12

    line = {
        # BASELINE.json's metric string for the headline configuration; other --model / --dtype runs name themselves
        "metric": ("decode tokens/sec (and % HBM roofline), Llama-3.2-1B bf16 batch=1, 1 GPU" if (args.model == "llama-3.2-1b" and args.dtype == "bf16")
                   else f"decode tokens/sec (and % HBM roofline), {desc.name} {args.dtype} batch=1, 1 GPU"),
        "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "checkpoint" if args.model_dir else "synthetic",
        "config": {"workload": f"{desc.name} {args.dtype}, batch 1 per GPU: {args.prompt}-token prefill then greedy decode "
                               f"(one step = one token, context {T0}..{T0 + args.steps - 1})",
                   "replicas": world, "prompt_tokens": args.prompt, "prefill_ms": round(prefill_ms, 2),
                   "prefill_tflops": round(prefill_tflops, 1), "prefill_frac_of_2500_algorithmic": round(prefill_tflops / 2500.0, 4),
                   "prefill_tflops_executed": round(prefill_exec_tflops, 1), "prefill_frac_of_2500_executed": round(prefill_exec_tflops / 2500.0, 4),
                   "params": desc.param_count(), "graph": not args.no_graph},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    print(json.dumps(line), flush=True)
    if group: group.close()


if __name__ == "__main__":
    main()
