#!/usr/bin/env python3
"""Batched prefill timing (Llama-3.2-1B geometry, S tokens) — for rocprofv3 MFMA-utilisation passes."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import Model
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3.2-1b"); ap.add_argument("--seq", type=int, default=2048); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--gemm-tm", type=int, default=0); ap.add_argument("--opts", default=""); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--kv-budget", type=int, default=0, help="paged KV: option kv.budget_tokens")
a = ap.parse_args()
d = known_desc(a.model, a.dtype)
m = Model(d)
if a.kv_budget: m.set_option("kv.budget_tokens", a.kv_budget)
m.load_synthetic(1234, 0.02).finalize()
if a.gemm_tm: m.set_option("prefill.gemm_tm", a.gemm_tm)
for kv in filter(None, a.opts.split(";")):
    k, v = kv.split("="); m.set_option(k, int(v))
ids = synth.synth_prompt(d.vocab, a.seq, 1234)[None, :]
L, H, I = d.layers, d.hidden, d.inter
flops = 2.0 * a.seq * L * ((d.q_dim + 2 * d.kv_dim) * H + H * d.q_dim + (2 if d.family == 'gpt2' else 3) * I * H) + 4.0 * L * d.heads * d.head_dim * a.seq * (a.seq + 1) / 2
for r in range(a.reps):
    m.reset_cache(); m.synchronize()
    t0 = time.perf_counter(); m.forward(ids); m.synchronize(); dt = time.perf_counter() - t0
    print(f"{'paged ' if a.kv_budget else ''}{d.name} {a.dtype} prefill S={a.seq}: {dt * 1e3:.2f} ms  {flops / dt / 1e12:.1f} TFLOP/s algorithmic", flush=True)
