#!/usr/bin/env python3
"""Decode rate vs batch size (rows share each weight pass through the batched GEMV): aggregate tokens/s for B = 1, 2, 4, 8.

    python tools/batch_bench.py [--model llama-3.2-1b] [--prompt 512] [--steps 128]
"""
import argparse, dataclasses, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3.2-1b")
ap.add_argument("--prompt", type=int, default=512)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--batches", default="1,2,4,8")
ap.add_argument("--opts", default="", help="tgx_set_option pairs applied after finalize, e.g. 'gateup.ks=4;oproj.ks=2'")
ap.add_argument("--kv-budget", type=int, default=0, help="paged KV: option kv.budget_tokens (set before finalize); 0 = one max_ctx slab per row")
ap.add_argument("--sampler", default="", help="e.g. 'temperature=0.8,top_p=0.9' (default: greedy)")
args = ap.parse_args()
batches = [int(b) for b in args.batches.split(",")]
desc = dataclasses.replace(known_desc(args.model), max_batch=max(batches), max_ctx=args.prompt + 2 * args.steps + 64)
m = Model(desc, product_backend())
if args.kv_budget:
    m.set_option("kv.budget_tokens", args.kv_budget)
for name, bits in synth.synth_checkpoint(desc, 1234, 0.02):
    m.upload(name, bits)
m.finalize()
for kv in filter(None, args.opts.split(";")):
    k, v = kv.split("="); m.set_option(k, int(v))
cfg = GREEDY
if args.sampler:
    from tinygpt_amd.ffi import SamplerCfg
    kw = {}
    for item in args.sampler.split(","):
        k, v = item.split("="); kw[k] = int(v) if k == "top_k" else float(v)
    cfg = SamplerCfg(**kw)
for B in batches:
    m.reset_cache()
    ids = np.stack([synth.synth_prompt(desc.vocab, args.prompt, 77 + b) for b in range(B)])
    m.forward(ids); m.sample(cfg, seed=1)
    m.decode(16 + args.steps, cfg, seed=1, fetch=False); m.synchronize()      # (untimed: every graph the timed steps replay is captured here)
    m.reset_cache(); m.forward(ids); m.sample(cfg, seed=1)
    m.decode(16, cfg, seed=1, fetch=False); m.synchronize()
    t0 = time.perf_counter(); m.decode(args.steps, cfg, seed=1, fetch=False); m.synchronize(); dt = time.perf_counter() - t0
    print(f"{'paged ' if args.kv_budget else ''}{args.sampler + ' ' if args.sampler else ''}B={B}: {dt / args.steps * 1e3:.3f} ms/step, {B * args.steps / dt:.0f} tokens/s aggregate", flush=True)
