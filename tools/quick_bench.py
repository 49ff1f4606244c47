#!/usr/bin/env python3
"""Development harness: decode tok/s + per-kernel-class HIP-event times for one synthetic model."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinygpt_amd import known_desc, synth  # noqa: E402
from tinygpt_amd.ffi import GREEDY, Model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3.2-1b")
ap.add_argument("--prompt", type=int, default=64)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--ctx", type=int, default=0)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--opt", action="append", default=[], help="key=value for tgx_set_option (after finalize), e.g. oproj.sliced=0")
args = ap.parse_args()

d = known_desc(args.model, args.dtype)
if args.ctx:
    d.max_ctx = args.ctx
t0 = time.time()
m = Model(d).load_synthetic(1234, 0.02).finalize()
for kv in args.opt:
    k, v = kv.split("=")
    m.set_option(k, int(v))
print(f"load {time.time() - t0:.1f}s  params {d.param_count() / 1e9:.3f}B", flush=True)
ids = synth.synth_prompt(d.vocab, args.prompt, 1234)[None, :]
t0 = time.time(); m.forward(ids); m.synchronize(); tp = time.time() - t0
print(f"prefill {args.prompt} tok: {tp * 1e3:.1f} ms")
m.sample(GREEDY)
m.decode(8, GREEDY)          # warmup (graph instantiate)
m.synchronize()
t0 = time.time(); out = m.decode(args.steps, GREEDY, fetch=False); m.synchronize(); dt = time.time() - t0
T = m.past_length - args.steps / 2
bpt = m.bytes_per_token(int(T))
print(f"decode {args.steps} tok: {dt * 1e3 / args.steps:.3f} ms/tok  {args.steps / dt:.1f} tok/s  "
      f"{bpt * args.steps / dt / 1e9:.0f} GB/s algorithmic ({bpt * args.steps / dt / 8e12 * 100:.1f}% of 8 TB/s)")
prof = m.profile_decode(8)
tot = 0
for k, (n, ms) in prof.items():
    if n:
        print(f"  {k:8s} launches {n:5d}  avg {ms / n * 1e3:8.2f} us   total/token {ms / 8 * 1e3:8.1f} us")
        tot += ms / 8
print(f"  sum of kernel classes per token: {tot * 1e3:.1f} us")
