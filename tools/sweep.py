#!/usr/bin/env python3
"""Launch-geometry sweep on one loaded model: ms/token of graph-replayed greedy decode per option set."""
import argparse
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinygpt_amd import known_desc, synth  # noqa: E402
from tinygpt_amd.ffi import GREEDY, Model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3.2-1b")
ap.add_argument("--prompt", type=int, default=256)
ap.add_argument("--steps", type=int, default=96)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--grid", default="", help="e.g. 'down.ks=1,2,4;gateup.bpc=4,8' (cartesian product)")
ap.add_argument("--pre", default="", help="options set before finalize, e.g. 'attn.nsplit=16;lmhead.bpc=8'")
args = ap.parse_args()

d = known_desc(args.model, args.dtype)
m = Model(d)
for kv in filter(None, args.pre.split(";")):
    k, v = kv.split("=")
    m.set_option(k, int(v))
m.load_synthetic(1234, 0.02).finalize()
ids = synth.synth_prompt(d.vocab, args.prompt, 1234)[None, :]
m.forward(ids); m.sample(GREEDY)
axes = []
for ax in filter(None, args.grid.split(";")):
    k, vs = ax.split("=")
    axes.append([(k, int(v)) for v in vs.split(",")])
for combo in itertools.product(*axes) if axes else [()]:
    for k, v in combo:
        m.set_option(k, v)
    m.reset_cache(); m.forward(ids); m.sample(GREEDY)       # same context length for every option set
    m.decode(8, GREEDY, fetch=False); m.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); m.decode(args.steps, GREEDY, fetch=False); m.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.steps)
    print(" ".join(f"{k}={v}" for k, v in combo) or "default", f"-> {best * 1e3:.4f} ms/tok  ({1 / best:.0f} tok/s) T~{m.past_length}", flush=True)
