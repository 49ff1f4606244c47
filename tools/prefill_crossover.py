#!/usr/bin/env python3
"""Short prompts: skinny MFMA prefill vs tiled split-K MFMA prefill vs passes through the decode kernels (4 positions per pass), ms per prompt length.

    python tools/prefill_crossover.py [--model llama-3.2-1b] [--lens 4,8,16,24,32,48,64,96]
"""
import argparse, dataclasses, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="llama-3.2-1b")
ap.add_argument("--lens", default="4,8,16,24,32,48,64,96")
args = ap.parse_args()
desc = dataclasses.replace(known_desc(args.model), max_ctx=512)
m = Model(desc, product_backend()).load_synthetic(1234, 0.02).finalize()
for S in [int(x) for x in args.lens.split(",")]:
    ids = synth.synth_prompt(desc.vocab, S, 5)[None, :]
    row = []
    for mfma, skinny in ((1, 1), (1, 0), (0, 0)):
        m.set_option("prefill.mfma", mfma)
        m.set_option("prefill.skinny", skinny)
        m.set_option("prefill.min_rows", 4)
        best = 1e9
        for _ in range(4):
            m.reset_cache(); m.synchronize()
            t0 = time.perf_counter(); m.forward(ids); dt = time.perf_counter() - t0
            best = min(best, dt)
        row.append(best * 1e3)
    print(f"S={S:4d}: skinny MFMA (<= 32 rows) {row[0]:7.2f} ms   tiled split-K GEMM {row[1]:7.2f} ms   decode-kernel passes {row[2]:7.2f} ms", flush=True)
