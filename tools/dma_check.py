#!/usr/bin/env python3
"""LDS-DMA prefill GEMM (option prefill.gemm_dma) vs the register-staged one: logits / KV equality and time."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model

name = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = copy.deepcopy(known_desc(name))
if layers:
    d.layers, d.vocab = layers, 8192
d.max_ctx = 2304
m = Model(d).load_synthetic(1234, 0.02).finalize()
for S in (300, 2048):
    ids = synth.synth_prompt(d.vocab, S, 3)[None, :]
    res = {}
    for dma in (0, 535, 543, 0, 535, 543):
        m.set_option("prefill.gemm_dma", dma)
        m.reset_cache(); m.synchronize()
        t0 = time.perf_counter(); m.forward(ids); m.synchronize(); dt = time.perf_counter() - t0
        lg = m.logits(False).copy(); kv = m.read_kv(0, d.layers - 1)
        print(f"S={S} dma={dma}: {dt * 1e3:.2f} ms", flush=True)
        res[dma] = (lg, kv)
    for v in (535, 543):
        a, b = res[0], res[v]
        err = np.abs(a[0] - b[0]).max() / np.abs(a[0]).max()
        print(f"S={S} dma={v}: logits rel diff {err:.2e}; K equal {np.array_equal(a[1][0], b[1][0])}, V equal {np.array_equal(a[1][1], b[1][1])}", flush=True)
