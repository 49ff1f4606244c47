#!/usr/bin/env python3
"""Prefill time (S = 2048, full model) over LDS-DMA GEMM geometries: option prefill.gemm_dma = flags | tall_sel << 4 | small_sel << 8,
sel = (k per stage == 32 ? 1 : 2) + 4 * (stages - 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import Model
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
d = known_desc(name); d.max_ctx = max(2304, S + 64)
m = Model(d).load_synthetic(1234, 0.02).finalize()
ids = synth.synth_prompt(d.vocab, S, 3)[None, :]
def t(v):
    m.set_option("prefill.gemm_dma", v)
    best = 1e9
    for _ in range(4):
        m.reset_cache(); m.synchronize()
        t0 = time.perf_counter(); m.forward(ids); m.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print(f"register-staged: {t(0):.2f} ms")
names = {1: "k32x2", 5: "k32x3", 9: "k32x4", 2: "k64x2", 6: "k64x3"}
for wide in (4, 0):
    for tall in ((1, 2) if len(sys.argv) > 3 else (1, 5, 9, 2)):
        for small in ((2,) if len(sys.argv) > 3 else (2, 6, 5, 9)):
            v = 3 | wide | (tall << 4) | (small << 8)
            print(f"wide256={'on' if wide else 'off'} tall={names[tall]} small={names[small]}: {t(v):.2f} ms", flush=True)
