#!/usr/bin/env python3
"""Long-context decode: the VALU attention kernel vs the MFMA one (option attn.mfma_min), ms/token and logits agreement."""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model
name = sys.argv[1] if len(sys.argv) > 1 else "llama-3.2-1b"
ctxs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2048,4096,8000").split(",")]
d = copy.deepcopy(known_desc(name)); d.max_ctx = max(ctxs) + 256
m = Model(d).load_synthetic(1234, 0.02).finalize()
for T in ctxs:
    ids = synth.synth_prompt(d.vocab, T, 9)[None, :]
    res = {}
    for mf in (1 << 30, 1):
        m.set_option("attn.mfma_min", mf)
        m.reset_cache(); m.forward(ids); m.sample(GREEDY)
        m.decode(8, GREEDY, fetch=False); m.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); m.decode(64, GREEDY, fetch=False); m.synchronize(); best = min(best, (time.perf_counter() - t0) / 64)
        m.reset_cache(); m.forward(ids); t = m.sample(GREEDY).copy(); out = m.decode(6, GREEDY).copy(); lg = m.logits(False).copy()
        prof = m.profile_decode(4)
        res[mf] = (best, t, out, lg, prof["attn"][1] / prof["attn"][0] * 1e3)
    a, b = res[1 << 30], res[1]
    err = np.abs(a[3] - b[3]).max() / np.abs(a[3]).max()
    print(f"{d.name} ctx {T}: VALU {a[0] * 1e3:.4f} ms/tok (attn class {a[4]:.1f} us/layer)  MFMA {b[0] * 1e3:.4f} ms/tok (attn {b[4]:.1f} us/layer)  "
          f"logits rel diff {err:.2e}  ids equal {np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])}", flush=True)
