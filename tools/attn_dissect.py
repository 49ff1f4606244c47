#!/usr/bin/env python3
"""Experiment: where do the ~12 us of the attention launch pair go?  ms/step of the decode graph with parts of the
attention kernels disabled (results are invalid in those modes; timing only)."""
import os, sys, time
# needs the experiment build:  TGX_DISSECT=1 python tinygpt_amd/build.py -f   (rebuild without it afterwards)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend
desc = known_desc("llama-3.2-1b")
m = Model(desc, product_backend())
for name, bits in synth.synth_checkpoint(desc, 1234, 0.02):
    m.upload(name, bits)
m.finalize()
prompt = synth.synth_prompt(desc.vocab, int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 1)[None, :]
def run(label, **opts):
    m.reset_cache()
    for k, v in opts.items():
        m.set_option(k, v)
    m.forward(prompt); m.sample(GREEDY)
    m.decode(16, GREEDY, fetch=False); m.synchronize()
    t0 = time.perf_counter(); m.decode(128, GREEDY, fetch=False); m.synchronize(); dt = (time.perf_counter() - t0) / 128
    print(f"{label:50s} {dt * 1e3:.4f} ms/step  ({dt * 1e6 / desc.layers:.2f} us/layer)", flush=True)
    for k in opts:
        m.set_option(k, 0)
    return dt
base = run("baseline")
for label, o in [("decode kernel exits at once (dbg 4)", {"debug.attn": 4}), ("no K/V work (dbg 1)", {"debug.attn": 1}),
                 ("no LDS merge (dbg 2)", {"debug.attn": 2}), ("no K/V, no merge (dbg 3)", {"debug.attn": 3}),
                 ("decode kernel not launched (skip 1)", {"debug.skip": 1}), ("combine not launched (skip 2)", {"debug.skip": 2}),
                 ("neither launched (skip 3)", {"debug.skip": 3}), ("exit at once + no combine", {"debug.attn": 4, "debug.skip": 2})]:
    d = run(label, **o)
    print(f"{'':50s} saves {(base - d) * 1e6 / desc.layers:.2f} us/layer")
