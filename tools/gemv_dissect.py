#!/usr/bin/env python3
"""Experiment: what the fixed cost of each GEMV launch class consists of — ms/step of the decode graph with parts of the
kernel of ONE class disabled (results invalid in those modes; timing only).  debug.gemv = mode | 1 << (8 + class)."""
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
prompt = synth.synth_prompt(desc.vocab, 512, 1)[None, :]
def run(val):
    m.reset_cache(); m.set_option("debug.gemv", val)
    m.forward(prompt); m.sample(GREEDY)
    m.decode(16, GREEDY, fetch=False); m.synchronize()
    t0 = time.perf_counter(); m.decode(128, GREEDY, fetch=False); m.synchronize(); dt = (time.perf_counter() - t0) / 128
    m.set_option("debug.gemv", 0)
    return dt
base = run(0)
print(f"baseline {base * 1e3:.4f} ms/step")
CLS = {"qkv": 0, "oproj": 2, "gateup": 3, "down": 4}
MODES = {"exit at once": 4, "no weight stream": 2, "no norm arithmetic": 1, "no epilogue": 8, "no stream + no epilogue": 10, "no stream, norm, epilogue": 11}
for cname, ci in CLS.items():
    for mname, mv in MODES.items():
        d = run(mv | (1 << (8 + ci)))
        print(f"{cname:7s} {mname:28s} saves {(base - d) * 1e6 / desc.layers:6.2f} us/layer", flush=True)
