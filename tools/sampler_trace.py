#!/usr/bin/env python3
"""Eager sampled decode steps for a rocprofv3 --kernel-trace pass (graph replays are opaque to the tracer):
   cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/st -o st -- python $R/tools/sampler_trace.py "T=0.8,top_p=0.9" ; python tools/rocpd_stats.py <db>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import Model, SamplerCfg, product_backend
kw = {}
for item in (sys.argv[1] if len(sys.argv) > 1 else "temperature=0.8,top_p=0.9").split(","):
    k, v = item.split("=")
    kw[k] = int(v) if k == "top_k" else float(v)
cfg = SamplerCfg(**kw)
desc = known_desc("llama-3.2-1b")
m = Model(desc, product_backend())
for name, bits in synth.synth_checkpoint(desc, 1234, 0.02):
    m.upload(name, bits)
m.finalize()
m.set_option("graph", 0)
m.forward(synth.synth_prompt(desc.vocab, 256, 1)[None, :]); m.sample(cfg, seed=1)
m.decode(64, cfg, seed=1, fetch=False); m.synchronize()
print("done", cfg)
