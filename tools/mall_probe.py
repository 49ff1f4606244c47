#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache speed up a weight stream that is already resident?  Per-class kernel time with
every launch on a different layer (HBM) vs all launches on layer 0 (cache-resident after the first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model
d = known_desc("llama-3.2-1b")
m = Model(d).load_synthetic(1234, 0.02).finalize()
m.forward(synth.synth_prompt(d.vocab, 2048, 1)[None, :]); m.sample(GREEDY); m.decode(4, GREEDY)
for same in (0, 1):
    m.set_option("debug.profile_same_layer", same)
    m.profile_decode(2)
    p = m.profile_decode(8)
    print("same_layer" if same else "all_layers", {k: round(ms / n * 1e3, 2) for k, (n, ms) in p.items() if n})
