#!/usr/bin/env python3
"""Where a sampled step's extra time goes, from INSIDE the captured graph: wall-clock stamps (100 MHz) written by the sampler's launches.
Experiment build only:  TGX_EXTRA_FLAGS=-DTGX_SAMP_TIMELINE python -c "from tinygpt_amd import build as b; b.build_lib()"
   python tools/sampler_timeline.py ["temperature=0.8,top_p=0.9" ...]
Stamps (workgroup 0, thread 0): 0/1 first vocabulary pass entry / exit (level 0, or the min-p stage), 2/3 second pass (compaction, or stage 1),
4 tail / pick entry, 5 threshold known (tail) or normaliser summed (pick), 6 tile chosen, 7 token written, 8 embedding row gathered."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import Model, SamplerCfg, product_backend
desc = known_desc("llama-3.2-1b")
be = product_backend()
m = Model(desc, be)
for name, bits in synth.synth_checkpoint(desc, 1234, 0.02):
    m.upload(name, bits)
m.finalize()
lib = be.lib if hasattr(be, "lib") else be._lib
fn = lib.tgx_debug_samp_timeline
fn.restype = ctypes.c_int
prompt = synth.synth_prompt(desc.vocab, 256, 1)[None, :]
from tinygpt_amd.ffi import GREEDY
for _ in range(2):
    m.reset_cache(); m.forward(prompt); m.sample(GREEDY); m.decode(16, GREEDY, fetch=False); m.synchronize()
    t0 = time.perf_counter(); m.decode(64, GREEDY, fetch=False); m.synchronize()
    print(f'== greedy: host clock {(time.perf_counter() - t0) / 64 * 1e6:.1f} us/step', flush=True)
N = int(os.environ.get('STEPS', '64'))
for spec in (sys.argv[1:] or ["temperature=0.8,top_p=0.9", "temperature=1.0", "temperature=0.8,top_k=50", "temperature=1.0,min_p=0.05"]):
    kw = {}
    for item in spec.split(","):
        k, v = item.split("=")
        kw[k] = int(v) if k == "top_k" else float(v)
    cfg = SamplerCfg(**kw)
    m.reset_cache(); m.forward(prompt); m.sample(cfg, seed=1)
    m.decode(16, cfg, seed=1, fetch=False); m.synchronize()
    t0 = time.perf_counter(); m.decode(N, cfg, seed=1, fetch=False); m.synchronize(); host = (time.perf_counter() - t0) / N
    out = np.zeros((64, 10), dtype=np.uint64); n = ctypes.c_uint(0)
    assert fn(m._ctx, out.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong)), ctypes.byref(n)) == 0
    order = [(n.value - 1 - i) & 63 for i in range(48)][::-1]          # the last 48 complete steps, oldest first
    t = out[order].astype(np.int64) * 10                                 # ns
    first = np.where(t[:, 0] > 0, t[:, 0], t[:, 2])                      # chains without a first pass start at stamp 2
    period = np.diff(first)
    rel = (t - first[:, None]) / 1e3
    rel[t == 0] = np.nan
    med = np.nanmedian(rel, axis=0)
    print(f"== {spec}: host clock {host * 1e6:.1f} us/step; step period mean {period.mean() / 1e3:.1f}, largest {np.sort(period)[-4:] / 1e3} us, median {np.median(period) / 1e3:.1f} us; stamps relative to the first sampler launch's entry (us, median of 48 steps):")
    print("   " + "  ".join(f"[{k}] {med[k]:6.2f}" for k in range(9) if not np.isnan(med[k])))
    print(f"   sampler span (first entry -> embedding gathered) {med[8]:.2f} us; rest of the step {np.median(period) / 1e3 - med[8]:.1f} us", flush=True)
