#!/usr/bin/env python3
"""Debug: injected-cache decode step vs the oracle, per-layer divergence of the appended cache row."""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend
from oracle.oracle_ffi import OracleModel, build_oracle, oracle_backend
build_oracle(); oracle_backend().set_threads(32)
name, S = sys.argv[1], int(sys.argv[2])
d = copy.deepcopy(known_desc(name)); d.max_ctx = S + 16
if len(sys.argv) > 3: d.layers = int(sys.argv[3])
ref, gpu = OracleModel(d), Model(d, product_backend())
for n, b in synth.synth_checkpoint(d, 1234, 0.02):
    ref.upload(n, b); gpu.upload(n, b)
ref.finalize(); gpu.finalize()
p = synth.synth_prompt(d.vocab, S, 3)[None, :]
ref.forward(p); gpu.forward(p)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
print("prefill logits rel", rel(gpu.logits(False), ref.logits(False)))
for step in range(3):
    tok = ref.sample(GREEDY)
    for l in range(d.layers):
        k, v = ref.read_kv(0, l); gpu.write_kv(0, l, k, v)
    # verify the injection
    k0, v0 = gpu.read_kv(0, 0); kr, vr = ref.read_kv(0, 0)
    assert np.array_equal(k0, kr) and np.array_equal(v0, vr)
    ref.forward(tok[None, :]); gpu.forward(tok[None, :])
    print(f"step {step}: logits rel {rel(gpu.logits(False), ref.logits(False)):.3e}")
    T = ref.past_length
    out = []
    for l in range(d.layers):
        kg, vg = gpu.read_kv(0, l); kr, vr = ref.read_kv(0, l)
        assert np.array_equal(kg[:T - 1], kr[:T - 1])
        dk = np.abs(kg[T - 1] - kr[T - 1]).max() / np.abs(kr[T - 1]).max(); dv = np.abs(vg[T - 1] - vr[T - 1]).max() / np.abs(vr[T - 1]).max()
        out.append(f"{l}:{dk:.1e}/{dv:.1e}")
    print("  new row rel diff k/v per layer:", " ".join(out))
