#!/usr/bin/env python3
"""Debug: where do the KV rows of the 5-row llama_tiny batch differ from the oracle's (position, head, dim, values)?"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import conftest  # noqa
from test_hip_parity import make_pair, GREEDY
from oracle.oracle_ffi import build_oracle, oracle_backend
build_oracle()
olib = oracle_backend()
from tinygpt_amd.ffi import product_backend
hip = product_backend()
rows = 5
gpu, ref, g = make_pair("llama_tiny", hip, olib, max_batch=rows, dtype="bf16")
for kv in filter(None, os.environ.get("OPTS", "").split(";")):
    k, v = kv.split("="); gpu.set_option(k, int(v))
p = g["prompt"]; V = gpu.desc.vocab
ids = np.concatenate([(p + 3 * b) % V for b in range(rows)])
print("prompt shape", ids.shape)
gpu.forward(ids); ref.forward(ids)
def cmp(tag):
    for row in range(5):
        for layer in (0, 1):
            for name, g_, r_ in zip("KV", gpu.read_kv(row, layer), ref.read_kv(row, layer)):
                d = np.abs(g_ - r_)
                bad = d > 2.0 ** -7 * (np.maximum(np.abs(g_), np.abs(r_)) + 1e-3 * np.abs(r_).max())
                bad &= d > 0.004
                if bad.any():
                    idx = np.argwhere(bad)
                    for i in idx[:4]:
                        print(tag, "row", row, "layer", layer, name, "idx", tuple(i), "gpu", g_[tuple(i)], "ref", r_[tuple(i)])
cmp("after prefill")
tok = ref.sample(GREEDY); gpu.sample(GREEDY)
for step in range(6):
    onehot = np.full((rows, V), -1.0, np.float32); onehot[np.arange(rows), tok] = 1.0
    gpu.set_logits(onehot); gpu.sample(GREEDY)
    gpu.decode(1, GREEDY); tr = ref.decode(1, GREEDY)[0]
    tok = tr
    cmp(f"after step {step}")
