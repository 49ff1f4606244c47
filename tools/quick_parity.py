#!/usr/bin/env python3
"""Fast GPU-vs-oracle check on the three tiny fixtures (seconds): prefill logits + 8 greedy steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, rel_err
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, product_backend
from oracle.oracle_ffi import OracleModel
ok = True
for fam in ["llama_tiny", "qwen2_tiny", "mistral_tiny"]:
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, "bf16")
    gpu = Model(d, product_backend()).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    ref = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    gpu.forward(g["prompt"]); ref.forward(g["prompt"])
    e0 = rel_err(gpu.logits(False), ref.logits(False))
    a, b = gpu.sample(GREEDY), ref.sample(GREEDY)
    da, db = gpu.decode(8, GREEDY), ref.decode(8, GREEDY)
    e1 = rel_err(gpu.logits(False), ref.logits(False))
    good = e0 < 1e-3 and e1 < 1e-3 and (a == b).all() and (da == db).all()
    ok &= bool(good)
    print(f"{fam}: prefill rel {e0:.2e}, after 8 steps {e1:.2e}, ids equal {bool((da == db).all())} -> {'OK' if good else 'FAIL'}")
sys.exit(0 if ok else 1)
