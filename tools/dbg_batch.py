#!/usr/bin/env python3
"""Debug: batched MFMA decode step vs the 4-row GEMV groups vs the oracle, per row and step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
from conftest import load_golden
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, product_backend
from oracle.oracle_ffi import OracleModel, build_oracle
build_oracle()
fam = sys.argv[1] if len(sys.argv) > 1 else "qwen3_tiny"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 7
cfg, g = load_golden(fam)
d = desc_from_hf_config(cfg, "bf16", max_batch=rows)
gpu = Model(d, product_backend()).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
ref = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
p = g["prompt"]
ids = np.concatenate([(p + 3 * b) % d.vocab for b in range(rows)])
for mode in ("mfma", "gemv"):
    gpu.set_option("decode.mfma_min_batch", 5 if mode == "mfma" else 1 << 20)
    gpu.reset_cache(); ref.reset_cache()
    gpu.forward(ids); ref.forward(ids)
    tg, tr = gpu.sample(GREEDY), ref.sample(GREEDY)
    print(mode, "first tokens equal:", (tg == tr).all())
    for step in range(6):
        a, b = gpu.decode(1, GREEDY), ref.decode(1, GREEDY)
        lg, lr = gpu.logits(False), ref.logits(False)
        err = np.abs(lg - lr).max(axis=1) / np.abs(lr).max()
        print(mode, "step", step, "ids equal", (a == b).all(), "row errs", " ".join("%.1e" % e for e in err), flush=True)
        if not (a == b).all():
            print("   gpu", a.ravel(), "\n   ref", b.ravel())
            break
