import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np
from conftest import load_golden, rel_err
from tinygpt_amd import synth
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, product_backend
from oracle.oracle_ffi import OracleModel
for fam, dtype, rows, plen in (("mistral_tiny", "bf16", 24, 61), ("mistral_tiny", "fp16", 24, 61), ("llama_tiny", "bf16", 26, 70)):
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, dtype, max_batch=rows); d.max_ctx = plen + 8
    seed, std = int(g["seed"]), float(g["std"])
    ref = OracleModel(d).load_synthetic(seed, std).finalize()
    V = d.vocab
    ids = np.stack([synth.synth_prompt(V, plen, 40 + b) for b in range(rows)])
    ref.forward(ids); toks = [ref.sample(GREEDY).copy()]; lrs = []
    for s in range(6):
        toks.append(ref.decode(1, GREEDY)[0].copy()); lrs.append(ref.logits(rounded=False).copy())
    for form in (24, 0):
        gpu = Model(d, product_backend()).load_synthetic(seed, std).finalize()
        gpu.set_option("attn.batch_mfma", form)
        gpu.forward(ids); gpu.sample(GREEDY)
        errs = []
        for s in range(6):
            onehot = np.full((rows, V), -1.0, np.float32); onehot[np.arange(rows), toks[s]] = 1.0
            gpu.set_logits(onehot); gpu.sample(GREEDY); gpu.decode(1, GREEDY)
            errs.append(rel_err(gpu.logits(rounded=False), lrs[s]))
        print(fam, dtype, "mfma" if form else "valu", ["%.2e" % e for e in errs])
