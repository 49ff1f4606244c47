"""debug: per-row distance of the appended K / V rows from the oracle's after 6 forced steps of a 40-row batch, by step form"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np
from conftest import load_golden, rel_err
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, product_backend
from oracle.oracle_ffi import OracleModel
fam, dtype, rows = sys.argv[1] if len(sys.argv) > 1 else "qwen2_tiny", "bf16", int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg, g = load_golden(fam)
d = desc_from_hf_config(cfg, dtype, max_batch=rows)
seed, std = int(g["seed"]), float(g["std"])
ref = OracleModel(d).load_synthetic(seed, std).finalize()
p = g["prompt"]; V = d.vocab
ids = np.concatenate([(p + 3 * b) % V for b in range(rows)])
ref.forward(ids); toks = [ref.sample(GREEDY).copy()]
for s in range(6): toks.append(ref.decode(1, GREEDY)[0].copy())
lr = ref.logits(rounded=False).copy()
refkv = [[ref.read_kv(r, l) for l in range(d.layers)] for r in range(rows)]
for name, opts in (("default", []), ("step_rows=32", [("decode.step_rows", 32)]), ("raw_fuse=0", [("attn.raw_fuse", 0)]), ("batch_mfma=0", [("attn.batch_mfma", 0)])):
    gpu = Model(d, product_backend()).load_synthetic(seed, std).finalize()
    for k, v in opts: gpu.set_option(k, v)
    gpu.forward(ids); gpu.sample(GREEDY)
    for s in range(6):
        onehot = np.full((rows, V), -1.0, np.float32); onehot[np.arange(rows), toks[s]] = 1.0
        gpu.set_logits(onehot); gpu.sample(GREEDY); gpu.decode(1, GREEDY)
    worst = []
    for r in range(rows):
        for l in range(d.layers):
            for which, (a, b) in enumerate(zip(gpu.read_kv(r, l), refkv[r][l])):
                dif = np.abs(a - b)
                i = np.unravel_index(np.argmax(dif), dif.shape)
                worst.append((float(dif.max()) / max(1e-9, float(np.abs(b[i]))), r, l, "kv"[which], i, float(a[i]), float(b[i])))
    ulp, floor = 2.0 ** -7, 1e-3
    for r in range(rows):
        for l in range(d.layers):
            for which, (a, b) in enumerate(zip(gpu.read_kv(r, l), refkv[r][l])):
                fl = floor if l == 0 else max(floor, 2e-2)
                lim = ulp * (np.maximum(np.abs(a), np.abs(b)) + fl * np.abs(b).max())
                bad = np.abs(a - b) > lim
                for i in zip(*np.nonzero(bad)):
                    print("   BAD", name, "row", r, "layer", l, "kv"[which], tuple(int(x) for x in i), "gpu %.7g ref %.7g diff %.3g limit %.3g max|ref| %.3g" % (a[i], b[i], abs(a[i] - b[i]), lim[i], np.abs(b).max()))
    worst.sort(reverse=True)
    print(name, "logits rel err %.2e" % rel_err(gpu.logits(rounded=False), lr), "worst entries (rel, row, layer, k/v, index, gpu, ref):")
    for w in worst[:4]: print("   ", w)
