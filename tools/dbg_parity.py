import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_golden, rel_err
from tinygpt_amd.desc import desc_from_hf_config
from tinygpt_amd.ffi import GREEDY, Model, product_backend
from oracle.oracle_ffi import OracleModel
for fam in ["llama_tiny", "qwen2_tiny", "mistral_tiny"]:
    cfg, g = load_golden(fam)
    d = desc_from_hf_config(cfg, "bf16")
    gpu = Model(d, product_backend()).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    ref = OracleModel(d).load_synthetic(int(g["seed"]), float(g["std"])).finalize()
    ids = g["ids_bf16"]
    gpu.forward(g["prompt"]); ref.forward(g["prompt"])
    errs = [rel_err(gpu.logits(False), ref.logits(False))]
    for i in range(1, ids.shape[1]):
        gpu.forward(ids[:, i-1:i]); ref.forward(ids[:, i-1:i])
        errs.append(rel_err(gpu.logits(False), ref.logits(False)))
    print(fam, "teacher-forced:", " ".join("%.1e" % e for e in errs))
    kvd = []
    for l in range(d.layers):
        kg, vg = gpu.read_kv(0, l); kr, vr = ref.read_kv(0, l)
        kvd.append((float(np.mean(kg != kr)), float(np.mean(vg != vr)), int(np.argmax((kg != kr).reshape(kg.shape[0], -1).any(1))) ))
    print("   kv mismatch frac / first bad token:", kvd)
    gpu.reset_cache(); ref.reset_cache()
    gpu.forward(g["prompt"]); ref.forward(g["prompt"]); gpu.sample(GREEDY); ref.sample(GREEDY)
    errs = []
    for i in range(15):
        a = gpu.decode(1, GREEDY); b = ref.decode(1, GREEDY)
        errs.append(rel_err(gpu.logits(False), ref.logits(False)))
    print(fam, "free-running  :", " ".join("%.1e" % e for e in errs))
