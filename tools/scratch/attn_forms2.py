import copy, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend
def rel(a, b): return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
d = copy.deepcopy(known_desc("llama-3.2-1b")); d.layers, d.vocab, d.max_ctx = 2, 4096, 2048
m = Model(d, product_backend()).load_synthetic(1234, 0.02).finalize()
m.set_option("attn.direct_max", 0)
for n in (1, 2, 3, 4, 5, 8):
    prompt = synth.synth_prompt(d.vocab, n, 100 + n)[None, :]
    out = []
    for k in (1, 2, 3, 4, 8):
        res = {}
        for sl in (0, 1):
            m.set_option("oproj.sliced", sl)
            m.reset_cache(); m.forward(prompt); m.sample(GREEDY); m.decode(k, GREEDY)
            res[sl] = m.logits(rounded=False).copy()
        out.append("%d steps %.1e" % (k, rel(res[1], res[0])))
    print("prompt", n, " ".join(out), "  max|logit| %.3f" % np.abs(res[0]).max(), flush=True)
