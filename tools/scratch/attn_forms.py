import copy, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model, product_backend
def rel(a, b): return float(np.abs(a.astype(np.float64) - b).max() / np.abs(b).max())
d = copy.deepcopy(known_desc("llama-3.2-1b")); d.layers, d.vocab, d.max_ctx = 2, 4096, 2048
m = Model(d, product_backend()).load_synthetic(1234, 0.02).finalize()
for n in (1, 5, 31, 128, 513):
    prompt = synth.synth_prompt(d.vocab, n, 100 + n)[None, :]
    res = {}
    for name, dm, sl in (("direct", 1 << 20, 1), ("split+sliced", 0, 1), ("split+combine", 0, 0)):
        m.set_option("attn.direct_max", dm); m.set_option("oproj.sliced", sl)
        m.reset_cache(); m.forward(prompt); m.sample(GREEDY); m.decode(4, GREEDY)
        res[name] = m.logits(rounded=False).copy()
    print(n, "split+sliced vs direct %.2e   split+combine vs direct %.2e   sliced vs combine %.2e" % (rel(res["split+sliced"], res["direct"]), rel(res["split+combine"], res["direct"]), rel(res["split+sliced"], res["split+combine"])), flush=True)
