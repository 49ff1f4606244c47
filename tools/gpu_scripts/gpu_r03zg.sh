#!/bin/bash
# round 3: eight activation blocks (65-128 rows) on the LDS-DMA ring kernel: decode batches of 96 / 128 rows in one pass, prompts of 65-128 rows
R=$GRAFT_REPO_ROOT
for o in "decode.step_rows=64" "decode.step_rows=128"; do
  echo "## llama-3.2-1b $o"; python $R/tools/batch_bench.py --batches 64,80,96,128 --steps 48 --opts "$o" 2>&1 | grep "B="
  echo "## mistral-7b $o"; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --batches 96,128 --steps 32 --opts "$o" 2>&1 | grep "B="
done
for S in 65 80 96 128; do for o in "prefill.skinny_rows=64" "prefill.skinny_rows=128"; do echo -n "S=$S $o: "; python $R/tools/prefill_bench.py --seq $S --reps 5 --opts "$o" | tail -1; done; done
for o in "prefill.skinny_rows=64" "prefill.skinny_rows=128"; do echo -n "mistral S=96 $o: "; python $R/tools/prefill_bench.py --model mistral-7b-v0.3 --seq 96 --reps 4 --opts "$o" | tail -1; done
python - <<'PY'
# agreement: a 100-row prompt and 3 steps of a 100-row batch, eight-block path vs the 64-row forms
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, copy
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model
d = copy.deepcopy(known_desc("llama-3.2-1b")); d.layers = 3; d.vocab = 4096; d.max_ctx = 256; d.max_batch = 100
m = Model(d).load_synthetic(1234, 0.02).finalize()
one = synth.synth_prompt(d.vocab, 100, 5)[None, :]
res = {}
for rows in (128, 64):
    m.set_option("prefill.skinny_rows", rows); m.reset_cache(); m.forward(one); res[rows] = m.logits(False).copy()
print("100-row prompt, eight-block skinny vs tiled: rel diff %.2e" % (np.abs(res[128] - res[64]).max() / np.abs(res[64]).max()))
ids = np.stack([synth.synth_prompt(d.vocab, 20, 9 + b) for b in range(100)])
out = {}
for rows in (128, 64):
    m.set_option("decode.step_rows", rows); m.reset_cache(); m.forward(ids); m.sample(GREEDY); t = m.decode(4, GREEDY).copy(); out[rows] = (t, m.logits(False).copy())
print("100-row batch, one pass vs 64 + 36: ids equal %s, logits rel diff %.2e" % (np.array_equal(out[128][0], out[64][0]), np.abs(out[128][1] - out[64][1]).max() / np.abs(out[64][1]).max()))
PY
