#!/bin/bash
# round 3: (1) o_proj / down on the 256 x 256 kernel for prompts that fill the chip with such tiles (A/B + result agreement), (2) kernel trace of a B = 32 step
O=gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for spec in "llama-3.2-1b 8192" "mistral-7b-v0.3 4096" "llama-3.2-3b 8192"; do
  set -- $spec
  python $R/tools/prefill_bench.py --model $1 --seq $2 --reps 3 2>&1 | tail -1
  python $R/tools/prefill_bench.py --model $1 --seq $2 --reps 3 --opts "prefill.hidden_256=0" 2>&1 | tail -1
done
python - <<'PY'
import os, sys, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import Model
d = copy.deepcopy(known_desc("llama-3.2-1b")); d.max_ctx = 8192 + 8
m = Model(d).load_synthetic(1234, 0.02).finalize()
ids = synth.synth_prompt(d.vocab, 8192, 3)[None, :]
outs = []
for v in (1, 0, 1):
    m.set_option("prefill.hidden_256", v); m.reset_cache(); m.forward(ids); outs.append(m.logits(False).copy())
print("hidden_256 on/off: rel diff %.2e; on/on equal: %s" % (np.abs(outs[0] - outs[1]).max() / np.abs(outs[1]).max(), np.array_equal(outs[0], outs[2])))
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/b32 -o b -- python $R/tools/batch_bench.py --batches 32 --steps 48 > $R/$O/b32.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b32 -name "*.db" | head -1) > $R/$O/b32_kernel_stats.txt 2>&1; head -30 $R/$O/b32_kernel_stats.txt | cut -c1-180
rocprofv3 --kernel-trace --stats -d /tmp/b16 -o b -- python $R/tools/batch_bench.py --batches 16 --steps 48 > $R/$O/b16.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b16 -name "*.db" | head -1) > $R/$O/b16_kernel_stats.txt 2>&1; head -30 $R/$O/b16_kernel_stats.txt | cut -c1-180
