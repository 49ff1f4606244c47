#!/bin/bash
mkdir -p gpurun_out/r03r
R=$GRAFT_REPO_ROOT
(timeout 2700 python -m pytest tests -m gpu -x -q) > gpurun_out/r03r/pytest.log 2>&1; tail -4 gpurun_out/r03r/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for m in qwen2.5-0.5b llama-3.2-1b; do
  python $R/tools/quick_bench.py --model $m --prompt 16 --steps 200 2>&1 | grep "decode"
  python $R/tools/quick_bench.py --model $m --prompt 16 --steps 200 --opt attn.direct_nw4=0 2>&1 | grep "decode"
done
python $R/tools/quick_bench.py --prompt 2048 --steps 256 --ctx 2400 2>&1 | grep decode
# a generation that crosses both limits, twice (the second one re-uses the captured graphs): wall time of 1000 steps from a 16-token prompt
python - <<'PY'
import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from tinygpt_amd import known_desc, synth
from tinygpt_amd.ffi import GREEDY, Model
d = known_desc("llama-3.2-1b"); d.max_ctx = 2048
m = Model(d).load_synthetic(1234, 0.02).finalize()
ids = synth.synth_prompt(d.vocab, 16, 1)[None, :]
for rep in range(3):
    m.reset_cache(); m.forward(ids); m.sample(GREEDY); m.synchronize()
    t0 = time.perf_counter(); out = m.decode(1000, GREEDY); m.synchronize(); dt = time.perf_counter() - t0
    print(f"generation {rep}: 1000 steps from context 17 (crosses 256 and 768): {dt * 1e3:.1f} ms, checksum {int(out.sum())}")
PY
