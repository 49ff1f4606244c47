#!/bin/bash
# round 3: decode batches of 33-64 rows in one weight pass (decode.step_rows): tests, then ms/step against passes of 32 rows
O=gpurun_out/r03y; mkdir -p $O
R=$GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "batches_beyond or matrix_cores or batch") > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for m in llama-3.2-1b mistral-7b-v0.3 qwen2.5-0.5b; do
  for o in "decode.step_rows=64" "decode.step_rows=32"; do
    echo "## $m prompt 512 $o"
    python $R/tools/batch_bench.py --model $m --prompt 512 --batches 32,40,48,64 --steps 48 --opts "$o" 2>&1 | grep "B="
  done
done
