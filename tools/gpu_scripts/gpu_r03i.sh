#!/bin/bash
mkdir -p gpurun_out/r03i
timeout 900 python -m pytest tests/test_hip_prefill.py tests/test_hip_fullsize.py -x -q -k "prefill or operating or full_size" > gpurun_out/r03i/tests.txt 2>&1; tail -4 gpurun_out/r03i/tests.txt
for m in llama-3.2-1b llama-3.2-3b mistral-7b-v0.3; do
 for v in 1 0; do
  echo "== $m prefill.qkv_balanced=$v"; timeout 300 python tools/prefill_bench.py --model $m --reps 4 --opts "prefill.qkv_balanced=$v" 2>&1 | tail -2
 done
done
