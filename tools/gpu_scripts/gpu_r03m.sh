#!/bin/bash
mkdir -p gpurun_out/r03m
timeout 1200 python -m pytest tests/test_hip_prefill.py tests/test_hip_fuzz.py tests/test_hip_parity.py -x -q > gpurun_out/r03m/tests.txt 2>&1; tail -4 gpurun_out/r03m/tests.txt
for S in 40 48 64 96 128 192 256 384 512; do
  a=$(timeout 120 python tools/prefill_bench.py --seq $S --reps 5 --opts "prefill.splitk_dma=1" 2>&1 | tail -1)
  b=$(timeout 120 python tools/prefill_bench.py --seq $S --reps 5 --opts "prefill.splitk_dma=0" 2>&1 | tail -1)
  echo "S=$S dma: $a"; echo "S=$S x2 : $b"
done | tee gpurun_out/r03m/ab.txt
for m in mistral-7b-v0.3 qwen2.5-0.5b; do for S in 48 128; do
  a=$(timeout 120 python tools/prefill_bench.py --model $m --seq $S --reps 4 --opts "prefill.splitk_dma=1" 2>&1 | tail -1)
  b=$(timeout 120 python tools/prefill_bench.py --model $m --seq $S --reps 4 --opts "prefill.splitk_dma=0" 2>&1 | tail -1)
  echo "$m S=$S dma: $a"; echo "$m S=$S x2 : $b"
done; done | tee -a gpurun_out/r03m/ab.txt
