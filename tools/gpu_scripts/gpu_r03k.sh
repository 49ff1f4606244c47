#!/bin/bash
mkdir -p gpurun_out/r03k
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "fused_into_lm_head or folded or teacher or prefill_logits" > gpurun_out/r03k/tests.txt 2>&1; tail -3 gpurun_out/r03k/tests.txt
for i in 1 2 3; do timeout 300 python tools/quick_bench.py --prompt 2048 --steps 256 --ctx 2400 2>&1 | grep "decode"; done
timeout 300 python tools/quick_bench.py --model qwen2.5-0.5b --prompt 16 --steps 256 2>&1 | grep "decode"
