#!/bin/bash
# round 3: engine integrated in the library: tests + A/B at the bench operating point
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_hip_engine.py -x -q > gpurun_out/r03c/tests.txt 2>&1; echo "exit $?" >> gpurun_out/r03c/tests.txt
tail -15 gpurun_out/r03c/tests.txt
for m in 0 1 2; do
  timeout 300 python tools/quick_bench.py --prompt 2048 --steps 256 --ctx 2400 --opt engine.mode=$m > gpurun_out/r03c/bench_m$m.txt 2>&1
  echo "== engine.mode=$m"; grep "decode\|sum of" gpurun_out/r03c/bench_m$m.txt
done
