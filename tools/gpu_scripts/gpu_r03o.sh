#!/bin/bash
mkdir -p gpurun_out/r03o
R=$GRAFT_REPO_ROOT
for o in "skinny.ksplit=1" "skinny.ksplit=2"; do echo "== $o"; python $R/tools/batch_bench.py --batches 17,24,32 --opts "$o" 2>&1; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --batches 24,32 --steps 48 --opts "$o" 2>&1; python $R/tools/batch_bench.py --model llama-3.2-3b --batches 24,32 --steps 48 --opts "$o" 2>&1; done | tee gpurun_out/r03o/ks2.txt
