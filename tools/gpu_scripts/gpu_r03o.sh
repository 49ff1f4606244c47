#!/bin/bash
mkdir -p gpurun_out/r03o
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "beyond_four or batched or eight_rows or alone_and_inside" 2>&1 | tail -3
for o in "skinny.terms=1" "skinny.terms=0"; do echo "== $o"; python $R/tools/batch_bench.py --batches 17,24,32 --opts "$o" 2>&1; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --batches 24,32 --steps 48 --opts "$o" 2>&1; done | tee gpurun_out/r03o/terms.txt
