#!/bin/bash
# launch-geometry sweep at the bench's operating point
mkdir -p gpurun_out/sweep
R=$GRAFT_REPO_ROOT
for g in "qkv.bpc=4,6,8" "qkv.ks=1,2,4" "oproj.ks=1,2,4" "oproj.bpc=2,4,8" "gateup.bpc=2,4,6,8" "gateup.ks=1,2" "down.bpc=2,4,8" "down.ks=2,4" "attn.gmax=1,2,4"; do
  echo "== $g"; python $R/tools/sweep.py --prompt 2048 --steps 128 --grid "$g" 2>&1 | tail -5
done | tee gpurun_out/sweep/r03.txt
