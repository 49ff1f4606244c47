#!/bin/bash
# round 3: tile geometry of the four-block skinny products (B = 48 / 64), split choice
R=$GRAFT_REPO_ROOT
for o in "skinny.cfg_mid=0" "skinny.cfg_mid=1" "skinny.cfg_mid=2" "skinny.cfg_mid=2;skinny.gu_split=1" "skinny.wgs=256" "skinny.wgs=768" "skinny.wgs=1024"; do
  echo "## llama-3.2-1b prompt 512 $o"; python $R/tools/batch_bench.py --prompt 512 --batches 32,64 --steps 64 --opts "$o" 2>&1 | grep "B="
done
