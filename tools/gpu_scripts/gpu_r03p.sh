#!/bin/bash
mkdir -p gpurun_out/r03p
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for B in 4 8; do
  rm -rf /tmp/bb$B; TGX_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d /tmp/bb$B -o p -- python $R/tools/batch_bench.py --batches $B --steps 48 > $R/gpurun_out/r03p/run_$B.txt 2>&1
  echo "== B=$B"; python $R/tools/rocpd_stats.py $(find /tmp/bb$B -name "*.db" | head -1) 2>&1 | head -18 | cut -c1-170 | tee $R/gpurun_out/r03p/stats_$B.txt
done
