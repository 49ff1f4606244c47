#!/bin/bash
# round 3: where does a mid-size prompt's prefill go? (kernel trace at S = 48, 128, 256)
mkdir -p gpurun_out/r03j
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for S in 48 128 256; do
  rm -rf /tmp/pp$S; rocprofv3 --kernel-trace --stats -d /tmp/pp$S -o p -- python $R/tools/prefill_bench.py --seq $S --reps 6 > $R/gpurun_out/r03j/run_$S.txt 2>&1
  echo "== S=$S: $(tail -1 $R/gpurun_out/r03j/run_$S.txt)"
  python $R/tools/rocpd_stats.py $(find /tmp/pp$S -name "*.db" | head -1) 2>&1 | head -16 | cut -c1-170 | tee $R/gpurun_out/r03j/stats_$S.txt
done
