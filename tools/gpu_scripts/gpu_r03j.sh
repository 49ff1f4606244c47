#!/bin/bash
mkdir -p gpurun_out/r03j
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for S in 48; do
  rm -rf /tmp/pp$S; rocprofv3 --kernel-trace --stats -d /tmp/pp$S -o p -- python $R/tools/prefill_bench.py --seq $S --reps 6 > $R/gpurun_out/r03j/run_$S.txt 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/pp$S -name "*.db" | head -1) 2>&1 | head -14 | cut -c1-170 | tee $R/gpurun_out/r03j/stats2_$S.txt
  python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob('/tmp/pp$S/**/*.db',recursive=True)[0]); cur=db.cursor()
rows=cur.execute("select name,start,end from kernels order by start").fetchall()
# last layer sequence of the last forward: print 14 consecutive kernels from the middle of the last pass
n=len(rows); seg=rows[n-60:n-30]
t0=seg[0][1]
for nm,s,e in seg: print(f"{(s-t0)/1e3:8.2f} {(e-s)/1e3:7.2f}  {nm[:80]}")
PY
done
