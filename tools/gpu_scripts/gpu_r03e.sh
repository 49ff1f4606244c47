#!/bin/bash
mkdir -p gpurun_out/r03e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base pf pf_oproj; do
  opts=""; [ $v = pf ] && opts="--opt pf.mode=1 --opt pf.wgs=128"; [ $v = pf_oproj ] && opts="--opt pf.mode=1 --opt pf.wgs=128 --opt pf.gu_kb=0 --opt pf.qkv_kb=0"
  TGX_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o out -- python $R/tools/quick_bench.py --prompt 2048 --steps 64 --ctx 2400 $opts > $R/gpurun_out/r03e/run_$v.txt 2>&1
  f=$(find /tmp/prof_$v -name "*.db" | head -1)
  echo "== $v"; python $R/tools/rocpd_seq.py "$f" 2>&1 | tee $R/gpurun_out/r03e/seq_$v.txt
done
