#!/bin/bash
# round 3: rocprofv3 evidence for the bench line (kernel trace stats + PMC FETCH / WRITE in separate passes)
O=gpurun_out/r03prof; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/bt -o b -- python $R/bench.py --no-graph --no-cpu-baseline > $R/$O/bench_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/bt -name "*.db" | head -1) > $R/$O/bench_kernel_stats.txt 2>&1; head -14 $R/$O/bench_kernel_stats.txt | cut -c1-200
python $R/tools/rocpd_seq.py $(find /tmp/bt -name "*.db" | head -1) > $R/$O/bench_kernel_seq.txt 2>&1; cat $R/$O/bench_kernel_seq.txt
tail -1 $R/$O/bench_trace.log > $R/$O/bench_trace_line.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/bf -o f -- python $R/bench.py --no-graph --no-cpu-baseline --steps 32 --warmup 4 > $R/$O/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/bw -o w -- python $R/bench.py --no-graph --no-cpu-baseline --steps 32 --warmup 4 > $R/$O/bench_pmc_write.log 2>&1
cd $R
python tools/rocpd_pmc.py $(find /tmp/bf -name "*.db" | head -1) > $O/bench_pmc_fetch.txt 2>&1; head -12 $O/bench_pmc_fetch.txt | cut -c1-200
python tools/rocpd_pmc.py $(find /tmp/bw -name "*.db" | head -1) > $O/bench_pmc_write.txt 2>&1; head -8 $O/bench_pmc_write.txt | cut -c1-200
