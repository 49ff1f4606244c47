#!/bin/bash
# round 3: batch tests after the shared RoPE helpers; kernel trace of a 48-row prompt
O=gpurun_out/r03w; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(timeout 800 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "matrix_cores or batch") > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python $R/tools/batch_bench.py --batches 24,32 --steps 64 | grep B=
python $R/tools/batch_bench.py --model qwen3-1.7b --batches 32 --steps 64 | grep B=
python $R/tools/prefill_bench.py --seq 48 --reps 5 | tail -2
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/s48 -o b -- python $R/tools/prefill_bench.py --seq 48 --reps 20 > $R/$O/s48.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/s48 -name "*.db" | head -1) > $R/$O/s48_kernel_stats.txt 2>&1; head -24 $R/$O/s48_kernel_stats.txt | cut -c1-200
