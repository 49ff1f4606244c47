#!/bin/bash
# round 3: prompts of 33-64 rows on the four-block skinny kernels: tests, then ms per prompt against the tiled split-K path
O=gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_hip_prefill.py -m gpu -x -q -k "33_to_64") > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for S in 33 40 48 56 64; do
  for o in "prefill.skinny_rows=64" "prefill.skinny_rows=32"; do
    echo -n "llama-3.2-1b S=$S $o: "; python $R/tools/prefill_bench.py --seq $S --reps 6 --opts "$o" | tail -1
  done
done
for m in mistral-7b-v0.3 llama-3.2-3b qwen2.5-0.5b; do
  for o in "prefill.skinny_rows=64" "prefill.skinny_rows=32"; do
    echo -n "$m S=48 $o: "; python $R/tools/prefill_bench.py --model $m --seq 48 --reps 5 --opts "$o" | tail -1
  done
done
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/s48 -o b -- python $R/tools/prefill_bench.py --seq 48 --reps 20 > $R/$O/s48.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/s48 -name "*.db" | head -1) > $R/$O/s48_kernel_stats.txt 2>&1; head -16 $R/$O/s48_kernel_stats.txt | cut -c1-200
