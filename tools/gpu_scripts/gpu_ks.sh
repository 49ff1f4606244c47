cd /root/repo
python -m pytest tests/test_hip_fullsize.py tests/test_hip_prefill.py -m gpu -x -q 2>&1 | tail -3
for s in 8 16; do python tools/prefill_bench.py --seq $s --reps 4 2>&1 | tail -1; python tools/prefill_bench.py --seq $s --reps 4 --opts "skinny.ksplit=0" 2>&1 | tail -1 | sed 's/^/   ksplit off: /'; done
