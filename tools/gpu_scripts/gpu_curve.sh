cd /root/repo
python -m pytest tests/test_hip_prefill.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
for s in 64 128 256 512 1024; do python tools/prefill_bench.py --seq $s --reps 4 2>&1 | tail -1; python tools/prefill_bench.py --seq $s --reps 4 --opts "prefill.defer_reduce=0" 2>&1 | tail -1 | sed 's/^/   no defer: /'; done
