cd /root/repo
python -m pytest tests/test_hip_prefill.py -m gpu -x -q 2>&1 | tail -2
for s in 2048 3072 4096 8192; do python tools/prefill_bench.py --seq $s --reps 3 2>&1 | tail -1; done
python tools/prefill_bench.py --model qwen2.5-0.5b --seq 8192 --reps 3 2>&1 | tail -1
