cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_hip_prefill.py -m gpu -x -q 2>&1 | tail -2
python tools/prefill_bench.py --dtype fp32 --seq 256 --reps 3 2>&1 | tail -1
python tools/prefill_bench.py --dtype fp32 --seq 1000 --model gpt2 --reps 3 2>&1 | tail -1
python tools/prefill_bench.py --dtype fp32 --seq 2048 --reps 3 2>&1 | tail -1
