cd /root/repo
python -m pytest tests/test_hip_prefill.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -2
python tools/prefill_bench.py --model mistral-7b-v0.3 --reps 3 2>&1 | tail -1
python tools/prefill_bench.py --model llama-3.2-3b --reps 3 2>&1 | tail -1
python tools/prefill_bench.py --model qwen3-1.7b --reps 3 2>&1 | tail -1
python tools/prefill_bench.py --model mistral-7b-v0.3 --seq 8192 --reps 2 2>&1 | tail -1
