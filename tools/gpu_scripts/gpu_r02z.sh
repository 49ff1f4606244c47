set -x
R=$PWD; O=gpurun_out/r02z; mkdir -p $O
python -m pytest tests/test_hip_prefill.py -m gpu -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
python tools/prefill_bench.py --reps 6 2>&1 | tail -3
python tools/prefill_bench.py --reps 3 --model mistral-7b-v0.3 2>&1 | tail -2
python tools/prefill_bench.py --reps 3 --seq 8192 2>&1 | tail -2
