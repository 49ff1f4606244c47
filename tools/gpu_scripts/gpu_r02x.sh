set -x
O=gpurun_out/r02x; mkdir -p $O
python tools/prefill_bench.py --reps 5 > $O/pf.log 2>&1
python tools/prefill_bench.py --model mistral-7b-v0.3 --reps 3 --seq 2048 >> $O/pf.log 2>&1
python tools/prefill_bench.py --model gpt2 --reps 3 --seq 1000 >> $O/pf.log 2>&1
cat $O/pf.log
python -m pytest tests/test_hip_prefill.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
