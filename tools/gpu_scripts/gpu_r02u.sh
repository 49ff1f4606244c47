set -x
O=gpurun_out/r02u; mkdir -p $O
python tools/prefill_bench.py --reps 5 > $O/pf.log 2>&1
python tools/prefill_bench.py --model mistral-7b-v0.3 --reps 3 --seq 2048 >> $O/pf.log 2>&1
python tools/prefill_bench.py --model gpt2 --reps 3 --seq 1000 >> $O/pf.log 2>&1
python tools/prefill_bench.py --dtype fp32 --seq 256 --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --seq 8 --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --seq 32 --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --seq 64 --reps 3 >> $O/pf.log 2>&1
cat $O/pf.log
python tools/batch_bench.py --batches 1,4,8,16,32 > $O/batch.log 2>&1; cat $O/batch.log
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
