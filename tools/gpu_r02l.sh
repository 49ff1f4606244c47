set -x
O=gpurun_out/r02l; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
python tools/prefill_bench.py --reps 4 > $O/prefill.log 2>&1; cat $O/prefill.log
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-400
