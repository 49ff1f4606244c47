set -x
O=gpurun_out/r02h; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12) > $O/pytest.log 2>&1
tail -25 $O/pytest.log
python tools/batch_bench.py --batches 1,2,3,4,8,16,32 > $O/batch_1b.log 2>&1; cat $O/batch_1b.log
python tools/batch_bench.py --model mistral-7b-v0.3 --batches 1,2,3,4,8,16,32 --steps 64 > $O/batch_7b.log 2>&1; cat $O/batch_7b.log
