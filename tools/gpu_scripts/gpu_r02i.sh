set -x
O=gpurun_out/r02i; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_prefill.py tests/test_hip_fuzz.py tests/test_hip_parity.py -m gpu -x -q) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
python tools/prefill_crossover.py --lens 4,5,6,7,8,12,16,24,32,33,48,64 > $O/cross_1b.log 2>&1; cat $O/cross_1b.log
python tools/prefill_crossover.py --model mistral-7b-v0.3 --lens 4,6,8,16,32,33,64 > $O/cross_7b.log 2>&1; cat $O/cross_7b.log
