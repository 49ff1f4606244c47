set -x
O=gpurun_out/r02p; mkdir -p $O
(time timeout 2400 python -m pytest tests/test_hip_prefill.py tests/test_hip_fullsize.py tests/test_hip_fuzz.py -m gpu -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python tools/prefill_bench.py --reps 5 > $O/pf.log 2>&1; cat $O/pf.log
python tools/prefill_bench.py --reps 4 --opts "prefill.gemm_dma=535" >> $O/pf.log 2>&1; tail -4 $O/pf.log
python tools/prefill_bench.py --reps 4 --seq 4096 > $O/pf4k.log 2>&1; cat $O/pf4k.log
python tools/prefill_bench.py --reps 4 --seq 4096 --opts "prefill.gemm_dma=0" >> $O/pf4k.log 2>&1; tail -4 $O/pf4k.log
python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-700
