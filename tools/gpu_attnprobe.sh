cd /root/repo; mkdir -p gpurun_out
for d in 0 1 31; do timeout 60 tools/probes/build/attn_probe_$d; done > gpurun_out/attn_probe.txt 2>&1
cat gpurun_out/attn_probe.txt
python -m pytest tests/test_hip_prefill.py -m gpu -x -q 2>&1 | tail -3
python tools/prefill_bench.py --reps 4 2>&1 | tail -2
