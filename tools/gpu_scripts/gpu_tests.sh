cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/tests_gpu.log 2>&1; tail -5 gpurun_out/tests_gpu.log
