set -x
O=gpurun_out/r02n; mkdir -p $O
timeout 300 python tools/dma_check.py llama-3.2-1b 0 > $O/dma_full.log 2>&1; cat $O/dma_full.log
bash tools/gpu_r02o.sh
