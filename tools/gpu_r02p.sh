set -x
O=gpurun_out/r02p; mkdir -p $O
timeout 600 python tools/dma_sweep.py llama-3.2-1b 2048 quick > $O/sweep2.log 2>&1; cat $O/sweep2.log
