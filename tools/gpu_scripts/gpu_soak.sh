cd /root/repo; mkdir -p gpurun_out
timeout 1200 python tools/soak.py llama-3.2-1b 1500 2>&1 | tail -10
timeout 600 python tools/soak.py qwen3-1.7b 100 2>&1 | tail -3
