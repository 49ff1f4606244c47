set -x
O=gpurun_out/r02r; mkdir -p $O
timeout 600 python tools/attn_long.py llama-3.2-1b 2048,8000 > $O/a1b.log 2>&1; cat $O/a1b.log
timeout 600 python tools/attn_long.py mistral-7b-v0.3 8192,24000 > $O/a7b.log 2>&1; cat $O/a7b.log
timeout 600 python tools/attn_long.py qwen2.5-0.5b 30000 > $O/aq.log 2>&1; cat $O/aq.log
