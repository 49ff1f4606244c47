set -x
O=gpurun_out/r02ab; mkdir -p $O
timeout 900 python tools/attn_long.py llama-3.2-1b 4096,6000,12000,16000,30000 > $O/a1b.log 2>&1; cat $O/a1b.log
timeout 900 python tools/attn_long.py mistral-7b-v0.3 12000,16000,30000 > $O/a7b.log 2>&1; cat $O/a7b.log
timeout 900 python tools/attn_long.py qwen2.5-0.5b 6000,10000,16000 > $O/aq.log 2>&1; cat $O/aq.log
timeout 900 python tools/attn_long.py llama-3.2-3b 8000,16000 > $O/a3b.log 2>&1; cat $O/a3b.log
