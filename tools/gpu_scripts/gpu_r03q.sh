#!/bin/bash
R=$GRAFT_REPO_ROOT
for o in "attn.direct_nw4=0" "attn.direct_nw4=100000"; do
  echo "== $o"
  python $R/tools/quick_bench.py --model qwen2.5-0.5b --prompt 16 --steps 256 --opt $o 2>&1 | grep "decode\|attn"
  python $R/tools/quick_bench.py --model llama-3.2-1b --prompt 16 --steps 256 --opt $o 2>&1 | grep "decode\|attn"
  python $R/tools/quick_bench.py --model llama-3.2-1b --prompt 256 --steps 256 --opt $o 2>&1 | grep "decode\|attn"
  python $R/tools/quick_bench.py --model mistral-7b-v0.3 --prompt 16 --steps 128 --opt $o 2>&1 | grep "decode\|attn"
done
