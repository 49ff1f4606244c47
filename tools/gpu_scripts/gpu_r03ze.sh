#!/bin/bash
# round 3: the QKV (three terms) and o_proj products of the batched step on the LDS-DMA ring kernel too: tests, then ms/step by option
O=gpurun_out/r03ze; mkdir -p $O
R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_prefill.py tests/test_hip_fuzz.py tests/test_hip_fullsize.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for o in "skinny.dma_qkv=0;skinny.dma_oproj=0" "skinny.dma_qkv=1;skinny.dma_oproj=0" "skinny.dma_qkv=0;skinny.dma_oproj=1" "skinny.dma_qkv=1;skinny.dma_oproj=1" "skinny.ksplit=0"; do
  echo "## llama-3.2-1b prompt 512 $o"; python $R/tools/batch_bench.py --prompt 512 --batches 8,16,17,24,32,48,64 --steps 64 --opts "$o" 2>&1 | grep "B="
done
for m in mistral-7b-v0.3 qwen2.5-0.5b llama-3.2-3b; do for o in "skinny.dma_qkv=0;skinny.dma_oproj=0" "skinny.dma_qkv=1;skinny.dma_oproj=1"; do
  echo "## $m $o"; python $R/tools/batch_bench.py --model $m --batches 17,32,64 --steps 48 --opts "$o" 2>&1 | grep "B="
done; done
for o in "prefill.skinny_hidden_max=2048" "prefill.skinny_hidden_max=8192"; do echo -n "mistral prefill S=48 $o: "; python $R/tools/prefill_bench.py --model mistral-7b-v0.3 --seq 48 --reps 5 --opts "$o" | tail -1;  echo -n "3b prefill S=48 $o: "; python $R/tools/prefill_bench.py --model llama-3.2-3b --seq 48 --reps 5 --opts "$o" | tail -1; done
