#!/bin/bash
# round 3: the LDS-DMA ring skinny kernel in the product: bit-identity test, batch tests, ms/step on / off
O=gpurun_out/r03zd; mkdir -p $O
R=$GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_prefill.py tests/test_hip_fuzz.py -m gpu -x -q -k "dma_ring or batches_beyond or matrix_cores or 33_to_64 or batched_steps") > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for o in "skinny.dma=0" "skinny.dma=1" "skinny.dma=1;skinny.dma_nbw=1" "skinny.dma=1;skinny.dma_rows=1"; do
  echo "## llama-3.2-1b prompt 512 $o"; python $R/tools/batch_bench.py --prompt 512 --batches 8,16,17,24,32,48,64 --steps 64 --opts "$o" 2>&1 | grep "B="
done
for m in mistral-7b-v0.3 qwen2.5-0.5b; do for o in "skinny.dma=0" "skinny.dma=1"; do
  echo "## $m $o"; python $R/tools/batch_bench.py --model $m --batches 17,32,64 --steps 48 --opts "$o" 2>&1 | grep "B="
done; done
for S in 24 48 64; do for o in "skinny.dma=0" "skinny.dma=1"; do echo -n "prefill S=$S $o: "; python $R/tools/prefill_bench.py --seq $S --reps 5 --opts "$o" | tail -1; done; done
