#!/bin/bash
# round 3: batched direct-form attention on the matrix cores (attn.batch_mfma) — A/B per batch size and context, then the batch parity tests
O=gpurun_out/r03t; mkdir -p $O
R=$GRAFT_REPO_ROOT
for P in 512 2048; do
  for o in "attn.batch_mfma=0" "attn.batch_mfma=1;attn.batch_mfma_nw=8" "attn.batch_mfma=1;attn.batch_mfma_nw=4"; do
    echo "## llama-3.2-1b prompt $P  $o"
    python $R/tools/batch_bench.py --prompt $P --batches 8,12,16,24,32 --steps 64 --opts "$o" 2>&1 | grep "B="
  done
done
for o in "attn.batch_mfma=0" "attn.batch_mfma=1"; do
  echo "## mistral-7b prompt 512 $o"
  python $R/tools/batch_bench.py --model mistral-7b-v0.3 --prompt 512 --batches 8,16,32 --steps 48 --opts "$o" 2>&1 | grep "B="
  echo "## qwen2.5-0.5b prompt 512 $o"
  python $R/tools/batch_bench.py --model qwen2.5-0.5b --prompt 512 --batches 8,16,32 --steps 48 --opts "$o" 2>&1 | grep "B="
done
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_fuzz.py -m gpu -x -q -k "batch or rows") > $O/pytest.log 2>&1; tail -5 $O/pytest.log
