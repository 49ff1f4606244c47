#!/bin/bash
# round 3: the QKV finish fused into the batched MFMA attention (attn.raw_fuse) — tests, then ms/step by form for each batch size
O=gpurun_out/r03v; mkdir -p $O
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "matrix_cores or batch") > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for P in 512 2048; do
  for o in "attn.batch_mfma=0" "attn.batch_mfma=1;attn.raw_fuse=0" "attn.batch_mfma=1"; do
    echo "## llama-3.2-1b prompt $P  $o"
    python $R/tools/batch_bench.py --prompt $P --batches 5,8,12,16,24,32 --steps 64 --opts "$o" 2>&1 | grep "B="
  done
done
for m in mistral-7b-v0.3 qwen2.5-0.5b qwen3-1.7b; do
for o in "attn.batch_mfma=0" "attn.batch_mfma=1"; do
  echo "## $m prompt 512 $o"
  python $R/tools/batch_bench.py --model $m --prompt 512 --batches 8,16,32 --steps 48 --opts "$o" 2>&1 | grep "B="
done
done
