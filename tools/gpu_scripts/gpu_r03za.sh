#!/bin/bash
# round 3: batched MFMA attention without the look-ahead register set when its workgroups outnumber the CUs; 16-byte loads in the argmax partials
O=gpurun_out/r03za; mkdir -p $O
R=$GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_sampler.py -m gpu -x -q -k "batch or matrix_cores or rows or sampl") > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for o in "attn.batch_la=-1" "attn.batch_la=1" "attn.batch_la=0"; do
  echo "## llama-3.2-1b prompt 512 $o"; python $R/tools/batch_bench.py --prompt 512 --batches 24,32,48,64 --steps 64 --opts "$o" 2>&1 | grep "B="
  echo "## llama-3.2-1b prompt 2048 $o"; python $R/tools/batch_bench.py --prompt 2048 --batches 32,64 --steps 64 --opts "$o" 2>&1 | grep "B="
done
echo "## qwen2.5-0.5b"; python $R/tools/batch_bench.py --model qwen2.5-0.5b --batches 32,64 --steps 64 | grep B=
