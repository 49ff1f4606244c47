#!/bin/bash
mkdir -p gpurun_out/r03n
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_hip_fullsize.py tests/test_hip_host_engine.py -x -q 2>&1 | tail -4
echo "== 1B 512"; python $R/tools/batch_bench.py --batches 1,2,4,8,16,32 2>&1 | tee $R/gpurun_out/r03n/final_512.txt
echo "== 1B 2048"; python $R/tools/batch_bench.py --prompt 2048 --batches 4,8,16,32 --steps 64 2>&1 | tee $R/gpurun_out/r03n/final_2k.txt
echo "== mistral 512"; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --batches 4,8,16,32 --steps 48 2>&1 | tee $R/gpurun_out/r03n/final_mistral.txt
echo "== 3B 512"; python $R/tools/batch_bench.py --model llama-3.2-3b --batches 4,8,16,32 --steps 48 2>&1 | tee $R/gpurun_out/r03n/final_3b.txt
