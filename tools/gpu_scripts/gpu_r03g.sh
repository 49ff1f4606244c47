#!/bin/bash
mkdir -p gpurun_out/r03g
bash tools/bench_configs.sh $GRAFT_REPO_ROOT/gpurun_out/r03g/bench_lines.jsonl 2>&1 | tee gpurun_out/r03g/log.txt
cp profiles/pmc_traffic.json gpurun_out/r03g/pmc_traffic.json
# config #5's flow on a one-GPU box: two replicas of Llama-3.2-3B through the launcher, sharing the GPU
TGX_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --model llama-3.2-3b --prompt 16 --steps 128 --warmup 16 > gpurun_out/r03g/n2_3b.txt 2>&1
tail -1 gpurun_out/r03g/n2_3b.txt | cut -c1-300
