#!/bin/bash
mkdir -p gpurun_out/r03b
cd tools/probes
timeout 120 ./build/engine_probe 16 7 0 0 1b 3 1 > ../../gpurun_out/r03b/mlp.txt 2>&1; echo "exit $?" >> ../../gpurun_out/r03b/mlp.txt
timeout 120 ./build/engine_probe 16 7 0 1 1b 3 1 > ../../gpurun_out/r03b/mlp_stats.txt 2>&1
grep -h "engine (1\|launches\|parity\|give-up\|depth\|rel diff" ../../gpurun_out/r03b/mlp.txt
cat ../../gpurun_out/r03b/mlp_stats.txt | tail -34
