#!/bin/bash
mkdir -p gpurun_out/r03a
cd tools/probes
timeout 120 ./build/engine_probe 16 7 0 0 1b 3 > ../../gpurun_out/r03a/t0.txt 2>&1; echo "exit $?" >> ../../gpurun_out/r03a/t0.txt
timeout 120 ./build/engine_probe 16 7 0 1 1b 3 > ../../gpurun_out/r03a/t0_stats.txt 2>&1
grep -h "engine (1\|launches\|parity\|give-up\|depth" ../../gpurun_out/r03a/t0.txt
cat ../../gpurun_out/r03a/t0_stats.txt | tail -44
