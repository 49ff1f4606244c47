#!/bin/bash
# the whole -m gpu suite
mkdir -p gpurun_out/tests
timeout 2400 python -m pytest tests -m gpu -x -q -s > gpurun_out/tests/gpu_suite.txt 2>&1; echo "exit $?" >> gpurun_out/tests/gpu_suite.txt
grep -v "^$" gpurun_out/tests/gpu_suite.txt | tail -30
