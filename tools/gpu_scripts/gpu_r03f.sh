#!/bin/bash
mkdir -p gpurun_out/r03f
timeout 1500 python -m pytest tests/test_hip_parity_bar.py -x -q -s -k full_depth > gpurun_out/r03f/tests2.txt 2>&1; echo "exit $?" >> gpurun_out/r03f/tests2.txt
grep "end to end\|passed\|failed\|Error" gpurun_out/r03f/tests2.txt | head
