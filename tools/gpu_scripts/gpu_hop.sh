#!/bin/bash
# hand-off cost probe (tools/probes/hop_probe.hip)
cd /root/repo
mkdir -p gpurun_out
for g in 256 128; do timeout 120 tools/probes/build/hop_probe $g; done > gpurun_out/hop_probe.txt 2>&1
cat gpurun_out/hop_probe.txt
