cd /root/repo; mkdir -p gpurun_out
for d in 0 2 8 14; do timeout 120 tools/probes/build/skinny_probe_$d | tail -2; done > gpurun_out/skinny_probe2.txt 2>&1
cat gpurun_out/skinny_probe2.txt
