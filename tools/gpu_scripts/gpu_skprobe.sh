cd /root/repo; mkdir -p gpurun_out
for i in 1 2; do for a in 2 0; do timeout 120 tools/probes/build/skinny_asrc_$a | tail -2; done; done
