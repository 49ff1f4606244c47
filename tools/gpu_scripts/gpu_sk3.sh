cd /root/repo
timeout 120 tools/probes/build/skinny3_probe
timeout 120 tools/probes/build/skinny_asrc_0 | tail -2
