cd /root/repo
for i in 1 2; do for s in 1024 2048 4096 8192; do for la in 2 1; do tools/probes/build/attn_probe_la${la}_128 $s; done; done; done
