cd /root/repo; mkdir -p gpurun_out
for i in 1 2 3; do echo old; tools/probes/build/attn_probe_old; echo new; tools/probes/build/attn_probe_0; done
for i in 1 2; do echo old; tools/probes/build/attn_probe_old128; echo new; tools/probes/build/attn_probe_new128; done
for i in 1 2; do echo old; tools/probes/build/attn_probe_old 8192; echo new; tools/probes/build/attn_probe_0 8192; done
