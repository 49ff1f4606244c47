cd /root/repo; timeout 60 tools/probes/build/tr_read_probe
