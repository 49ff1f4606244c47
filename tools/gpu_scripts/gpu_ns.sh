cd /root/repo
for ns in 4 8 12 16 32; do echo "nsplit=$ns"; python tools/sweep.py --prompt 2048 --steps 128 --pre "attn.nsplit=$ns" 2>&1 | tail -1; done
