set -x
O=gpurun_out/r02t; mkdir -p $O
for ns in 17 18 20 24 32; do echo "nsplit=$ns" >> $O/nsplit.log; python tools/sweep.py --prompt 2048 --steps 128 --pre "attn.nsplit=$ns" >> $O/nsplit.log 2>&1; done
cat $O/nsplit.log
