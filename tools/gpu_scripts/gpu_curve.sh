cd /root/repo
for i in 1 2; do python tools/prefill_bench.py --reps 5 2>&1 | tail -1; python tools/prefill_bench.py --reps 5 --gemm-tm 64 2>&1 | tail -1 | sed 's/^/   qkv 64-row tiles: /'; done
