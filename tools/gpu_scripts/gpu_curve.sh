R=/root/repo; cd /tmp; export TMPDIR=/tmp; O=$R/gpurun_out/curve; mkdir -p $O
for s in 64 128 256 512; do
rocprofv3 --kernel-trace --stats -d $O/p$s -o p -- python $R/tools/prefill_bench.py --seq $s --reps 3 > $O/p$s.log 2>&1
echo "S=$s"; python $R/tools/rocpd_stats.py $(find $O/p$s -name "*.db" | head -1) 2>&1 | head -10 | cut -c1-150
rm -rf $O/p$s
done
