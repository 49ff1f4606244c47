set -x
R=$PWD; O=gpurun_out/r02v_$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in llama-3.2-1b mistral-7b-v0.3; do
rocprofv3 --kernel-trace --stats -d $R/$O/prof_$m -o p -- python $R/tools/prefill_bench.py --model $m --seq 2048 --reps 3 > $R/$O/prof_$m.log 2>&1
python $R/tools/rocpd_stats.py $(find $R/$O/prof_$m -name "*.db" | head -1) 2>&1 | head -12 | cut -c1-170 > $R/$O/sum_$m.txt
cat $R/$O/sum_$m.txt
rm -rf $R/$O/prof_$m
done
