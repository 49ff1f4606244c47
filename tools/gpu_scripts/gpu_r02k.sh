set -x
O=gpurun_out/r02k; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for cfg in "llama-3.2-1b fp32 256" "llama-3.2-1b fp32 2048" "gpt2 fp32 1000"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -d $R/$O/prof_$1_$3 -o p -- python $R/tools/prefill_bench.py --model $1 --dtype $2 --seq $3 --reps 2 > $R/$O/prof_$1_$3.log 2>&1
  python $R/tools/rocpd_stats.py $R/$O/prof_$1_$3/p_results.db > $R/$O/prof_$1_$3_stats.txt 2>&1; head -14 $R/$O/prof_$1_$3_stats.txt
done
