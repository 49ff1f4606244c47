set -x
O=gpurun_out/r02f; mkdir -p $O
python tools/dbg_batch.py qwen3_tiny 7 > $O/dbg_q3.log 2>&1; cat $O/dbg_q3.log
python tools/dbg_batch.py llama_tiny 7 > $O/dbg_l.log 2>&1; cat $O/dbg_l.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
TGX_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $R/$O/prof_b8 -o b8 -- python $R/tools/batch_bench.py --batches 8 --steps 64 > $R/$O/prof_b8.log 2>&1
cd $R; python tools/rocpd_stats.py $O/prof_b8/b8_results.db > $O/prof_b8_stats.txt 2>&1; head -24 $O/prof_b8_stats.txt
