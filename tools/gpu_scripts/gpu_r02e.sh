set -x
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
TGX_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $R/$O/prof_b8 -o b8 -- python $R/tools/batch_bench.py --batches 8 --steps 64 > $R/$O/prof_b8.log 2>&1
cd $R
ls -R $O/prof_b8 | head -20
python tools/rocpd_stats.py $O/prof_b8 > $O/prof_b8_stats.txt 2>&1 || true
head -40 $O/prof_b8_stats.txt
find $O/prof_b8 -name "*stats*" | head
