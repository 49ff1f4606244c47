set -x
O=gpurun_out/r02d; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "beyond_four or eight_rows or batched_decode or rows_are") > $O/pytest.log 2>&1
tail -25 $O/pytest.log
python tools/batch_bench.py --batches 4,5,8,16,32 > $O/batch_1b.log 2>&1; cat $O/batch_1b.log
python tools/batch_bench.py --batches 8,16 --opts "skinny.gu_split=0" > $O/batch_1b_gemv.log 2>&1; cat $O/batch_1b_gemv.log
python tools/batch_bench.py --model mistral-7b-v0.3 --batches 4,8,16,32 --steps 64 > $O/batch_7b.log 2>&1; cat $O/batch_7b.log
python tools/batch_bench.py --batches 8,16 --opts "skinny.wgs=768" > $O/batch_w768.log 2>&1; cat $O/batch_w768.log
python tools/batch_bench.py --batches 8,16 --opts "skinny.wgs=256" > $O/batch_w256.log 2>&1; cat $O/batch_w256.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
TGX_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $R/$O/prof_b8 -o b8 -- python $R/tools/batch_bench.py --batches 8 --steps 64 > $R/$O/prof_b8.log 2>&1
cd $R; python tools/rocpd_stats.py $O/prof_b8/b8_results.db > $O/prof_b8_stats.txt 2>&1; head -16 $O/prof_b8_stats.txt
