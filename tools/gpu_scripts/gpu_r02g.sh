set -x
O=gpurun_out/r02g; mkdir -p $O
(time timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "beyond_four or eight_rows or batched_decode or rows_are") > $O/pytest.log 2>&1
tail -12 $O/pytest.log
for opts in "skinny.cfg_mid=0" "skinny.cfg_mid=1;skinny.wgs=512" "skinny.cfg_mid=2" "skinny.cfg_mid=0;skinny.wgs=512;skinny.gu_split=1"; do
  echo "== $opts" >> $O/batch_cfgs.log
  python tools/batch_bench.py --batches 8,16,32 --opts "$opts" >> $O/batch_cfgs.log 2>&1
done
cat $O/batch_cfgs.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
for cfg in 0 1; do
TGX_NO_GRAPH=1 rocprofv3 --kernel-trace --stats -d $R/$O/prof_c$cfg -o b8 -- python $R/tools/batch_bench.py --batches 8 --steps 64 --opts "skinny.cfg_mid=$cfg;skinny.wgs=$((256 + 768*cfg))" > $R/$O/prof_c$cfg.log 2>&1
python $R/tools/rocpd_stats.py $R/$O/prof_c$cfg/b8_results.db > $R/$O/prof_c${cfg}_stats.txt 2>&1; head -12 $R/$O/prof_c${cfg}_stats.txt
done
