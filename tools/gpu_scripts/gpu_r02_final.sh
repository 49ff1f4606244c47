set -x
O=gpurun_out/r02final; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/bench_trace -o b -- python $R/bench.py --no-graph --no-cpu-baseline > $R/$O/bench_trace.log 2>&1
python $R/tools/rocpd_stats.py $R/$O/bench_trace/b_results.db > $R/$O/bench_kernel_stats.txt 2>&1; head -24 $R/$O/bench_kernel_stats.txt
tail -1 $R/$O/bench_trace.log > $R/$O/bench_trace_line.json
rm -rf $R/$O/bench_trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/bench_pmc_fetch -o f -- python $R/bench.py --no-graph --no-cpu-baseline --steps 32 --warmup 4 > $R/$O/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/bench_pmc_write -o w -- python $R/bench.py --no-graph --no-cpu-baseline --steps 32 --warmup 4 > $R/$O/bench_pmc_write.log 2>&1
cd $R
python tools/rocpd_pmc.py $O/bench_pmc_fetch/f_results.db > $O/bench_pmc_fetch.txt 2>&1; head -20 $O/bench_pmc_fetch.txt
python tools/rocpd_pmc.py $O/bench_pmc_write/w_results.db > $O/bench_pmc_write.txt 2>&1; head -12 $O/bench_pmc_write.txt
rm -rf $O/bench_pmc_fetch $O/bench_pmc_write
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-400 $O/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.log 2>&1; tail -1 $O/bench_driver.log | cut -c1-300
python tools/batch_bench.py --batches 1,2,4,8,16,32 > $O/batch_1b.log 2>&1; cat $O/batch_1b.log
