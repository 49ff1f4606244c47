#!/bin/bash
# round 3, closing: the driver's sequence (suite, smoke, bench) + kernel traces of a B = 8 and a B = 16 step
O=gpurun_out/r03zh; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(time timeout 2700 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-400 $O/bench_line.json
cd /tmp
for B in 8 16; do
rocprofv3 --kernel-trace --stats -d /tmp/b$B -o b -- python $R/tools/batch_bench.py --batches $B --steps 48 > $R/$O/b$B.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b$B -name "*.db" | head -1) > $R/$O/b${B}_kernel_stats.txt 2>&1; head -16 $R/$O/b${B}_kernel_stats.txt | cut -c1-180
done
