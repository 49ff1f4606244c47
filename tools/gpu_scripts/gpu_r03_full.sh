#!/bin/bash
# round 3: the driver's own sequence — GPU suite, smoke, default bench
mkdir -p gpurun_out/r03full
(time timeout 2700 python -m pytest tests -m gpu -x -q) > gpurun_out/r03full/pytest.log 2>&1; tail -5 gpurun_out/r03full/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03full/smoke.log 2>&1; tail -2 gpurun_out/r03full/smoke.log
python bench.py > gpurun_out/r03full/bench.log 2>&1; tail -1 gpurun_out/r03full/bench.log > gpurun_out/r03full/bench_line.json; cut -c1-600 gpurun_out/r03full/bench_line.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03full/bench_driver.log 2>&1; tail -1 gpurun_out/r03full/bench_driver.log | cut -c1-300
