set -x
mkdir -p gpurun_out/r02a
cd $GRAFT_REPO_ROOT
nproc > gpurun_out/r02a/host.txt; lscpu | head -20 >> gpurun_out/r02a/host.txt
(time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15) > gpurun_out/r02a/pytest.log 2>&1
tail -30 gpurun_out/r02a/pytest.log
(time python bench.py) > gpurun_out/r02a/bench_default.log 2>&1
tail -3 gpurun_out/r02a/bench_default.log
(time python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline) > gpurun_out/r02a/bench_driver.log 2>&1
tail -2 gpurun_out/r02a/bench_driver.log
(time TGX_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5) > gpurun_out/r02a/bench_n2_shared.log 2>&1
tail -3 gpurun_out/r02a/bench_n2_shared.log
