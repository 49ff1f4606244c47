cd /root/repo; mkdir -p gpurun_out
(time TGX_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5) > gpurun_out/bench_n2_shared.log 2>&1
tail -4 gpurun_out/bench_n2_shared.log | cut -c1-400
