mkdir -p gpurun_out/r6v
timeout 900 python -m pytest tests/test_hip_paged.py -x -q -m gpu 2>&1 | tail -5
{
for i in 1 2; do
python tools/batch_bench.py --prompt 2048 --steps 128 --batches 1
python tools/batch_bench.py --prompt 2048 --steps 128 --batches 1 --kv-budget 65536
python tools/batch_bench.py --prompt 256 --steps 128 --batches 1,8,32,64
python tools/batch_bench.py --prompt 256 --steps 128 --batches 1,8,32,64 --kv-budget 65536
done
} 2>&1 | grep "B=" | tee gpurun_out/r6v/paged_cost.txt
