# paged vs unpaged, alternating, one box:  gpurun -- 'bash tools/probes/paged_cost.sh <tag>'
T=${1:-paged}; mkdir -p gpurun_out/$T
{
for i in 1 2; do
python tools/batch_bench.py --prompt 2048 --steps 128 --batches 1
python tools/batch_bench.py --prompt 2048 --steps 128 --batches 1 --kv-budget 65536
python tools/batch_bench.py --prompt 256 --steps 128 --batches 1,8,32,64
python tools/batch_bench.py --prompt 256 --steps 128 --batches 1,8,32,64 --kv-budget 65536
python tools/prefill_bench.py --reps 4
python tools/prefill_bench.py --reps 4 --kv-budget 8192
python tools/prefill_bench.py --reps 3 --seq 4096
python tools/prefill_bench.py --reps 3 --seq 4096 --kv-budget 8192
done
} 2>&1 | grep "B=\|prefill" | tee gpurun_out/$T/paged_cost.txt
