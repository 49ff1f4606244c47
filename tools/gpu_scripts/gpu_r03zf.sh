#!/bin/bash
# round 3: closing tables after the LDS-DMA ring skinny kernel: batch decode, short prompts, B = 32 / 64 traces, tests
O=gpurun_out/r03zf; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(timeout 2000 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
echo "# final_512";  python $R/tools/batch_bench.py --prompt 512 --batches 1,2,4,5,8,12,16,17,24,32,48,64 --steps 96 2>&1 | grep "B="
echo "# final_2k";   python $R/tools/batch_bench.py --prompt 2048 --batches 4,8,16,24,32,48,64 --steps 96 2>&1 | grep "B="
echo "# final_mistral"; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --prompt 512 --batches 4,8,16,24,32,48,64 --steps 48 2>&1 | grep "B="
echo "# final_3b"; python $R/tools/batch_bench.py --model llama-3.2-3b --prompt 512 --batches 4,8,16,24,32,48,64 --steps 48 2>&1 | grep "B="
echo "# final_qwen05"; python $R/tools/batch_bench.py --model qwen2.5-0.5b --prompt 512 --batches 8,16,32,64 --steps 48 2>&1 | grep "B="
echo "# final_qwen3"; python $R/tools/batch_bench.py --model qwen3-1.7b --prompt 512 --batches 8,16,32,64 --steps 48 2>&1 | grep "B="
for S in 8 16 24 32 33 40 48 56 64 96 128; do echo -n "llama-3.2-1b "; python $R/tools/prefill_bench.py --seq $S --reps 6 | tail -1; done
for m in mistral-7b-v0.3 llama-3.2-3b qwen2.5-0.5b; do echo -n "$m "; python $R/tools/prefill_bench.py --model $m --seq 48 --reps 5 | tail -1; done
cd /tmp
for B in 32 64; do
rocprofv3 --kernel-trace --stats -d /tmp/b$B -o b -- python $R/tools/batch_bench.py --batches $B --steps 48 > $R/$O/b$B.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b$B -name "*.db" | head -1) > $R/$O/b${B}_kernel_stats.txt 2>&1; head -16 $R/$O/b${B}_kernel_stats.txt | cut -c1-180
done
rocprofv3 --kernel-trace --stats -d /tmp/s48 -o b -- python $R/tools/prefill_bench.py --seq 48 --reps 20 > $R/$O/s48.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/s48 -name "*.db" | head -1) > $R/$O/s48_kernel_stats.txt 2>&1; head -14 $R/$O/s48_kernel_stats.txt | cut -c1-180
