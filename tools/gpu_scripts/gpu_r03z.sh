#!/bin/bash
# round 3: the batch table of the closing build (default options) + the B = 64 kernel trace
O=gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "# final_512";  python $R/tools/batch_bench.py --prompt 512 --batches 1,2,4,8,16,24,32,48,64 --steps 96 2>&1 | grep "B="
echo "# final_2k";   python $R/tools/batch_bench.py --prompt 2048 --batches 4,8,16,24,32,48,64 --steps 96 2>&1 | grep "B="
echo "# final_mistral"; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --prompt 512 --batches 4,8,16,24,32,48,64 --steps 48 2>&1 | grep "B="
echo "# final_3b"; python $R/tools/batch_bench.py --model llama-3.2-3b --prompt 512 --batches 4,8,16,24,32,48,64 --steps 48 2>&1 | grep "B="
echo "# final_qwen05"; python $R/tools/batch_bench.py --model qwen2.5-0.5b --prompt 512 --batches 8,16,32,64 --steps 48 2>&1 | grep "B="
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/b64 -o b -- python $R/tools/batch_bench.py --batches 64 --steps 48 > $R/$O/b64.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b64 -name "*.db" | head -1) > $R/$O/b64_kernel_stats.txt 2>&1; head -22 $R/$O/b64_kernel_stats.txt | cut -c1-180
rocprofv3 --kernel-trace --stats -d /tmp/b32 -o b -- python $R/tools/batch_bench.py --batches 32 --steps 48 > $R/$O/b32.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b32 -name "*.db" | head -1) > $R/$O/b32_kernel_stats.txt 2>&1; head -22 $R/$O/b32_kernel_stats.txt | cut -c1-180
