#!/bin/bash
# round 3: default build with the batched MFMA attention from 24 rows: the batch table, a B = 32 kernel trace, the batch / attention tests
O=gpurun_out/r03u; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
echo "# final_512";  python $R/tools/batch_bench.py --prompt 512 --batches 4,8,16,24,32 --steps 96 2>&1 | grep "B="
echo "# final_2k";   python $R/tools/batch_bench.py --prompt 2048 --batches 4,8,16,24,32 --steps 96 2>&1 | grep "B="
echo "# final_mistral"; python $R/tools/batch_bench.py --model mistral-7b-v0.3 --prompt 512 --batches 16,24,32 --steps 48 2>&1 | grep "B="
echo "# final_3b"; python $R/tools/batch_bench.py --model llama-3.2-3b --prompt 512 --batches 16,24,32 --steps 48 2>&1 | grep "B="
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/b32 -o b -- python $R/tools/batch_bench.py --batches 32 --steps 48 > $R/$O/b32.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/b32 -name "*.db" | head -1) > $R/$O/b32_kernel_stats.txt 2>&1; head -12 $R/$O/b32_kernel_stats.txt | cut -c1-180
cd $R
(timeout 2000 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_fuzz.py -m gpu -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
