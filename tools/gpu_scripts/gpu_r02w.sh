set -x
R=$PWD; O=gpurun_out/r02w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/$O/avail.txt 2>&1 || rocprofv3-avail list > $R/$O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $R/$O/avail.txt | sort -u | tr '\n' ' ' > $R/$O/sq_counters.txt
i=0
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/$O/p$i -o p -- python $R/tools/prefill_bench.py --reps 2 > $R/$O/p$i.log 2>&1
  python $R/tools/rocpd_pmc.py $R/$O/p$i/p_results.db 2>&1 | grep "gemm_dma8_kernel\|attn_prefill\|gemm_dma8k\|gemm_dma_kernel" | cut -c1-170 >> $R/$O/pmc.txt
  rm -rf $R/$O/p$i
done
cat $R/$O/pmc.txt
