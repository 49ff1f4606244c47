set -x
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/p1 -o p -- python $R/tools/prefill_bench.py --reps 3 --opts "prefill.gemm_dma=5" > $R/$O/p1.log 2>&1
python $R/tools/rocpd_stats.py $R/$O/p1/p_results.db > $R/$O/p1_stats.txt 2>&1; head -9 $R/$O/p1_stats.txt; grep prefill $R/$O/p1.log
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/p2 -o p -- python $R/tools/prefill_bench.py --reps 2 --opts "prefill.gemm_dma=5" > $R/$O/p2.log 2>&1
python $R/tools/rocpd_pmc.py $R/$O/p2/p_results.db > $R/$O/p2_pmc.txt 2>&1; grep "gemm_dma\|attn_prefill" $R/$O/p2_pmc.txt | cut -c1-170
