set -x
R=$PWD; O=gpurun_out/r02aa; mkdir -p $O
python tools/prefill_bench.py --reps 5 > $O/pf.log 2>&1
python tools/prefill_bench.py --model mistral-7b-v0.3 --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --model llama-3.2-3b --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --seq 4096 --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --seq 8192 --reps 3 >> $O/pf.log 2>&1
python tools/prefill_bench.py --model gpt2 --seq 1000 --reps 3 >> $O/pf.log 2>&1
cat $O/pf.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/st -o p -- python $R/tools/prefill_bench.py --reps 3 > $R/$O/st.log 2>&1
python $R/tools/rocpd_stats.py $(find $R/$O/st -name "*.db" | head -1) 2>&1 | head -14 | cut -c1-170 > $R/$O/stats.txt; cat $R/$O/stats.txt
rm -rf $R/$O/st
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/p2 -o p -- python $R/tools/prefill_bench.py --reps 2 > $R/$O/p2.log 2>&1
python $R/tools/rocpd_pmc.py $R/$O/p2/p_results.db > $R/$O/pmc.txt 2>&1; grep "gemm_dma\|attn_prefill" $R/$O/pmc.txt | cut -c1-170
rm -rf $R/$O/p2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $R/$O/p3 -o p -- python $R/tools/prefill_bench.py --reps 2 > $R/$O/p3.log 2>&1
python $R/tools/rocpd_pmc.py $R/$O/p3/p_results.db > $R/$O/pmc3.txt 2>&1; grep "gemm_dma\|attn_prefill" $R/$O/pmc3.txt | cut -c1-170
rm -rf $R/$O/p3
