set -x
O=gpurun_out/r02q; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for i in 1 2 3; do timeout 300 python tools/dma_check.py llama-3.2-1b 2 2>&1 | grep "rel diff" | sort | uniq -c; done > $O/repeat.log 2>&1; cat $O/repeat.log
python tools/prefill_bench.py --reps 4 > $O/pf.log 2>&1; cat $O/pf.log
python tools/prefill_bench.py --model mistral-7b-v0.3 --reps 3 --seq 2048 > $O/pf7b.log 2>&1; cat $O/pf7b.log
python tools/prefill_bench.py --model mistral-7b-v0.3 --reps 3 --seq 2048 --opts "prefill.gemm_dma=0" >> $O/pf7b.log 2>&1; tail -3 $O/pf7b.log
python tools/prefill_bench.py --model gpt2 --reps 3 --seq 1000 > $O/pfgpt2.log 2>&1; cat $O/pfgpt2.log
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/$O/p1 -o p -- python $R/tools/prefill_bench.py --reps 3 > $R/$O/p1.log 2>&1
python $R/tools/rocpd_stats.py $R/$O/p1/p_results.db > $R/$O/p1_stats.txt 2>&1; head -10 $R/$O/p1_stats.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$O/p2 -o p -- python $R/tools/prefill_bench.py --reps 2 > $R/$O/p2.log 2>&1
python $R/tools/rocpd_pmc.py $R/$O/p2/p_results.db > $R/$O/p2_pmc.txt 2>&1; grep "gemm_dma\|attn_prefill" $R/$O/p2_pmc.txt | cut -c1-170
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $R/$O/p3 -o p -- python $R/tools/prefill_bench.py --reps 2 > $R/$O/p3.log 2>&1
python $R/tools/rocpd_pmc.py $R/$O/p3/p_results.db > $R/$O/p3_pmc.txt 2>&1; grep "gemm_dma\|attn_prefill" $R/$O/p3_pmc.txt | cut -c1-170
