#!/bin/bash
# round 3: FETCH_SIZE calibration by load instruction (tools/probes/fetch_calib_probe.hip), then the prefill GEMMs' PMC passes
O=gpurun_out/r03zc; mkdir -p $O
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/fc -o f -- $R/tools/probes/build/fetch_calib_probe > $R/$O/calib.log 2>&1
python $R/tools/rocpd_pmc.py $(find /tmp/fc -name "*.db" | head -1) > $R/$O/calib_fetch.txt 2>&1; cat $R/$O/calib_fetch.txt | cut -c1-170; tail -1 $R/$O/calib.log
rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python $R/tools/prefill_bench.py --reps 3 > $R/$O/prefill_trace.log 2>&1
python $R/tools/rocpd_stats.py $(find /tmp/pt -name "*.db" | head -1) > $R/$O/prefill_kernel_stats.txt 2>&1; head -12 $R/$O/prefill_kernel_stats.txt | cut -c1-170; grep prefill $R/$O/prefill_trace.log
for ctr in FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pp_$ctr -o p -- python $R/tools/prefill_bench.py --reps 2 > $R/$O/prefill_pmc_$ctr.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pp_$ctr -name "*.db" | head -1) > $R/$O/prefill_pmc_$ctr.txt 2>&1; head -8 $R/$O/prefill_pmc_$ctr.txt | cut -c1-170
done
