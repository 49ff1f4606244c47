#!/bin/bash
# One parametrised job script for the GPU box (replaces round 1-3's tools/gpu_scripts/gpu_r0*.sh one-offs):
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <out-tag> <job> [<job> ...]'
# Every job writes under gpurun_out/<out-tag>/ (merged back by gpurun); summaries worth keeping are copied into profiles/ by hand.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R

job_suite()   { (time timeout 2700 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -5 $O/pytest.log; }
job_smoke()   { python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log; }
job_bench()   { python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/bench_line.json; cut -c1-600 $O/bench_line.json; }
job_quick()   { python tools/quick_bench.py --prompt 2048 --steps 256 ${QB_ARGS:-} > $O/quick.log 2>&1; cat $O/quick.log; }
# the attention pair at the bench's operating point: splits per kv head x o_proj form (K-sliced with the merge in its prologue / combine + row-sliced)
job_attn_pair() {
  for ns in 4 6 8 12 17 32; do
    python tools/sweep.py --prompt 2048 --steps 256 --pre "attn.nsplit=$ns" --grid "oproj.sliced=0,1" 2>&1 | sed "s/^/nsplit=$ns /"
  done > $O/attn_pair.log 2>&1; cat $O/attn_pair.log
}
job_kernarg_probe() { for b in kernarg_probe kernarg_probe_pre; do echo "== $b"; timeout 120 tools/probes/build/$b; done > $O/kernarg_probe.log 2>&1; cat $O/kernarg_probe.log; }
# rocprofv3 kernel trace of the bench command (eager launches: rocprofv3 cannot trace hipGraphLaunch) + the per-class sequence table
job_trace() {
  cd /tmp; rm -rf /tmp/tr
  rocprofv3 --kernel-trace --stats -d /tmp/tr -o tr -- python $R/bench.py --no-graph --no-cpu-baseline > $O/trace_bench.log 2>&1
  db=$(find /tmp/tr -name "*.db" | head -1)
  python $R/tools/rocpd_stats.py $db > $O/kernel_stats.txt 2>&1; python $R/tools/rocpd_seq.py $db >> $O/kernel_stats.txt 2>&1
  tail -1 $O/trace_bench.log >> $O/kernel_stats.txt; head -24 $O/kernel_stats.txt | cut -c1-200; cd $R
}

job_atomic_probe() { timeout 120 tools/probes/build/atomic_probe > $O/atomic_probe.log 2>&1; cat $O/atomic_probe.log; }
job_lab() { for g in ${LAB_GEOMS:-1b}; do for ns in ${LAB_NSPLIT:-0}; do timeout 300 tools/probes/build/layer_lab $g ${LAB_POS:-2064} 16 $([ $ns -gt 0 ] && echo $ns); done; done > $O/lab.log 2>&1; cat $O/lab.log; }

# launch-geometry axes of the GEMV classes on one model (tools/sweep.py): SWEEP_MODEL, SWEEP_PROMPT
job_sweep() {
  for g in "qkv.ks=1,2,4" "qkv.bpc=2,4,8" "oproj.ks=1,2,4" "oproj.bpc=1,2,4" "gateup.ks=1,2" "gateup.bpc=2,4,8" "down.ks=1,2,4" "down.bpc=1,2,4,8"; do
    echo "== $g"; python tools/sweep.py --model ${SWEEP_MODEL:-qwen2.5-0.5b} --prompt ${SWEEP_PROMPT:-16} --steps 64 --grid "$g" 2>&1 | tail -4
  done > $O/sweep.log 2>&1; cat $O/sweep.log
}

# per-kernel durations of the batched prefill (rocprofv3 --kernel-trace; PREFILL_ARGS e.g. "--seq 2048")
job_prefill_trace() {
  cd /tmp; rm -rf /tmp/pt
  rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python $R/tools/prefill_bench.py --reps 4 ${PREFILL_ARGS:-} > $O/prefill_trace.log 2>&1
  python $R/tools/rocpd_stats.py $(find /tmp/pt -name "*.db" | head -1) > $O/prefill_kernels.txt 2>&1
  tail -4 $O/prefill_trace.log; head -14 $O/prefill_kernels.txt | cut -c1-180; cd $R
}

for j in "$@"; do echo "=== job $j"; job_$j; done
