#!/bin/bash
# the prefill attention probe in every built configuration (tools/probes/build/attn_probe_*), then the prefill parity tests and the A/B of the key split in the model
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for S in 2048 4096 8192; do for b in tools/probes/build/attn_probe_*; do timeout 60 $b $S; done; done > $O/attn_probe.txt 2>&1
cat $O/attn_probe.txt
for o in 0 1 0 1; do python tools/prefill_bench.py --reps 4 --opts "prefill.attn_ksplit=$o" 2>&1 | tail -2 | sed "s/^/ksplit=$o /"; done | tee $O/prefill_ab.txt
for m in mistral-7b-v0.3 llama-3.2-3b; do for o in 0 1; do python tools/prefill_bench.py --model $m --reps 3 --opts "prefill.attn_ksplit=$o" 2>&1 | tail -1 | sed "s/^/ksplit=$o /"; done; done | tee -a $O/prefill_ab.txt
(timeout 900 python -m pytest tests/test_hip_prefill.py -x -q -m gpu 2>&1 | tail -5) | tee $O/pytest_prefill.txt
