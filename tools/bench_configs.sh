#!/bin/bash
# One bench.py line per BASELINE.json configuration, with the dominant kernel's measured HBM traffic per (model, dtype):
#   1. for every config: a separate `rocprofv3 --pmc FETCH_SIZE` pass of `bench.py --no-graph` -> FETCH_SIZE per gate_up launch, x2 (the gfx950
#      correction of MI355X_MICROARCH.md section HBM) -> profiles/pmc_traffic.json by_config["model:dtype"]
#   2. then `bench.py --model ...` (graph replay, the real line; roofline.traffic picks the figure up) -> one JSON line each
# Usage (on the GPU box): bash tools/bench_configs.sh <out.jsonl>
# configs[0] GPT-2 124M fp32 (the reference's CPU case, also run on the GPU), [1] Qwen2.5-0.5B, [2] the headline, [3] Mistral-7B-v0.3,
# [4] Llama-3.2-3B (per GPU; the 8-replica aggregate is the driver's SCALE run).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/bench_lines.jsonl}
export TMPDIR=/tmp
: > "$OUT"
run_cfg() {   # model dtype prompt steps
  local model=$1 dtype=$2 prompt=$3 steps=$4
  ( cd /tmp && rm -rf /tmp/pmc_$model && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_$model -o f -- \
      python $R/bench.py --model $model --dtype $dtype --prompt $prompt --no-graph --no-cpu-baseline --steps 16 --warmup 2 > /tmp/pmc_$model.log 2>&1 )
  local db=$(find /tmp/pmc_$model -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/pmc_traffic_update.py "$db" $model $dtype $R/profiles/pmc_traffic.json; fi
  timeout 900 python $R/bench.py --model $model --dtype $dtype --prompt $prompt --steps $steps --warmup 16 > /tmp/line_$model.log 2>&1
  tail -1 /tmp/line_$model.log >> "$OUT"
  tail -1 /tmp/line_$model.log | cut -c1-260
}
run_cfg gpt2 fp32 16 256
run_cfg qwen2.5-0.5b bf16 16 256
run_cfg llama-3.2-1b bf16 2048 256
run_cfg mistral-7b-v0.3 bf16 16 128
run_cfg llama-3.2-3b bf16 16 256
