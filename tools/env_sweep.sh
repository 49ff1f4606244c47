#!/bin/bash
# runtime (ROCclr) environment knobs against the decode graph: does any of them change what a kernel boundary costs?   gpurun -- 'bash tools/env_sweep.sh <tag>'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p $O; cd $R
run() { echo "== $*"; env "$@" timeout 90 python tools/quick_bench.py --prompt 2048 --steps 256 2>&1 | grep -i "tok/s\|ms/tok\|error" | head -3; }
{
run X=0
run GPU_FLUSH_ON_EXECUTION=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run ROC_ACTIVE_WAIT_TIMEOUT=0
run AMD_DIRECT_DISPATCH=0
run HIP_FORCE_DEV_KERNARG=0
run ROC_USE_FGS_KERNARG=0
run DEBUG_HIP_KERNARG_COPY_OPT=0
run X=0
} 2>&1 | tee $O/env_sweep.txt
