#!/bin/bash
# A/B of two builds on ONE box in ONE gpurun call (the pool's boxes differ by up to 5 %, so numbers from different calls cannot be compared):
#   git worktree add -f .ab_old <commit> && (cd .ab_old && python -c "from tinygpt_amd import build as b; b.build_lib()")     # .ab_old/ is git-ignored, it travels with the snapshot
#   gpurun -- 'bash tools/ab.sh <out-tag> "<python tool and args, relative to a repo root>" [rounds]'
# runs the command alternately from .ab_old (A) and from the tree (B) and prints every line the tool prints, tagged.
TAG=$1; CMD=$2; N=${3:-3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $N); do
  (cd $R/.ab_old && python $CMD 2>&1 | sed "s/^/A(old) /") | tee -a $O/ab.txt
  (cd $R && python $CMD 2>&1 | sed "s/^/B(new) /") | tee -a $O/ab.txt
done
