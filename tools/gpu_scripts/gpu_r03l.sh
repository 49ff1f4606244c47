#!/bin/bash
mkdir -p gpurun_out/r03l
run() { name=$1; shift; timeout 300 python tools/quick_bench.py --prompt 2048 --steps 256 --ctx 2400 "$@" > gpurun_out/r03l/$name.txt 2>&1; echo "== $name: $(grep 'decode' gpurun_out/r03l/$name.txt)"; }
OFF="--opt pf.mode=1 --opt pf.oproj_kb=0 --opt pf.gu_kb=0 --opt pf.qkv_kb=0 --opt pf.wgs=128"
run base
run comb1m $OFF --opt pf.comb_kb=1024
run comb2m $OFF --opt pf.comb_kb=2048
run comb3m $OFF --opt pf.comb_kb=3072
run oproj1m $OFF --opt pf.oproj_gu_kb=1024
run oproj2m $OFF --opt pf.oproj_gu_kb=2048
run comb2_oproj1 $OFF --opt pf.comb_kb=2048 --opt pf.oproj_gu_kb=1024
run comb2m_w256 $OFF --opt pf.comb_kb=2048 --opt pf.wgs=256
run comb2m_w64 $OFF --opt pf.comb_kb=2048 --opt pf.wgs=64
