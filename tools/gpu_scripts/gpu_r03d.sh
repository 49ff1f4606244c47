#!/bin/bash
# round 3: L2 prefetch chaining A/B at the bench operating point
mkdir -p gpurun_out/r03d
run() { name=$1; shift; timeout 300 python tools/quick_bench.py --prompt 2048 --steps 256 --ctx 2400 "$@" > gpurun_out/r03d/$name.txt 2>&1; echo "== $name: $(grep 'decode' gpurun_out/r03d/$name.txt)"; }
run base
run pf_default --opt pf.mode=1
run pf_oproj_only --opt pf.mode=1 --opt pf.gu_kb=0 --opt pf.qkv_kb=0
run pf_qkv_only --opt pf.mode=1 --opt pf.gu_kb=0 --opt pf.oproj_kb=0
run pf_gu_only --opt pf.mode=1 --opt pf.oproj_kb=0 --opt pf.qkv_kb=0
run pf_gu1m --opt pf.mode=1 --opt pf.gu_kb=1024
run pf_gu3m --opt pf.mode=1 --opt pf.gu_kb=3072
run pf_dn1m --opt pf.mode=1 --opt pf.dn_kb=1024
run pf_wgs128 --opt pf.mode=1 --opt pf.wgs=128
run pf_wgs32 --opt pf.mode=1 --opt pf.wgs=32
run pf_s64 --opt pf.mode=1 --opt pf.stride=64
run pf_s256 --opt pf.mode=1 --opt pf.stride=256
