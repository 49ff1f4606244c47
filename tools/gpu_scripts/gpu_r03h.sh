#!/bin/bash
mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "last_arriving" > gpurun_out/r03h/tests.txt 2>&1; tail -5 gpurun_out/r03h/tests.txt
run() { name=$1; shift; timeout 300 python tools/quick_bench.py --prompt 2048 --steps 256 --ctx 2400 "$@" > gpurun_out/r03h/$name.txt 2>&1; echo "== $name: $(grep 'decode' gpurun_out/r03h/$name.txt)"; }
run base
run ticket --opt attn.fold_ticket=1
run base2
run ticket2 --opt attn.fold_ticket=1
timeout 300 python tools/quick_bench.py --model qwen2.5-0.5b --prompt 1024 --steps 256 --ctx 1400 > gpurun_out/r03h/q_base.txt 2>&1; echo "== qwen base: $(grep decode gpurun_out/r03h/q_base.txt)"
timeout 300 python tools/quick_bench.py --model qwen2.5-0.5b --prompt 1024 --steps 256 --ctx 1400 --opt attn.fold_ticket=1 > gpurun_out/r03h/q_ticket.txt 2>&1; echo "== qwen ticket: $(grep decode gpurun_out/r03h/q_ticket.txt)"
timeout 300 python tools/quick_bench.py --model mistral-7b-v0.3 --prompt 2048 --steps 128 --ctx 2400 > gpurun_out/r03h/m_base.txt 2>&1; echo "== mistral base: $(grep decode gpurun_out/r03h/m_base.txt)"
timeout 300 python tools/quick_bench.py --model mistral-7b-v0.3 --prompt 2048 --steps 128 --ctx 2400 --opt attn.fold_ticket=1 > gpurun_out/r03h/m_ticket.txt 2>&1; echo "== mistral ticket: $(grep decode gpurun_out/r03h/m_ticket.txt)"
