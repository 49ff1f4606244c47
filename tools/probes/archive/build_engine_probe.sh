#!/bin/bash
# builds tools/probes/build/engine_probe (and prints the engine kernel's register / scratch use)
set -e
cd "$(dirname "$0")"
mkdir -p build
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../tinygpt_amd/csrc engine_probe.hip -o build/engine_probe -save-temps=obj 2>&1 | grep -E "error|warning: v" || true
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|private_segment_fixed_size|name):" build/engine_probe-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - | grep engine
