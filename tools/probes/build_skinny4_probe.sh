#!/bin/bash
# builds tools/probes/build/skinny4_probe_<bits> for the dissection bits of kernels/skinny.h
set -e
cd "$(dirname "$0")"; mkdir -p build
for d in 0 1 2 8 16 17 10 26; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I../../tinygpt_amd/csrc -DTGX_SKINNY_DIS=$d skinny4_probe.hip -o build/skinny4_probe_$d &
done
wait; ls build/skinny4_probe_*
