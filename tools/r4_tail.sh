#!/bin/bash
# Round 4: which workgroups of k_pan_mul are late -- the same ones in every run / step, or different ones?  (tools/ubench7.hip, two-launch form)
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench7.hip -o /tmp/ubench7 || exit 1
{ for i in 1 2 3 4; do echo "##### run $i"; timeout 120 /tmp/ubench7 12 21 0 20 | grep -E "us per step|slowest|fastest|wave 15 done"; done; } > gpurun_out/r4_tail.txt 2>&1
cat gpurun_out/r4_tail.txt
