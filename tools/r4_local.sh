#!/bin/bash
# Round 4: XCD-local one-launch panel step (k_pan_step<RPT, true>) against the two-launch form, phase clocks (tools/ubench7.hip)
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench7.hip -o /tmp/ubench7 || exit 1
{
for cfg in "12 21 0 20" "16 16 0 20" "16 16 1 20" "16 16 2 20" "16 16 2 5" "16 16 2 0" "10 24 2 20" "10 24 0 20"; do
  echo "##### NP NB mode spin_us = $cfg   (mode 0: k_pan_mul + k_pan_fin, 1: k_pan_step with sc1 traffic, 2: k_pan_step XCD-local)"
  timeout 120 /tmp/ubench7 $cfg
done
} > gpurun_out/r4_local.txt 2>&1
tail -5 gpurun_out/r4_local.txt
