#!/bin/bash
# Round 4 (late): with the band out of the tiles -- does the one-launch step win now?  shapes?  gather/panel cross-over?
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench7.hip -o /tmp/ubench7 || exit 1
{
for cfg in "12 21 0 20 1" "12 21 1 20 1" "12 21 1 5 1" "16 16 2 20 1" "16 16 0 20 1" "10 24 0 20 1" "14 18 0 20 1" "12 20 0 20 1"; do echo "##### NP NB mode spin band = $cfg"; timeout 120 /tmp/ubench7 $cfg | grep -E "us per step|wave 15 done|slices finished|ticket|drained"; done
} > gpurun_out/r4_band2.txt 2>&1
cat gpurun_out/r4_band2.txt
python -m pytest tests -q -m gpu -x -k "panel or solver_variants or teacher_forced_config4" 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 3 --no-same-node --no-cpu"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), d["roofline"]["frac"], d["eig_ms_per_iter"], d["lanczos_steps_per_iter"])'
{
MACHIP_PANEL_FUSED=1 $B 2>/dev/null | tail -1 | python -c "$P" fused
for mm in 120 140 155 170; do MACHIP_PANEL_MIN_MEAN10=$mm $B 2>/dev/null | tail -1 | python -c "$P" min_mean10=$mm; done
} | tee -a gpurun_out/r4_band2.txt
