#!/bin/bash
# Round 4 (late): band split of the column-panel form (diagonal + chain neighbours out of the tiles, added by k_pan_fin): phase clocks + bench
set -u
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench7.hip -o /tmp/ubench7 || exit 1
{ for b in 0 1; do echo "##### band=$b"; timeout 120 /tmp/ubench7 12 21 0 20 $b | grep -E "hash|us per step|slowest|fastest|wave 15 done|padding"; done; } > gpurun_out/r4_band.txt 2>&1
cat gpurun_out/r4_band.txt
python -m pytest tests -q -m gpu -x -k "panel or solver_variants or teacher_forced_config4" 2>&1 | tail -3
for b in 0 1; do MACHIP_PANEL_BAND=$b python bench.py --steps 20 --warmup 3 --no-same-node --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('band=$b', round(d['value'],1), d['roofline']['frac'], d['eig_ms_per_iter'], d['lanczos_steps_per_iter'], d['lambda2_first_last'])"; done | tee -a gpurun_out/r4_band.txt
