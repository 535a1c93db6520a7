#!/bin/bash
# round 4: one-launch column-panel step (k_pan_step) against k_pan_mul + k_pan_fin: parity tests, then bench configs[3]
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "panel or teacher_forced_config4 or solver_variants or full_size_config4" 2>&1 | tail -5
for v in "1 20" "1 0" "0 0"; do
  set -- $v
  MACHIP_PANEL_FUSED=$1 MACHIP_PANEL_SPIN_US=$2 timeout 300 python bench.py --config c4 --steps 20 --warmup 2 --no-cpu --no-pmc --min-seconds 1 > gpurun_out/r4_fused_$1_$2.json 2> gpurun_out/r4_fused_$1_$2.err
  echo "fused=$1 spin=$2 rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4_fused_$1_$2.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","lanczos_steps_per_iter","eig_ms_per_iter","lambda2_first_last")}, d["roofline"].get("frac"), d["roofline"].get("step_us"))
except Exception as e: print("ERR", e); print(open("gpurun_out/r4_fused_$1_$2.err").read()[-2000:])
PY
done
