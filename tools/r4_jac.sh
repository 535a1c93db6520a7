#!/bin/bash
# round 4: Jacobi-preconditioned LOBPCG (MACHIP_SOLVER=jacobi) against the Lanczos path on the ER bench configurations
mkdir -p gpurun_out
for cfg in c4 c2; do
  for sol in auto jacobi; do
    MACHIP_SOLVER=$sol timeout 300 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu --no-pmc --min-seconds 1 > gpurun_out/r4_jac_${cfg}_${sol}.json 2> gpurun_out/r4_jac_${cfg}_${sol}.err
    echo "$cfg $sol rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4_jac_${cfg}_${sol}.json").read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","lanczos_steps_per_iter","eig_ms_per_iter","lambda2_first_last")})
except Exception as e: print("ERR", e); print(open("gpurun_out/r4_jac_${cfg}_${sol}.err").read()[-2000:])
PY
  done
done
