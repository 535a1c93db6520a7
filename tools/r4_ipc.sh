#!/bin/bash
# round 4: the inter-process row-partitioned eigen-solve on ONE GPU (MACHIP_SHARE_GPU=1: R rank processes share the device):
# completes, same trajectory, and what a step costs with the device-side flags against the same gather step in one process
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; out=gpurun_out/r4_ipc.txt; : > $out
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ("value","n_gpus","ms_per_step","lanczos_steps_per_iter","eig_ms_per_iter","lambda2_first_last")}, d["config"]["parallelism"][:140], "step_us", d.get("roofline",{}).get("avg_launch_us"))
except Exception as e: print("ERR", e, open(sys.argv[1]).read()[-500:])
PY
}
for cfg in c2 c4; do
  echo "== $cfg: one process, gather step (MACHIP_PANEL=0)" >> $out
  MACHIP_PANEL=0 timeout 300 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 1 > gpurun_out/r4_ipc_${cfg}_1.json 2>> $out; show gpurun_out/r4_ipc_${cfg}_1.json >> $out
  for R in 2 4; do
    echo "== $cfg: $R processes on one GPU, IPC row-partitioned eigen-solve" >> $out
    MACHIP_SHARE_GPU=1 timeout 600 python bench.py --gpus $R --config $cfg --steps 20 --warmup 2 --no-cpu --no-pmc --min-seconds 1 > gpurun_out/r4_ipc_${cfg}_$R.json 2>> $out; echo "rc=$?" >> $out; show gpurun_out/r4_ipc_${cfg}_$R.json >> $out
  done
done
cat $out
