#!/bin/bash
# round 5: the inter-process row-partitioned eigen-solve on ONE GPU (MACHIP_SHARE_GPU=1) after folding publish + wait into one launch:
# us per Lanczos step, 2 and 4 rank processes, configs[1] and configs[3]; then the IPC tests
cd /root/repo; mkdir -p gpurun_out; out=gpurun_out/r5_ipc.txt; : > $out
for cfg in c2 c4; do for R in 2 4; do
  echo "== $cfg: $R processes on one GPU, --mode ipc_eig" >> $out
  MACHIP_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python bench.py --gpus $R --config $cfg --mode ipc_eig --steps 20 --warmup 2 --min-seconds 0.5 --max-repeats 3 > gpurun_out/r5_ipc_${cfg}_$R.json 2>> $out </dev/null
  python - gpurun_out/r5_ipc_${cfg}_$R.json >> $out <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    ip = d.get("ipc_eig", {})
    print({k: ip.get(k) for k in ("value", "us_per_lanczos_step", "comm_mode", "lambda2_first_last")}, "errors", d.get("errors"))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-400:])
PY
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ipc or dry_run or two_ranks or row_partitioned" 2>&1 | tail -3 >> $out
cat $out
