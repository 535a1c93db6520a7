#!/bin/bash
# the driver's command, timed from outside
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s.%N)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err
echo "rc=$? wall $(echo "$(date +%s.%N) - $t0" | bc) s"
tail -3 gpurun_out/r4_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","lanczos_steps_per_iter","warm_start")})
print(d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["avg_launch_us"])
print(json.dumps(d["cpu_baseline"])[:1800])
print({k:v for k,v in d.items() if k.startswith("speedup")})
PY
