#!/bin/bash
# round 4: the exact chain + closures preconditioner without rocSOLVER: inverse microbench, parity tests, intel both ways, kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 tools/bin/ubench_gj > gpurun_out/r4_ubench_gj.txt 2>&1; cat gpurun_out/r4_ubench_gj.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "exact_chain or stiff_chain or preconditioned or pose_graph_fiedler or lobpcg or round2_advice" 2>&1 | tail -3
for m in auto lobpcg; do MACHIP_SOLVER=$m timeout 200 python bench.py --config c3 --steps 20 --warmup 2 --no-cpu --no-pmc --min-seconds 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"c3\", \"$m\", d[\"value\"], d[\"eig_ms_per_iter\"], d[\"lanczos_steps_per_iter\"])"; done
MACHIP_SOLVER=lobpcg bash tools/kstats.sh 16 python bench.py --config c3 --steps 20 --warmup 0 --no-cpu --no-pmc --min-seconds 0 --max-repeats 1 > gpurun_out/r4_c3_lobpcg_kstats.txt 2>&1; cat gpurun_out/r4_c3_lobpcg_kstats.txt
