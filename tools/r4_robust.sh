#!/bin/bash
# Round-4 robustness round on the GPU box: solver-mode fuzz (automatic mode now includes the exact mode on small graphs and the
# hand-written inverse everywhere), soak (repeated suites + bench determinism), determinism of the exact mode
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r4robust
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python tools/fuzz_modes.py 120 0 0 > $out/fuzz_auto.txt 2>&1; tail -2 $out/fuzz_auto.txt
timeout 900 python tools/fuzz_modes.py 40 300 120 > $out/fuzz_modes.txt 2>&1; tail -2 $out/fuzz_modes.txt
timeout 600 python tools/det_exact.py 15 > $out/det_exact.txt 2>&1; tail -2 $out/det_exact.txt
bash tools/soak.sh 3 > $out/soak.txt 2>&1; tail -8 $out/soak.txt
