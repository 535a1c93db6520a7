#!/bin/bash
# tail-less chunks: A/B against tailless=0 on every Lanczos config, then the parity subset
cd /root/repo
mkdir -p gpurun_out
{
for c in c4 c2 c5b c3; do timeout 300 python tools/ab_multi.py $c 5 "-" "tailless=0" 2>&1 | tail -4; done
} > gpurun_out/r5_o_ab.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fiedler or fw or lanczos or pose or panel or mixed or start or sweep" 2>&1 | tail -8 > gpurun_out/r5_o_tests.txt
cat gpurun_out/r5_o_ab.txt gpurun_out/r5_o_tests.txt
