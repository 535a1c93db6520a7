#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
{
for c in c4 c2 c5b; do timeout 300 python tools/ab_multi.py $c 5 "-" "near_eager=1" 2>&1 | tail -2; done
} > gpurun_out/r5_o_ab.txt 2>&1
cat gpurun_out/r5_o_ab.txt
