#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5k
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --warmup 1 --no-cpu --no-pmc --no-warm --no-roofline --min-seconds 0 --max-repeats 1"
MACHIP_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1; echo "graph run rc=$?"
f=$(find $out/trace -name "t_kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/tools/gap_analysis.py "$f" 21 | tee $out/gaps_graph.txt; fi
rm -rf $out/trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- $B > $out/trace2.log 2>&1; echo "eager run rc=$?"
f=$(find $out/trace -name "t_kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/tools/gap_analysis.py "$f" 21 | tee $out/gaps_eager.txt; fi
rm -rf $out/trace
