#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5i
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "bench or dry or two_ranks or fw_run or ipc" --durations=5 > $out/pytest.log 2>&1; tail -12 $out/pytest.log
