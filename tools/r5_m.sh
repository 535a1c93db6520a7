#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5m
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "solve or trajectory or petersen or sweep or smoke or assembl" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
for c in c4 c2 c5b c3 c5a; do timeout 300 python tools/ab_multi.py $c 5 "-" 2>&1 | tail -1; done | tee $out/ab.txt
