#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $out
cd $GRAFT_REPO_ROOT
MACHIP_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --config c2 --steps 5 --warmup 1 --min-seconds 0.2 --max-repeats 3 > $out/share2_c2.json 2> $out/share2_c2.err; echo "rc=$?"; tail -c 1500 $out/share2_c2.err; cut -c1-3000 $out/share2_c2.json
timeout 900 python -m pytest tests -m gpu -x -q -k "bench or ipc or dry" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
