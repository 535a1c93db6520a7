#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5i
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "pose_graph_fiedler or multi_cell_panel or sweep_lanes" --durations=5 > $out/pytest.log 2>&1; tail -15 $out/pytest.log
