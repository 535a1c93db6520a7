#!/usr/bin/env python3
"""Overlap of the previous Fiedler vector with the next one along a bench trajectory run with use_cache semantics (the solver's
debug line "warm start: overlap ..."; solver.h warm_skip).  usage: warm_probe.py cfg  2>&1 | grep "warm start"'"""
import sys
sys.path.insert(0, ".")
import bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1]
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"]); P.set_option("debug", 1)
r = P.fw_run(w["k"], 20, warm_start=True)
print("steps", [int(st.lanczos_steps) for st in r["stats"]], file=sys.stderr)
