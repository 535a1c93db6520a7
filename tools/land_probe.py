#!/usr/bin/env python3
"""Per-iterate Lanczos steps of a bench trajectory with and without the landscape weighting of the cold start (option start_land).
usage: land_probe.py cfg ["start_land=0" "start_land=3,start_pow=128" ...]"""
import sys
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1]
sets = sys.argv[2:] or ["start_land=0", "-"]
w = bench.make_workload(cfg)
for a in sets:
    s = {} if a == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")}
    with _lib.default_options(**s):
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    P.set_x(w["x0"])
    rec = bench.run_pass(P, w["k"], 20, w["x0"])
    print(cfg, a, "steps", [x["steps"] for x in rec], "sum", sum(x["steps"] for x in rec))
    print("   lambda2", ["%.6f" % x["f"] for x in rec])
