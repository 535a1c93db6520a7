#!/usr/bin/env python3
"""MACHIP_DEBUG trace of ONE cold eigen-solve in the mixed mode. usage: mixed_trace.py cfg [fw_iters_before] [precision]"""
import os, sys
sys.path.insert(0, ".")
cfg = sys.argv[1]; pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0; prec = int(sys.argv[3]) if len(sys.argv) > 3 else 1
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
if len(sys.argv) > 4:      # pre-iterations already in the requested precision
    P.set_precision(prec)
bench.run_pass(P, w["k"], pre, w["x0"]) if pre else P.set_x(w["x0"])
P.set_precision(prec)
os.environ["MACHIP_DEBUG"] = "1"
P.assemble()
try:
    lam, _, _ = P.fiedler(want_vec=False)
    print("lambda2", lam, P.stats.asdict())
except Exception as e:
    print("FAILED", e, P.stats.asdict())
