#!/usr/bin/env python3
"""Launched vs used Lanczos steps of a solve (the streamed feeder's surplus) with and without the landscape start; third repetition with the
solver's debug trace.  usage: dbg_surplus.py"""
import sys, os
sys.path.insert(0, ".")
import numpy as np
from mac_amd import _lib
g = np.load(os.path.join("tests", "golden", "er2000_xfrac.npz"))
for opts in ({"start_land": 0}, {}):
    with _lib.default_options(**opts):
        P = _lib.Problem(int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])
    P.set_x(g["x"]); P.set_solver(1)
    for rep in range(3):
        if rep == 2: P.set_option("debug", 1)
        lam, v, _ = P.fiedler(tol=1e-8)
        st = P.stats
        print(opts, "er2000 rep", rep, "steps", st.lanczos_steps, "launched", st.steps_timed, "gpu_ms %.3f" % st.gpu_ms, flush=True)
    P.close()
