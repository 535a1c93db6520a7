#!/usr/bin/env python3
"""Lanczos steps and device time of a 20-iteration Frank-Wolfe pass by the floor under the landscape weighting (option start_floor_e6; round 6).
usage: floor_probe.py cfg [floors...]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from mac_amd import _lib
cfg = sys.argv[1]
floors = [int(t) for t in sys.argv[2:]] or [0, 1, 10, 100, 1000]
w = bench.make_workload(cfg)
for fl in floors + [-1]:
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    from mac_amd.utils.fiedler import reference_start_block
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    if fl < 0: P.set_option("start_land", 0)
    else: P.set_option("start_floor_e6", fl)
    tot = 0; ms = 0.0; modes = []
    for rep in range(2):
        P.set_x(w["x0"]); tot = 0; modes = []
        t0 = time.perf_counter()
        for it in range(20):
            f, d, g = P.fw_step(w["k"], it); tot += int(P.stats.lanczos_steps); modes.append(P.solve_mode()[0]); P.fw_commit()
        P.synchronize(); ms = 1e3 * (time.perf_counter() - t0)
    print(cfg, "floor_e6", fl if fl >= 0 else "landscape off", "steps", tot, "wall ms per iteration %.3f" % (ms / 20), "modes", "".join(str(m) for m in modes), "lam_last %.12g" % f, flush=True)
    P.close()
