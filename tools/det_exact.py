#!/usr/bin/env python3
"""Run-to-run bit reproducibility of the exact chain + closures preconditioned mode (round 4: hand-written inverse).
usage: det_exact.py [reps]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from conftest import load_golden
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for nm in ("g2o_intel", "g2o_kitti_05"):
    g = load_golden(nm)
    ref = None; bad = 0
    for rep in range(reps):
        P = _lib.Problem(int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])
        P.set_solver(2)
        P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
        P.set_x(g["x_init"])
        out = []
        for it in range(4):
            f, dual, gn = P.fw_step(int(g["k"]), it)
            out.append((f, dual, gn, int(P.stats.lanczos_steps))); P.fw_commit()
        x = P.get_x(); P.close()
        cur = (np.array(out), x)
        if ref is None: ref = cur
        elif not (np.array_equal(ref[0], cur[0]) and np.array_equal(ref[1], cur[1])):
            bad += 1
            print(nm, "rep", rep, "differs:", (ref[0] - cur[0]).tolist())
    print(nm, "reps", reps, "differing", bad)
