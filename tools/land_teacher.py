#!/usr/bin/env python3
"""Teacher-forced comparison of cold-start settings: Lanczos steps and solve time on the REFERENCE'S OWN 20 iterates of configs[1]
(tests/golden/er10k_vertices.npz) or configs[3] (er100k_arpack.npz) -- the same matrices for every setting, unlike free-running
trajectories, which part at the first near-tie of the top-K selection.
usage: land_teacher.py c2|c4 ["start_land=0" "-" ...]"""
import sys, os
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1]
sets = sys.argv[2:] or ["start_land=0", "-"]
w = bench.make_workload(cfg)
gv = np.load(os.path.join("tests", "golden", {"c2": "er10k_vertices.npz", "c4": "er100k_arpack.npz"}[cfg]))
lam_ref = gv["f_traj"] if cfg == "c2" else gv["lam_traj"]
bits = gv["ref_s_bits"]; m, k = len(w["cw"]), w["k"]
for a in sets:
    s = {} if a == "-" else {kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")}
    with _lib.default_options(**s):
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    for rep in range(2):           # second repetition: warmed-up launch model
        x = w["x0"].copy(); steps = []; ms = []; worst = 0.0
        for i in range(20):
            P.set_x(x)
            lam, _, _ = P.fiedler(tol=1e-8, want_vec=False)
            steps.append(int(P.stats.lanczos_steps)); ms.append(float(P.stats.gpu_ms))
            worst = max(worst, abs(lam - lam_ref[i]) / abs(lam_ref[i]))
            x = x + 2.0 / (i + 2) * (np.unpackbits(bits[i])[:m].astype(np.float64) - x)
    print(cfg, a, "steps", steps, "sum", sum(steps), "solve ms sum %.3f" % sum(ms), "worst rel lambda error %.1e" % worst, flush=True)
    P.close()
