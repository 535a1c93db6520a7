#!/usr/bin/env python3
"""Same-process, same-GPU A/B of an environment knob of libmachip: alternating passes of 20 Frank-Wolfe iterations.
usage: ab_env.py cfg VAR valueA valueB [rounds]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg, var, va, vb = sys.argv[1:5]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
res = {va: [], vb: []}
us = {va: [], vb: []}
for r in range(rounds + 1):
    for v in (va, vb):
        os.environ[var] = v
        P.set_x(w["x0"]); P.synchronize()
        t0 = time.perf_counter()
        rec = bench.run_pass(P, w["k"], 20, w["x0"])
        P.synchronize()
        el = time.perf_counter() - t0
        if r > 0:
            res[v].append(20 / el)
            us[v].append(1e3 * sum(x["step_ms"] for x in rec) / max(1, sum(x["steps_timed"] for x in rec)))
print(f"{cfg} {var}: " + "; ".join(f"{v}: {np.median(res[v]):.1f} it/s, {np.median(us[v]):.2f} us/step" for v in (va, vb)))
