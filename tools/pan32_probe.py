#!/usr/bin/env python3
"""Mixed panel mode (machip_set_precision(1): late Lanczos steps on fp32 tile values): steps, fp32 steps, residual of the explicit check and
restarts by the switch threshold (option pan32_switch_e9), on every iterate of the configs[3] trajectory.  usage: pan32_probe.py [thresholds_e9 ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
thr = [int(t) for t in sys.argv[1:]] or [1000000, 100000]
w = bench.make_workload("c4")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"])
P.set_option("panel", 1)
for it in range(20):
    P.set_precision(0)
    P.assemble()
    lam0, _, _ = P.fiedler(want_vec=False)
    s0 = P.stats.asdict()
    row = f"it {it:2d} nnz {s0['nnz']:8d} fp64 steps {s0['lanczos_steps']:4d} res {s0['residual']:.2e} |"
    P.set_precision(1)
    for t in thr:
        P.set_option("pan32_switch_e9", t)
        P.assemble()
        lam, _, _ = P.fiedler(want_vec=False)
        st = P.stats.asdict()
        row += f" [{t * 1e-9:.0e}: steps {st['lanczos_steps']} fp32 {st['steps_lowp']} restarts {st['restarts']} res {st['residual']:.2e} dlam {abs(lam - lam0) / lam0:.1e}]"
    print(row, flush=True)
    P.set_precision(0)
    P.fw_step(w["k"], it); P.fw_commit()
