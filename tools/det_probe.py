#!/usr/bin/env python3
"""Run-to-run determinism of one eigen-solve at a dense config-4 iterate, gather step vs column-panel step."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w = bench.make_workload("c4")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"])
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    P.fw_step(w["k"], it); P.fw_commit()
for mode in ("0", "1"):
    os.environ["MACHIP_PANEL"] = mode
    outs = []
    for rep in range(6):
        P.assemble()
        lam, v, _ = P.fiedler()
        outs.append((lam, float(v @ np.arange(len(v))), int(P.stats.lanczos_steps)))
    print("MACHIP_PANEL=" + mode, "identical" if all(o == outs[0] for o in outs) else "DIFFERENT", [f"{o[0]:.17g}/{o[2]}" for o in outs])
