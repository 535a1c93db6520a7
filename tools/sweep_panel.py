#!/usr/bin/env python3
"""In-solve duration of one Lanczos step (machip_solve_stats.step_ms / steps_timed) with the one-kernel gather step
(MACHIP_PANEL=0) against column-panel shapes (panel.h: MACHIP_PANEL=1, MACHIP_PANEL_NP / _NB / _B2 / _G2), on every
iterate of a BASELINE config's Frank-Wolfe run; lambda_2 of every variant is compared with the gather step's.
usage: sweep_panel.py [c4|c2] [iters] [shape indices]"""
import os, sys
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# (panel on?, NP, NB, B2, G2); None = library default for that knob
shapes = [(0, None, None, None, None), (1, None, None, None, None), (1, 9, 28, None, None), (1, 10, 25, None, None),
          (1, 11, 23, None, None), (1, 13, 19, None, None), (1, 14, 18, None, None), (1, 16, 16, None, None),
          (1, 12, 21, 256, None), (1, 12, 21, 1024, None), (1, 12, 21, 512, 128), (1, 12, 21, 512, 196), (1, 12, 20, None, None)]
if len(sys.argv) > 3:
    shapes = [shapes[int(t)] for t in sys.argv[3].split(",")]
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"])
keys = ("MACHIP_PANEL", "MACHIP_PANEL_NP", "MACHIP_PANEL_NB", "MACHIP_PANEL_B2", "MACHIP_PANEL_G2")
print("shapes:", shapes)
tot = np.zeros(len(shapes)); totsteps = np.zeros(len(shapes)); worst = np.zeros(len(shapes))
for it in range(iters):
    row, lams, stepsv = [], [], []
    for sh in shapes:
        for k_, v in zip(keys, sh):
            os.environ.pop(k_, None)
            if v is not None:
                os.environ[k_] = str(v)
        P.assemble()
        lam, _, _ = P.fiedler(want_vec=False)
        st = P.stats
        row.append(1e3 * st.step_ms / max(1, st.steps_timed)); lams.append(lam); stepsv.append(int(st.lanczos_steps))
    for k_ in keys:
        os.environ.pop(k_, None)
    os.environ["MACHIP_PANEL"] = "0"          # the trajectory itself follows the gather step
    f, d, g = P.fw_step(w["k"], it)
    os.environ.pop("MACHIP_PANEL", None)
    st = P.stats
    tot += np.array(row) * np.array(stepsv); totsteps += np.array(stepsv)
    rel = np.abs(np.array(lams) - lams[0]) / abs(lams[0])
    worst = np.maximum(worst, rel)
    print(f"it {it:2d} nnz {int(st.nnz):8d} steps {stepsv} maxrel {rel.max():.1e} | " + " ".join(f"{v:6.2f}" for v in row), flush=True)
    P.fw_commit()
print("step-weighted mean us/step per shape (worst |dlam|/lam vs shape 0):")
for sh, v, wr in zip(shapes, tot / totsteps, worst):
    print(f"   {str(sh):36s} {v:7.2f}   {wr:.1e}")
