#!/usr/bin/env python3
"""Round 6: in-solve duration of one Lanczos step (machip_solve_stats.step_ms / steps_timed) of the column-panel step in RECORD form
(panel.h: 16-byte records, k_pan_mul + k_pan_fin) against the SHIFTED recurrence (panel_u.h: 8-byte operand, k_pan_mul8 + k_pan_finu)
in several panel shapes, on every iterate of a BASELINE config's Frank-Wolfe run (the trajectory follows the library's defaults);
lambda_2 and the step count of every variant are compared with the first one's.
usage: sweep_panel_u.py [c4|c2] [iters] [variant indices]"""
import sys
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
AUTO = None
variants = [
    ("record 12x21", dict(panel=1, panel_u=0)),
    ("u 12x21", dict(panel=1, panel_u=1)),
    ("u 10x25", dict(panel=1, panel_u=1, panel_np=10, panel_nb=25)),
    ("u 9x28", dict(panel=1, panel_u=1, panel_np=9, panel_nb=28)),
    ("u 8x32", dict(panel=1, panel_u=1, panel_np=8, panel_nb=32)),
    ("u 7x36", dict(panel=1, panel_u=1, panel_np=7, panel_nb=36)),
    ("u 6x42", dict(panel=1, panel_u=1, panel_np=6, panel_nb=42)),
    ("u 12x21 fin256", dict(panel=1, panel_u=1, panel_b2=256)),
    ("u 8x32 fin256", dict(panel=1, panel_u=1, panel_np=8, panel_nb=32, panel_b2=256)),
    ("u 6x42 fin256", dict(panel=1, panel_u=1, panel_np=6, panel_nb=42, panel_b2=256)),
    ("gather", dict(panel=0)),
    ("u 6x42 fin1024", dict(panel=1, panel_u=1, panel_np=6, panel_nb=42, panel_b2=1024)),
    ("u 6x42 fin512 g2=128", dict(panel=1, panel_u=1, panel_np=6, panel_nb=42, panel_g2=128)),
    ("u 6x42 fin256 g2=256", dict(panel=1, panel_u=1, panel_np=6, panel_nb=42, panel_b2=256, panel_g2=256)),
]
if len(sys.argv) > 3:
    variants = [variants[int(t)] for t in sys.argv[3].split(",")]
keys = sorted({k for _, o in variants for k in o})
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"])
print("variants:", [nm for nm, _ in variants])
nv = len(variants)
tot = np.zeros(nv); totsteps = np.zeros(nv); worst = np.zeros(nv)
for it in range(iters):
    row, lams, stepsv = [], [], []
    for nm, o in variants:
        for k_ in keys: P.set_option(k_, o.get(k_))          # None = automatic
        P.assemble()
        lam, _, _ = P.fiedler(want_vec=False)
        lam, _, _ = P.fiedler(want_vec=False)                # (second solve: shape-dependent buffers exist, launch-time history is this form's)
        st = P.stats
        row.append(1e3 * st.step_ms / max(1, st.steps_timed)); lams.append(lam); stepsv.append(int(st.lanczos_steps))
    for k_ in keys: P.set_option(k_, None)
    f, d, g = P.fw_step(w["k"], it)
    st = P.stats
    tot += np.array(row) * np.array(stepsv); totsteps += np.array(stepsv)
    rel = np.abs(np.array(lams) - lams[0]) / abs(lams[0])
    worst = np.maximum(worst, rel)
    print(f"it {it:2d} nnz {int(st.nnz):8d} steps {stepsv} maxrel {rel.max():.1e} | " + " ".join(f"{v:6.2f}" for v in row), flush=True)
    P.fw_commit()
print("step-weighted mean us/step per variant (worst |dlam|/lam vs variant 0; total steps):")
for (nm, _), v, wr, ts in zip(variants, tot / totsteps, worst, totsteps):
    print(f"   {nm:20s} {v:7.2f}   {wr:.1e}   {int(ts)}")
