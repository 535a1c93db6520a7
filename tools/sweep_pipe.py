#!/usr/bin/env python3
"""In-solve duration of one fused Lanczos step (machip_solve_stats.step_ms / steps_timed) for a list of launch
shapes, on every iterate of a BASELINE config's Frank-Wolfe run.  usage: sweep_pipe.py [c4|c2] [iters]"""
import os, sys
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = [None, (4, 1, 1024, 256), (4, 2, 1024, 256), (8, 2, 1024, 256), (8, 4, 1024, 256), (16, 2, 1024, 256), (16, 4, 1024, 256),
          (8, 4, 512, 512), (8, 4, 512, 1024), (16, 2, 512, 512), (16, 2, 512, 1024), (4, 2, 512, 1024), (8, 2, 256, 1024),
          (16, 2, 1024, 512), (8, 2, 1024, 512), (8, 4, 1024, 512), (4, 4, 512, 1024), (8, 4, 256, 512), (8, 4, 256, 1024),
          (4, 2, 256, 512), (4, 2, 256, 1024), (16, 2, 256, 1024), (4, 1, 256, 1024)]
if len(sys.argv) > 3:      # restrict to a few shapes: indices
    shapes = [shapes[int(t)] for t in sys.argv[3].split(",")]
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"])
keys = ("MACHIP_G", "MACHIP_UNROLL", "MACHIP_BLOCK", "MACHIP_MAXGRID")
print("shapes:", shapes)
tot = np.zeros(len(shapes)); totsteps = 0
for it in range(iters):
    row = []
    for sh in shapes:
        for k_ in keys:
            os.environ.pop(k_, None)
        if sh is not None:
            for k_, v in zip(keys, sh):
                os.environ[k_] = str(v)
        P.assemble()
        lam, _, _ = P.fiedler(want_vec=False)
        st = P.stats
        row.append(1e3 * st.step_ms / max(1, st.steps_timed))
        steps = int(st.lanczos_steps)
    for k_ in keys:
        os.environ.pop(k_, None)
    f, d, g = P.fw_step(w["k"], it)
    st = P.stats
    tot += np.array(row) * steps; totsteps += steps
    print(f"it {it:2d} nnz {int(st.nnz):8d} steps {steps:4d} best {int(np.argmin(row)):2d} | " + " ".join(f"{v:6.2f}" for v in row), flush=True)
    P.fw_commit()
print("step-weighted mean us/step per shape:")
for sh, v in zip(shapes, tot / totsteps):
    print(f"   {str(sh):28s} {v:7.2f}")
