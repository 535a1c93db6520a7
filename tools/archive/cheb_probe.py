#!/usr/bin/env python3
"""One eigen-solve of a pose-graph golden with the solver's trace on stderr.  usage: cheb_probe.py intel|sphere2500 [iterate]"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from conftest import load_golden
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
nm = sys.argv[1] if len(sys.argv) > 1 else "intel"
g = load_golden("g2o_" + nm)
P = _lib.Problem(int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])
P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
P.set_x(g["x_init"])
k = int(g["k"])
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 0):
    P.fw_step(k, it); P.fw_commit()
os.environ["MACHIP_DEBUG"] = "1"
lam, _, _ = P.fiedler(want_vec=False)
st = P.stats
print(f"lam {lam:.12g} steps {st.lanczos_steps} spmv {st.spmv_total} gpu_ms {st.gpu_ms:.3f} res {st.residual:.2e}")
