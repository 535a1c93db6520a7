#!/usr/bin/env python3
"""Developer probe: C2 trajectory of the HIP path vs the reference fixture (tests/golden/er10k_solve.npz)."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mac_amd import _lib
from test_gpu_parity import make_er, reference_start_block
g = np.load("tests/golden/er10k_solve.npz")
n = 10000
ci, cj = make_er(n, 0.01, 0)
m, k = len(ci), int(g["k"])
fi = np.arange(n - 1, dtype=np.int32)
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-8
P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(m))
P.set_start(reference_start_block(n)[:, 0].copy())
x0 = np.zeros(m); x0[g["x0_idx"]] = 1.0
P.set_x(x0)
u = np.inf
for it in range(len(g["f_traj"])):
    f, dual, gn = P.fw_step(k, it, tol=tol)
    u = min(u, dual)
    print(it, f"{f:.12f} {g['f_traj'][it]:.12f} rel {abs(f-g['f_traj'][it])/f:.2e} supp {P.stats.support} {g['supp'][it]} res {P.stats.residual:.1e}")
    P.fw_commit()
print("upper", u, float(g["upper"]), abs(u - float(g["upper"])) / u)
w = P.get_x()
print("nnz", np.count_nonzero(w), int(g["unrounded_nnz"]), "head maxdiff", np.abs(w[:2048] - g["unrounded_head"]).max())
r = np.nonzero(P.round_nearest(k, decimals=10))[0]
print("rounded overlap", len(np.intersect1d(r, g["rounded_idx"])), "of", k)
