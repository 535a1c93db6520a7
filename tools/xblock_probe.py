#!/usr/bin/env python3
"""What the columns 1..q-1 of find_fiedler_pair's X block are worth: Rayleigh quotient, residual and the angle to the
true eigenvectors, per column, on the golden graphs.  usage: xblock_probe.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, scipy.sparse as sp
from conftest import load_golden
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
for nm in ("er300_x0", "er2000_x0", "er300_xfrac", "er2000_xfrac"):
    g = load_golden(nm)
    n = int(g["n"])
    P = _lib.Problem(n, g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])
    P.set_x(g["x"])
    ip, ix, da = P.laplacian_csr()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    ev, V = np.linalg.eigh(L.toarray())
    lam, v, X = P.fiedler(tol=1e-8, x0=reference_start_block(n)[:, 0].copy(), q=4)
    ninf = abs(L).sum(1).max()
    print(nm, "n", n, "steps", int(P.stats.lanczos_steps), "true", ev[1:6])
    G = X.T @ X
    print("   orthonormality", np.abs(G - np.eye(4)).max(), "vs ones", np.abs(X.sum(0)).max())
    for c in range(4):
        x = X[:, c]; rho = x @ (L @ x); r = L @ x - rho * x
        ang = np.sqrt(max(0.0, 1 - (V[:, c + 1] @ x) ** 2))
        span = np.sqrt(max(0.0, 1 - np.sum((V[:, 1:8].T @ x) ** 2)))
        print(f"   col {c}: rho {rho:.8f} (lambda_{c+2} {ev[c+1]:.8f}) resid_l1/|L| {np.abs(r).sum()/ninf:.2e} sin(angle to v_{c+2}) {ang:.2e} sin(angle to span v_2..v_8) {span:.2e}")
    P.close()
