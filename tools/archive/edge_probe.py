#!/usr/bin/env python3
"""Developer probe: closed-form and edge cases through the automatic solver selection."""
import sys
import numpy as np
sys.path.insert(0, ".")
from mac_amd import _lib

def path(n, w=1.0):
    fi = np.arange(n - 1, dtype=np.int32)
    # one dummy candidate (0, 2) kept at x = 0 so the handle has m >= 1
    return _lib.Problem(n, fi, fi + 1, np.full(n - 1, w), np.array([0], np.int32), np.array([2], np.int32), np.ones(1))

for n in (257, 1000, 5000, 16384, 20000):
    P = path(n); P.set_x(np.zeros(1))
    for mode in (0, 1, 2):
        P.set_solver(mode)
        try:
            lam, v, _ = P.fiedler()
            exact = 2 - 2 * np.cos(np.pi / n)
            print(f"path n={n} mode={mode}: lam={lam:.12e} exact={exact:.12e} rel={abs(lam-exact)/exact:.1e} steps={P.stats.lanczos_steps} res={P.stats.residual:.1e} ms={P.stats.gpu_ms:.2f}", flush=True)
        except Exception as e:
            print(f"path n={n} mode={mode}: {type(e).__name__}: {e}", flush=True)
    P.close()
# cycle through the candidate: C_n -> 2 - 2 cos(2 pi / n)
n = 4000
fi = np.arange(n - 1, dtype=np.int32)
P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), np.array([0], np.int32), np.array([n - 1], np.int32), np.ones(1))
P.set_x(np.ones(1))
for mode in (0, 1, 2):
    P.set_solver(mode); lam, v, _ = P.fiedler(); exact = 2 - 2 * np.cos(2 * np.pi / n)
    print(f"cycle n={n} mode={mode}: rel={abs(lam-exact)/exact:.1e} steps={P.stats.lanczos_steps} (double eigenvalue)", flush=True)
P.close()
# hub: node 0 connected to everything + chain
n = 3000
ci = np.zeros(n - 2, np.int32); cj = np.arange(2, n, dtype=np.int32)
P = _lib.Problem(n, fi[: n - 1], fi[: n - 1] + 1, np.ones(n - 1), ci, cj, np.full(n - 2, 0.5))
P.set_x(np.ones(n - 2))
res = []
for mode in (0, 1, 2):
    P.set_solver(mode); lam, v, _ = P.fiedler(); res.append(lam)
    print(f"hub n={n} mode={mode}: lam={lam:.12f} steps={P.stats.lanczos_steps} res={P.stats.residual:.1e}", flush=True)
P.close()
