#!/usr/bin/env python3
"""CPU emulation behind the landscape weighting of the cold-start vector (mac_amd/csrc/kernels.h k_land_*, solver.h landscape_start).

Runs the bench trajectory of a config in NumPy / SciPy (plain Lanczos on 1-perp, no re-orthogonalisation, stop when the recurrence's
residual estimate passes the reference's rule nx:232,246) and counts Lanczos steps per solve for several start vectors:
  ref        the reference's start column (RandomState(7).normal)
  warm       the previous iterate's Fiedler vector
  d^-p       ref scaled by degree^-p
  lsK^P      ref scaled by (u_K / max u_K)^P, u_K = K Jacobi sweeps on L u = 1 from u = 1/d   (what the library does: K = 3, P = 128)
  y+0.5z     the exact eigenvector plus noise (overlap 0.89): the ceiling of any start vector
Also prints the participation ratio of the Fiedler vector and its overlap with the previous one.
usage: start_landscape_emulation.py c2|c4|c5a|c5b [iterates]      (c4 takes ~1 min per iterate)
Measured here (steps summed over the first iterates; see profiles/r5_landscape.md):
  c2, 10 iterates: ref 2202, ls2^128 1856, ls3^128 1836, ls4^128 1830, ls6^256 1810, d^-6 2080
  c4, 12 iterates: ref 2878, ls2^128 2434, ls3^128 2336, ls4^256 2290, d^-24 2874, y+0.5z 2182
  c5b (city10000) iterate 0: ref 1650, ls2^64 1464;  c5a (sphere2500) 4 iterates: ref 3492, ls2^64 3470
"""
import sys, os
import numpy as np, scipy.sparse as sp
from scipy.linalg import eigh_tridiagonal
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = bench.make_workload(cfg)
n, ci, cj, k = wl["n"], wl["ci"], wl["cj"], wl["k"]
m = len(ci)


def lap(x):
    i = np.concatenate([wl["fi"], ci]); j = np.concatenate([wl["fj"], cj]); w = np.concatenate([wl["fw"], x * wl["cw"]])
    keep = w > 1e-10
    i, j, w = i[keep], j[keep], w[keep]
    A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsr()
    d = np.asarray(A.sum(1)).ravel()
    return (sp.diags(d) - A).tocsr(), A, d


def lanczos_steps(L, u0, tol=1e-8, maxit=4000, want=False):
    norm_inf = abs(L).sum(1).max()
    u = u0 - u0.mean(); v = u / np.linalg.norm(u)
    al, be = [], []
    V = [v] if want else None
    vprev = np.zeros(n); b = 0.0
    thr = tol * norm_inf / (0.8 * np.sqrt(n))          # ||r||_1 ~ 0.8 sqrt(n) ||r||_2 for a delocalised residual
    for j in range(maxit):
        w = L @ v - b * vprev
        a = v @ w; w -= a * v
        w -= w.mean()
        b2 = np.linalg.norm(w)
        al.append(a)
        if j >= 8 and j % 2 == 1:
            ev, S = eigh_tridiagonal(np.array(al), np.array(be), select="i", select_range=(0, 0))
            if b2 * abs(S[-1, 0]) < thr:
                return j + 1, ev[0], (np.array(V).T @ S[:, 0] if want else None)
        be.append(b2); vprev = v; v = w / b2; b = b2
        if want: V.append(v)
    raise SystemExit("no convergence")


z = np.random.RandomState(7).normal(size=(n,))
x = wl["x0"].copy(); yprev = None; tot = {}
for t in range(nit):
    L, A, d = lap(x)
    r0, lam, y = lanczos_steps(L, z, want=True)
    y /= np.linalg.norm(y)
    cands = {"d^-6": z * (d.min() / d) ** 6, "d^-24": z * (d.min() / d) ** 24, "y+0.5z": y + 0.5 * z / np.linalg.norm(z)}
    if yprev is not None: cands["warm"] = yprev
    u = 1.0 / d
    for kk in range(1, 5):
        u = (1.0 + A @ u) / d
        if kk >= 2:
            for p in (64, 128, 256):
                cands["ls%d^%d" % (kk, p)] = z * (u / u.max()) ** p
    res = {"ref": r0}
    for name, uu in cands.items(): res[name] = lanczos_steps(L, uu)[0]
    for kk, v in res.items(): tot[kk] = tot.get(kk, 0) + v
    print(t, "lambda2 %.4f  participation ratio %.1f  overlap with the previous vector %.3f" % (lam, 1.0 / np.sum(y ** 4), abs(y @ yprev) if yprev is not None else 0.0), res, flush=True)
    g = (y[ci] - y[cj]) ** 2 * wl["cw"]
    s = np.zeros(m); s[np.argpartition(g, -k)[-k:]] = 1
    x = x + 2.0 / (t + 2.0) * (s - x); yprev = y
print("TOTAL", tot)
