#!/usr/bin/env python3
"""Round 5, CPU emulation (NumPy / SciPy): how many block-size-1 LOBPCG iterations does city10000 need on the reference's own
Frank-Wolfe iterates under different preconditioners?  (a) the odometry chain T = tridiag(L) + sigma I alone (what precond.h runs
beyond the exact mode's closure limit), (b) additive two-level: T^-1 + P Ac^-1 P^T with piecewise-constant aggregates of
consecutive chain nodes (Ac = P^T L P dense, nc x nc), (c) multiplicative two-level.  Same stop rule as the device
(||L v - rho v||_1 / ||L||_inf < 1e-8)."""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import scipy.linalg as sla

g = np.load('tests/golden/g2o_city10000.npz'); V = np.load('tests/golden/city10000_vertices.npz')
n = int(g['n']); m = len(g['cw']); k = int(g['k'])
def lap(x):
    act = x > 1e-10
    i = np.r_[g['fi'], g['ci'][act]]; j = np.r_[g['fj'], g['cj'][act]]; w = np.r_[g['fw'], g['cw'][act] * x[act]]
    A = sp.coo_matrix((np.r_[-w, -w], (np.r_[i, j], np.r_[j, i])), shape=(n, n)).tocsr()
    return A + sp.diags(-np.asarray(A.sum(axis=1)).ravel())
# iterates of the reference run
xs = [V['x_init'].astype(float)]
for it in range(19):
    s = np.zeros(m); s[V['ref_s'][it]] = 1.0
    xs.append(xs[-1] + 2.0 / (it + 2) * (s - xs[-1]))

def lobpcg1(L, M, x0, tol=1e-8, maxit=3000):
    lnorm = abs(L).sum(axis=1).max()
    x = x0 - x0.mean(); x /= np.linalg.norm(x)
    p = None
    for it in range(maxit):
        Lx = L @ x; rho = x @ Lx; r = Lx - rho * x
        if np.abs(r).sum() / lnorm < tol:
            return it, rho
        w = M(r); w -= w.mean()
        S = [x, w] if p is None else [x, w, p]
        S = np.stack(S, 1)
        Q, _ = np.linalg.qr(S)
        H = Q.T @ (L @ Q)
        e, Y = np.linalg.eigh(H)
        xn = Q @ Y[:, 0]
        if xn @ x < 0: xn = -xn
        p = xn - x * (x @ xn)
        x = xn / np.linalg.norm(xn)
    return maxit, rho

rng = np.random.RandomState(7)
x0 = rng.normal(size=(4, n)).T[:, 0].copy()
which = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 5, 10, 19]
for it in which:
    L = lap(xs[it]).tocsr()
    lnorm = abs(L).sum(axis=1).max()
    sigma = 2.5e-7 * lnorm
    d = L.diagonal() + sigma; off = np.r_[L.diagonal(1)]
    Tb = np.zeros((3, n)); Tb[1] = d; Tb[0, 1:] = off; Tb[2, :-1] = off
    Tinv = lambda r: sla.solve_banded((1, 1), Tb, r)
    out = [f"iterate {it}: support {int((xs[it] > 1e-10).sum())}"]
    i1, rho = lobpcg1(L, Tinv, x0); out.append(f"T^-1: {i1}")
    for nc in (64, 256, 1024):
        agg = (np.arange(n) * nc // n)
        P = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
        Ac = (P.T @ L @ P).toarray() + sigma * np.eye(nc) * (n / nc)
        Aci = np.linalg.inv(Ac)
        add = lambda r: Tinv(r) + P @ (Aci @ (P.T @ r))
        def mult(r):
            z = P @ (Aci @ (P.T @ r))
            return z + Tinv(r - L @ z)
        ia, _ = lobpcg1(L, add, x0); im, _ = lobpcg1(L, mult, x0)
        out.append(f"nc={nc}: additive {ia} multiplicative {im}")
    print("; ".join(out) + f"; lambda2 {rho:.6g}", flush=True)
