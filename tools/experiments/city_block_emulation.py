#!/usr/bin/env python3
"""Round 5, CPU emulation: block LOBPCG (block size b) on city10000's Frank-Wolfe iterates with the chain preconditioner T^-1 and the
multiplicative two-level one (aggregates of consecutive chain nodes): iterations until the FIEDLER pair passes the reference's stop rule."""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.linalg as sla

g = np.load('tests/golden/g2o_city10000.npz'); V = np.load('tests/golden/city10000_vertices.npz')
n = int(g['n']); m = len(g['cw'])
def lap(x):
    act = x > 1e-10
    i = np.r_[g['fi'], g['ci'][act]]; j = np.r_[g['fj'], g['cj'][act]]; w = np.r_[g['fw'], g['cw'][act] * x[act]]
    A = sp.coo_matrix((np.r_[-w, -w], (np.r_[i, j], np.r_[j, i])), shape=(n, n)).tocsr()
    return A + sp.diags(-np.asarray(A.sum(axis=1)).ravel())
xs = [V['x_init'].astype(float)]
for it in range(19):
    s = np.zeros(m); s[V['ref_s'][it]] = 1.0
    xs.append(xs[-1] + 2.0 / (it + 2) * (s - xs[-1]))

def blobpcg(L, M, X, tol=1e-8, maxit=2000):
    lnorm = abs(L).sum(axis=1).max()
    X = X - X.mean(0); X, _ = np.linalg.qr(X)
    P = None
    for it in range(maxit):
        LX = L @ X
        H = X.T @ LX; e, Y = np.linalg.eigh(H); X = X @ Y; LX = LX @ Y
        R = LX - X * e
        if np.abs(R[:, 0]).sum() / lnorm < tol:
            return it, e[0]
        W = M(R); W -= W.mean(0)
        S = np.hstack([X, W] if P is None else [X, W, P])
        Q, _ = np.linalg.qr(S)
        Hs = Q.T @ (L @ Q)
        es, Ys = np.linalg.eigh(Hs)
        b = X.shape[1]
        Xn = Q @ Ys[:, :b]
        P = Xn - X @ (X.T @ Xn)
        X = Xn
    return maxit, e[0]

rng = np.random.RandomState(7)
X0 = rng.normal(size=(16, n)).T
for it in [int(a) for a in sys.argv[1:]] or [0, 5, 19]:
    L = lap(xs[it]).tocsr()
    lnorm = abs(L).sum(axis=1).max(); sigma = 2.5e-7 * lnorm
    d = L.diagonal() + sigma; off = L.diagonal(1)
    Tb = np.zeros((3, n)); Tb[1] = d; Tb[0, 1:] = off; Tb[2, :-1] = off
    Tinv = lambda r: sla.solve_banded((1, 1), Tb, r)
    out = [f"iterate {it}"]
    for nc in (0, 256, 1024):
        if nc:
            agg = (np.arange(n) * nc // n)
            Pm = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
            Aci = np.linalg.inv((Pm.T @ L @ Pm).toarray() + sigma * np.eye(nc) * (n / nc))
            def M(r, Pm=Pm, Aci=Aci):
                z = Pm @ (Aci @ (Pm.T @ r))
                return z + Tinv(r - L @ z)
        else:
            M = Tinv
        res = []
        for b in (1, 4, 8, 16):
            res.append(f"b={b}: {blobpcg(L, M, X0[:, :b].copy())[0]}")
        out.append((f"two-level nc={nc}" if nc else "T^-1") + " [" + ", ".join(res) + "]")
    print("; ".join(out), flush=True)
