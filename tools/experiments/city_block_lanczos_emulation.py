#!/usr/bin/env python3
"""Round 5, CPU emulation: block Lanczos (block size b, full reorthogonalisation = the best case) on city10000's Frank-Wolfe iterates:
block steps until the Fiedler Ritz pair passes the reference's stop rule.  cold = random start block (column 0 = the library's start
vector family); warm = the b lowest non-trivial eigenvectors of the PREVIOUS iterate (what a recycling scheme could hand over at best)."""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

g = np.load('tests/golden/g2o_city10000.npz'); V = np.load('tests/golden/city10000_vertices.npz')
n = int(g['n']); m = len(g['cw'])
def lap(x):
    act = x > 1e-10
    i = np.r_[g['fi'], g['ci'][act]]; j = np.r_[g['fj'], g['cj'][act]]; w = np.r_[g['fw'], g['cw'][act] * x[act]]
    A = sp.coo_matrix((np.r_[-w, -w], (np.r_[i, j], np.r_[j, i])), shape=(n, n)).tocsr()
    return (A + sp.diags(-np.asarray(A.sum(axis=1)).ravel())).tocsr()
xs = [V['x_init'].astype(float)]
for it in range(19):
    s = np.zeros(m); s[V['ref_s'][it]] = 1.0
    xs.append(xs[-1] + 2.0 / (it + 2) * (s - xs[-1]))

def block_lanczos(L, X, tol=1e-8, maxit=3000):
    lnorm = abs(L).sum(axis=1).max()
    X = X - X.mean(0); Q, _ = np.linalg.qr(X)
    b = Q.shape[1]
    basis = [Q]
    for it in range(1, maxit + 1):
        W = L @ basis[-1]
        W -= W.mean(0)
        B = np.hstack(basis)
        for _ in range(2): W -= B @ (B.T @ W)
        Qn, R = np.linalg.qr(W)
        if it % 4 == 0 or it < 8:
            H = B.T @ (L @ B)
            e, Y = np.linalg.eigh(H)
            y = B @ Y[:, 0]
            r = L @ y - e[0] * y
            if np.abs(r).sum() / lnorm < tol: return it, e[0]
        if np.abs(np.diag(R)).min() < 1e-12: return -it, 0.0
        basis.append(Qn)
    return maxit, 0.0

rng = np.random.RandomState(7)
X0 = rng.normal(size=(16, n)).T
for it in [int(a) for a in sys.argv[1:]] or [5, 19]:
    L = lap(xs[it]); Lp = lap(xs[it - 1])
    ev, evec = spla.eigsh(Lp, k=10, sigma=-1e-3, which='LM')
    out = [f"iterate {it}: lambda_2..6 of L = " + " ".join(f"{v:.4g}" for v in spla.eigsh(L, k=6, sigma=-1e-3, which='LM')[0][1:])]
    for b in (1, 2, 4, 8):
        c = block_lanczos(L, X0[:, :b].copy())[0]
        w = block_lanczos(L, evec[:, 1:1 + b].copy())[0]
        out.append(f"b={b}: cold {c} warm {w}")
    print("; ".join(out), flush=True)

# start block = [previous Fiedler vector, random columns]: what a first implementation would use
for it in [int(a) for a in sys.argv[1:]] or [5, 19]:
    L = lap(xs[it]); Lp = lap(xs[it - 1])
    ev, evec = spla.eigsh(Lp, k=4, sigma=-1e-3, which='LM')
    out = [f"iterate {it} (start = [v2 of the previous iterate, random])"]
    for b in (2, 4, 6):
        X = np.hstack([evec[:, 1:2], X0[:, :b - 1]])
        out.append(f"b={b}: {block_lanczos(L, X)[0]}")
    print("; ".join(out), flush=True)
