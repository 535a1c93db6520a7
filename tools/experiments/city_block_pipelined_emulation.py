#!/usr/bin/env python3
"""Round 5, CPU emulation of the arithmetic a one-launch-per-step BLOCK Lanczos kernel would run (no reorthogonalisation; the
normalisation of block j+1 from Gram matrices of the records of step j -- the block analogue of kernels.h pipe_coefs):
block steps until the Fiedler Ritz pair of the block tridiagonal passes the reference's stop rule on the explicit residual."""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import scipy.linalg as sla

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

def pipelined_block(L, U0, tol=1e-8, maxit=1500, check_every=8):
    lnorm = abs(L).sum(axis=1).max()
    b = U0.shape[1]
    T = U0.copy(); Vp = np.zeros_like(U0)
    As, Bs, basis = [], [], []
    for j in range(maxit):
        X = Vp.T @ T; TT = T.T @ T; M = Vp.T @ Vp; st = T.sum(0); sv = Vp.sum(0)
        A = 0.5 * (X + X.T)
        mu = (st - sv @ A) / n
        G = TT - A @ X - X.T @ A + A @ M @ A - n * np.outer(mu, mu)
        G = 0.5 * (G + G.T)
        try:
            R = np.linalg.cholesky(G).T
        except np.linalg.LinAlgError:
            return -j, 0.0, np.inf
        Ri = np.linalg.inv(R)
        Vn = (T - Vp @ A - mu) @ Ri
        W = L @ Vn
        Tn = W - Vp @ R.T
        if j > 0: As.append(A)
        Bs.append(R)                      # B_j (j = 0: the start block's R, not part of the matrix)
        basis.append(Vn)
        T, Vp = Tn, Vn
        J = len(As)
        if J >= 2 and J % check_every == 0:
            # block tridiagonal of J blocks: A_0..A_{J-1}, B_1..B_{J-1}; the residual estimate needs B_J = Bs[J]
            N = J * b
            H = np.zeros((N, N))
            for k in range(J):
                H[k*b:(k+1)*b, k*b:(k+1)*b] = As[k]
                if k + 1 < J:
                    H[(k+1)*b:(k+2)*b, k*b:(k+1)*b] = Bs[k+1]
                    H[k*b:(k+1)*b, (k+1)*b:(k+2)*b] = Bs[k+1].T
            e, Y = sla.eigh(H, subset_by_index=[0, 0])
            sl = Y[-b:, 0]
            est = np.abs(Bs[J] @ sl).sum() * np.sqrt(n) * 0.8
            if est / lnorm < 4 * tol:
                y = np.hstack(basis[:J]) @ Y[:, 0]
                y -= y.mean(); y /= np.linalg.norm(y)
                rq = y @ (L @ y)
                r = np.abs(L @ y - rq * y).sum() / lnorm
                if r < tol: return J, rq, np.linalg.cond(R)
    return maxit, 0.0, 0.0

rng = np.random.RandomState(7)
X0 = rng.normal(size=(16, n)).T
for it in [int(a) for a in sys.argv[1:]] or [19]:
    L = lap(xs[it]); Lp = lap(xs[it - 1])
    ev, evec = spla.eigsh(Lp, k=6, sigma=-1e-3, which='LM')
    lam_true = spla.eigsh(L, k=2, sigma=-1e-3, which='LM')[0][1]
    for b in (1, 2, 4):
        for name, U0 in (("v2prev+rand", np.hstack([evec[:, 1:2], X0[:, :b - 1]])), ("warm block", evec[:, 1:1 + b].copy())):
            J, rq, cond = pipelined_block(L, U0 - U0.mean(0))
            print(f"iterate {it} b={b} {name}: block steps {J}, rq - lambda_2 = {rq - lam_true:.2e}, cond(R_last) = {cond:.2e}", flush=True)
