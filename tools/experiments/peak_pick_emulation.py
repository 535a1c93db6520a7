#!/usr/bin/env python3
"""CPU emulation (round 6): cheaper stand-ins for the two-stage cold start (ball_start_emulation.py).  The landscape-weighted start spreads its mass
over several peaks; the ball start picks ONE by its local Dirichlet eigenvalue and starts from the local ground state.  Variants that need no local
eigen-solve: the peak is picked by the Rayleigh quotient of the landscape shape restricted to the peak's 2-hop ball, the start is that restricted
shape (plus a 1e-3 floor of the reference's draw).  Steps on the reference's own 20 iterates of configs[3].   usage: peak_pick_emulation.py c4"""
import numpy as np, scipy.sparse as sp, sys, os
from scipy.linalg import eigh_tridiagonal
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
cfg = sys.argv[1]
wl = bench.make_workload(cfg)
n = wl["n"]; ci, cj = wl["ci"], wl["cj"]; m = len(ci)
def lap(x):
    i = np.concatenate([wl["fi"], ci]); j = np.concatenate([wl["fj"], cj]); w = np.concatenate([wl["fw"], x * wl["cw"]])
    keep = w > 1e-10
    i, j, w = i[keep], j[keep], w[keep]
    A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsr()
    d = np.asarray(A.sum(1)).ravel()
    return (sp.diags(d) - A).tocsr(), d, A
def lanczos_steps(L, u0, tol=1e-8, maxit=4000):
    ninf = abs(L).sum(1).max()
    u = u0 - u0.mean(); v = u / np.linalg.norm(u)
    al = []; be = []; vprev = np.zeros(n); b = 0.0; l1 = np.abs(v).sum()
    for j in range(maxit):
        w = L @ v - b * vprev
        a = v @ w; w -= a * v; w -= w.mean()
        b2 = np.linalg.norm(w); al.append(a)
        if j >= 8 and j % 2 == 1:
            ev, S = eigh_tridiagonal(np.array(al), np.array(be), select='i', select_range=(0, 0))
            if b2 * abs(S[-1, 0]) * l1 < 0.95 * tol * ninf: return j + 1
        be.append(b2); vprev = v; v = w / b2; b = b2; l1 = np.abs(v).sum()
    return maxit
gv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", {"c2": "er10k_vertices.npz", "c4": "er100k_arpack.npz"}[cfg]))
bits = gv["ref_s_bits"]
z = np.random.RandomState(7).normal(size=(n,))
x = wl["x0"].copy(); tot = {}
FL = 1e-3
for t in range(20):
    L, d, A = lap(x)
    u = 1.0 / d
    for kk in range(3): u = (1.0 + A @ u) / d
    q = u / u.max()
    w128 = np.maximum(q ** 128, FL)
    res = {"landscape+floor": lanczos_steps(L, z * w128)}
    C = np.argsort(-u)[:16]
    best = None; cand = []
    for c in C:
        ball = np.array([c])
        for h in range(2): ball = np.unique(np.concatenate([ball, A[ball].indices]))
        f = np.zeros(n); f[ball] = q[ball] ** 64
        rq = (f @ (L @ f)) / (f @ f)
        cand.append((rq, c, ball, f))
    cand.sort(key=lambda t_: t_[0])
    rq, c, ball, f = cand[0]
    s1 = f / np.linalg.norm(f) + FL * z / np.sqrt(n) * 0 + FL * z * (np.abs(f).max() / np.abs(z).max())
    res["rq-picked peak, u^64 on its ball + floor"] = lanczos_steps(L, s1)
    f1 = np.zeros(n); b0 = cand[[k for k, t_ in enumerate(cand) if t_[1] == C[0]][0]][2]; f1[b0] = q[b0] ** 64
    res["highest peak only, u^64 on its ball + floor"] = lanczos_steps(L, f1 / np.linalg.norm(f1) + FL * z * (np.abs(f1).max() / np.linalg.norm(f1) / np.abs(z).max()))
    mask = np.zeros(n); mask[ball] = 1.0
    res["landscape weights on the rq-picked ball only + floor"] = lanczos_steps(L, z * np.maximum(q ** 128 * mask, FL))
    for k2, v in res.items(): tot[k2] = tot.get(k2, 0) + v
    print(t, "picked == highest:", c == C[0], res, flush=True)
    x = x + 2.0 / (t + 2) * (np.unpackbits(bits[t])[:m].astype(np.float64) - x)
print("TOTAL", tot)
