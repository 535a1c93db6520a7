#!/usr/bin/env python3
"""CPU emulation (round 6) of the panel step with an 8-byte operand: instead of gathering v_j (which needs the 16-byte record {t, v} AND the
coefficients of the running step), multiply the vector  tt = t_{j-1} - sigma v_{j-1}  (known when the previous step's row kernel ends) and
recover  L v_j = (L tt - (alpha_{j-1} - sigma) w_{j-1}) / beta_j  from the stored product w_{j-1} = L v_{j-1}  (L 1 = 0, so the mean term drops).
The drift d_j = w_j - L v_j obeys d_j = -((alpha_{j-1} - sigma)/beta_j) d_{j-1} + rounding: harmless while |alpha - sigma| < beta.
Prints steps to the stop rule, the final explicit residual, and max drift for both recurrences.   usage: shifted_operand_emulation.py c2|c4 [iterates]"""
import numpy as np, scipy.sparse as sp, sys, os
from scipy.linalg import eigh_tridiagonal
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
cfg = sys.argv[1]; nit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = bench.make_workload(cfg)
n = wl["n"]; ci, cj = wl["ci"], wl["cj"]; m = len(ci)
def lap(x):
    i = np.concatenate([wl["fi"], ci]); j = np.concatenate([wl["fj"], cj]); w = np.concatenate([wl["fw"], x * wl["cw"]])
    keep = w > 1e-10
    i, j, w = i[keep], j[keep], w[keep]
    A = sp.coo_matrix((np.concatenate([w, w]), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n)).tocsr()
    d = np.asarray(A.sum(1)).ravel()
    return (sp.diags(d) - A).tocsr(), d
def run(L, u0, mode, tol=1e-8, maxit=3000, sigma_rule="prev"):
    ninf = abs(L).sum(1).max()
    u = u0 - u0.mean(); v = u / np.linalg.norm(u)
    w = L @ v; t = w.copy(); beta = 0.0
    al = []; be = []; V = [v]
    drift = 0.0; amp = 0.0; A = 1.0
    sigma = 0.0
    for j in range(1, maxit):
        tv = t @ v; tt = t @ t; vv = v @ v
        a = tv
        uu = tt - 2 * a * tv + a * a * vv
        mu = (t.sum() - a * v.sum()) / n
        nrm2 = uu - n * mu * mu
        b = np.sqrt(nrm2); inv = 1.0 / b
        al.append(a)
        if j >= 8 and j % 2 == 0:
            ev, S = eigh_tridiagonal(np.array(al), np.array(be), select='i', select_range=(0, 0))
            est = b * abs(S[-1, 0]) * np.abs(V[-1]).sum()
            if est < 0.95 * tol * ninf:
                y = np.array(V).T @ S[:, 0]
                res = np.abs(L @ y - ev[0] * y).sum() / ninf
                return j, ev[0], res, drift, amp
        be.append(b)
        vn = ((t - a * v) - mu) * inv
        if mode == "direct":
            wn = L @ vn
        else:
            if sigma_rule == "prev": sg = sigma
            else: sg = 0.0
            q = L @ (t - sg * v)
            wn = (q - (a - sg) * w) * inv
            g = abs(a - sg) * inv
            A = g * A + 1.0; amp = max(amp, A)
            if j % 16 == 0:
                drift = max(drift, np.abs(wn - L @ vn).max() / ninf)
            sigma = a
        tn = wn - b * v
        v, w, t = vn, wn, tn
        V.append(v)
    return maxit, None, None, drift, amp
gv = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", {"c2": "er10k_vertices.npz", "c4": "er100k_arpack.npz"}[cfg]))
bits = gv["ref_s_bits"]
z = np.random.RandomState(7).normal(size=(n,))
x = wl["x0"].copy()
for t_ in range(nit):
    L, d = lap(x)
    A_ = (sp.diags(d) - L).tocsr()
    u = 1.0 / d
    for kk in range(3): u = (1.0 + A_ @ u) / d
    z0 = z * (u / u.max()) ** 128
    for start, nm in ((z, "plain"), (z0, "land")):
        r1 = run(L, start, "direct")
        r2 = run(L, start, "shift")
        r3 = run(L, start, "shift", sigma_rule="zero")
        print(t_, nm, "direct steps %d lam %.12g res %.2e | shifted steps %d lam %.12g res %.2e drift %.1e amp %.1f | unshifted steps %d res %s drift %.1e amp %.1e"
              % (r1[0], r1[1], r1[2], r2[0], r2[1], r2[2], r2[3], r2[4], r3[0], "%.2e" % r3[2] if r3[2] is not None else "-", r3[3], r3[4]), flush=True)
    x = x + 2.0 / (t_ + 2) * (np.unpackbits(bits[t_])[:m].astype(np.float64) - x)
