#!/usr/bin/env python3
"""Developer fuzz for the landscape-weighted cold start: random sparse graphs (chain fixed + random candidates, random sizes, densities,
weight spreads, fractional x, a few hubs and pendant-like vertices) -- lambda_2 of the weighted and the unweighted cold start against SciPy's
shift-invert value on the assembled Laplacian (n <= 6000; against each other beyond), residual rule, steps of both.  usage: fuzz_er.py [seeds] [seed0]"""
import sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, ".")
from mac_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0; tot_on = tot_off = 0
for s in range(seed0, seed0 + N):
    rng = np.random.default_rng(5000 + s)
    n = int(rng.choice([rng.integers(300, 3000), rng.integers(3000, 30000), rng.integers(30000, 120000)]))
    deg = float(rng.choice([1.0, 3.0, 8.0, 20.0]))
    m = int(n * deg / 2)
    a = rng.integers(0, n, m); b = rng.integers(0, n, m)
    if rng.random() < 0.3:      # a few hubs
        hubs = rng.integers(0, n, 3); a[: m // 20] = rng.choice(hubs, m // 20)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    spread = float(rng.choice([0.0, 1.0, 3.0]))
    cw = 10.0 ** rng.uniform(0, spread, len(ci))
    fw = 10.0 ** rng.uniform(0, spread, n - 1) * float(rng.choice([1.0, 0.05]))     # sometimes a weak chain
    x = rng.random(len(ci)); x[rng.random(len(ci)) < rng.choice([0.0, 0.5, 0.9])] = 0.0
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(x); P.set_solver(1)
    out = {}
    for tag, land in (("on", None), ("off", 0)):
        P.set_option("start_land", land)
        try:
            lam, v, _ = P.fiedler(tol=1e-8)
            out[tag] = (lam, int(P.stats.lanczos_steps), float(P.stats.residual))
        except Exception as e:      # noqa
            out[tag] = (float("nan"), -1, float("nan")); print("   seed", s, tag, type(e).__name__, e)
    ip, ix, da = P.laplacian_csr()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    lnorm = abs(L).sum(1).max()
    if n <= 6000:     # (a sparse LU of a random graph fills in: SciPy's shift-invert value only where it finishes at once; beyond, the two starts check each other)
        w = spla.eigsh(L + 1e-9 * lnorm * sp.identity(n), k=2, sigma=0, which="LM", return_eigenvectors=False)
        lam_ref = float(np.sort(w)[1]) - 1e-9 * lnorm
    else:
        lam_ref = out["off"][0]
    # the stop rule pins lambda_2 to ~ (1e-8 ||L||)^2 / gap absolutely; compare on the scale of ||L||_inf
    ok = all(np.isfinite(o[0]) and abs(o[0] - lam_ref) <= 1e-8 * lnorm and o[2] < 1e-8 for o in out.values())
    bad += (not ok); tot_on += max(0, out["on"][1]); tot_off += max(0, out["off"][1])
    print(f"{'ok ' if ok else 'BAD'} seed={s} n={n} deg={deg} spread={spread} nnz={L.nnz} lam_ref={lam_ref:.6e} on={out['on'][0]:.6e}/{out['on'][1]} off={out['off'][0]:.6e}/{out['off'][1]}", flush=True)
    P.close()
print("fuzz bad =", bad, " steps weighted", tot_on, "unweighted", tot_off)
