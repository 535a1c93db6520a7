#!/usr/bin/env python3
"""Developer fuzz for the round-4 rules beyond the single-workgroup sizes (solver.h solve(): Lanczos -> exact hand-over on the
scheduler's forecast, history rule): chain-like graphs with 3 100 <= n <= 16 384 nodes and 0.13 n .. 3 072 active closures (denser than
the static rule of the preconditioned mode admits).  Per seed: a fresh handle in the automatic mode (first solve: hand-over possible;
second solve: history), the forced preconditioned mode, forced Lanczos (step cap 60 000), and SciPy's shift-invert Lanczos as the
independent value.  usage: fuzz_handover.py [seeds] [first seed]"""
import sys, time
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, ".")
from mac_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
handed = 0
for s in range(seed0, seed0 + N):
    rng = np.random.default_rng(7000 + s)
    n = int(rng.integers(3100, 16385))
    lo = int(0.13 * n) + 1
    if lo >= 3072:
        continue
    act = int(rng.integers(lo, 3073))
    m = int(act / 0.7) + 8
    fi = np.arange(n - 1, dtype=np.int32)
    fw = 10.0 ** rng.uniform(0, rng.choice([0.3, 1.5, 2.5]), n - 1)
    a = rng.integers(0, n, m); span = int(rng.choice([30, 400, n]))
    b = np.clip(a + rng.integers(-span, span + 1, m), 0, n - 1)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    key = np.unique(ci.astype(np.int64) * n + cj)
    ci, cj = (key // n).astype(np.int32), (key % n).astype(np.int32)
    cw = 10.0 ** rng.uniform(0, 2.0, len(ci))
    x = rng.random(len(ci)); x[rng.random(len(ci)) < 0.3] = 0.0
    res = {}
    for tag, mode in (("auto1", 0), ("auto2", 0), ("pcg", 2), ("lanczos", 1)):
        if tag in ("auto1", "pcg", "lanczos"):
            P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
            P.set_x(x)
            P.set_solver(mode)
        t0 = time.perf_counter()
        try:
            lam, v, _ = P.fiedler(max_steps=60000) if tag == "lanczos" else P.fiedler()
            res[tag] = (lam, int(P.stats.lanczos_steps), P.stats.residual, (time.perf_counter() - t0) * 1e3)
        except Exception as e:      # noqa
            res[tag] = (float("nan"), -1, float("nan"), 0.0); print("   seed", s, tag, type(e).__name__, str(e)[:100])
        if tag in ("auto2", "pcg", "lanczos"):
            if tag == "auto2":
                ip, ix, da = P.laplacian_csr()
            P.close()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    lnorm = abs(L).sum(axis=1).max()
    try:
        w = spla.eigsh(L + 1e-10 * lnorm * sp.identity(n), k=2, sigma=0, which="LM", return_eigenvectors=False, tol=1e-12)
        ref = float(np.sort(w)[1] - 1e-10 * lnorm)
    except Exception as e:      # noqa
        ref = float("nan"); print("   seed", s, "scipy", type(e).__name__)
    # the residual rule (1e-8 ||L||) pins lambda_2 to ~1e-8 ||L|| / gap-ish: compare to 1e-6 relative or 1e-9 ||L|| absolute
    tol = max(1e-6 * abs(ref), 1e-9 * lnorm)
    ok = all(np.isfinite(res[t][0]) and abs(res[t][0] - ref) <= tol and res[t][2] < 1e-8 for t in ("auto1", "auto2", "pcg"))
    lan_ok = np.isfinite(res["lanczos"][0]) and abs(res["lanczos"][0] - ref) <= tol
    h = res["lanczos"][1] > 0 and 128 < res["auto1"][1] < res["lanczos"][1] and res["auto1"][1] != res["pcg"][1]
    handed += bool(h and res["auto1"][1] < 0.9 * res["lanczos"][1])
    bad += (not ok)
    print(f"{'ok ' if ok else 'BAD'} seed={s} n={n} active={int((x > 1e-10).sum())} lam={res['auto1'][0]:.6e} ref={ref:.6e} steps auto1/auto2/pcg/lanczos="
          f"{res['auto1'][1]}/{res['auto2'][1]}/{res['pcg'][1]}/{res['lanczos'][1]}{'' if lan_ok else '(lanczos off)'} ms={res['auto1'][3]:.1f}/{res['auto2'][3]:.1f}/{res['pcg'][3]:.1f}/{res['lanczos'][3]:.1f}", flush=True)
print("fuzz_handover bad =", bad, " solves that handed over or went exact early:", handed)
