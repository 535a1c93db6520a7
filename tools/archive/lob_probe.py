#!/usr/bin/env python3
"""Developer probe: preconditioned (LOBPCG + tridiagonal chain solve) vs Lanczos eigen-solver mode on
the pose graphs under tests/golden/data and on a synthetic stiff chain.  One cold solve each."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mac_amd import _lib
from mac_amd.utils.g2o import read_g2o_file, split_edges

def problem_from_g2o(path, pct):
    edges, n = read_g2o_file(path)
    odom, lc = split_edges(edges)
    f = lambda es: (np.array([e.i for e in es], np.int32), np.array([e.j for e in es], np.int32), np.array([e.weight for e in es]))
    fi, fj, fw = f(odom); ci, cj, cw = f(lc)
    P = _lib.Problem(n, fi, fj, fw, ci, cj, cw)
    k = int(pct * len(lc)); x = np.zeros(len(lc)); x[np.argsort(-cw, kind="stable")[:k]] = 1.0
    P.set_x(x)
    return P, n, k

def synthetic(n, nc):
    rng = np.random.default_rng(1)
    fi = np.arange(n - 1, dtype=np.int32); fw = rng.uniform(100, 1000, n - 1)
    a = rng.integers(0, n, nc); b = np.clip(a + rng.integers(-3000, 3000, nc), 0, n - 1)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, rng.uniform(100, 300, len(ci)))
    P.set_x(np.ones(len(ci)))
    return P, n, len(ci)

def run(name, P, n, k):
    P.assemble()
    rs = np.random.RandomState(7).normal(size=(min(4, n - 1), n)).T[:, 0].copy()
    P.set_start(rs)
    out = []
    for mode in ("lanczos", "lobpcg"):
        os.environ["MACHIP_SOLVER"] = mode
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            try:
                lam, v, _ = P.fiedler()
            except Exception as e:      # noqa
                lam, v = float("nan"), None; print("   ", mode, "failed:", e)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        s = P.stats
        out.append((mode, lam, s.lanczos_steps, s.restarts, best * 1e3, s.residual, v))
    (m1, l1, s1, r1, t1, e1, v1), (m2, l2, s2, r2, t2, e2, v2) = out
    dv = min(np.abs(v1 - v2).max(), np.abs(v1 + v2).max()) if v1 is not None and v2 is not None else float("nan")
    print(f"{name:28s} n={n:6d} k={k:5d} | lanczos {s1:6d} steps {t1:8.2f} ms res {e1:.1e} | lobpcg {s2:5d} its {r2} rst {t2:8.2f} ms res {e2:.1e} "
          f"| dlam/lam {abs(l1-l2)/abs(l1):.1e} dv {dv:.1e} speedup {t1/t2:.2f}", flush=True)

if __name__ == "__main__":
    d = os.path.join(ROOT, "tests/golden/data") + "/"
    cases = [("intel", .2), ("intel", 1.0), ("kitti_05", .2), ("kitti_05", 1.0), ("sphere2500", .2), ("city10000", .2), ("city10000", 1.0)]
    if os.environ.get("LOB_PROBE_SHORT"): cases = [("city10000", .2)]
    for nm, pct in cases:
        P, n, k = problem_from_g2o(d + nm + ".g2o", pct); run(f"{nm} {int(pct*100)}%", P, n, k); P.close()
    for n, nc in ([] if os.environ.get("LOB_PROBE_SHORT") else [(4661, 43), (15000, 320), (15000, 1600)]):
        P, n, k = synthetic(n, nc); run(f"synthetic chain {nc} closures", P, n, k); P.close()
