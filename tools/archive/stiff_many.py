#!/usr/bin/env python3
"""Developer probe: stiff chains (weights over three decades) with MANY active closures (beyond the first Woodbury tier).
usage: stiff_many.py n n_closures [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mac_amd import _lib
n = int(sys.argv[1]); nc = int(sys.argv[2]); seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
fi = np.arange(n - 1, dtype=np.int32); fw = 10.0 ** rng.uniform(0, 3, n - 1)
a = rng.integers(0, n, nc); span = int(sys.argv[4]) if len(sys.argv) > 4 else n
b = np.clip(a + rng.integers(-span, span + 1, nc), 0, n - 1)
keep = np.abs(a - b) > 1
ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
cw = 10.0 ** rng.uniform(0, 2.5, len(ci))
x = rng.uniform(0.2, 1.0, len(ci))
P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
P.set_x(x)
for rep in range(2):
    t0 = time.perf_counter()
    try:
        lam, v, _ = P.fiedler()
        st = P.stats
        print(f"n={n} closures={len(ci)} span={span}: lam={lam:.6e} lam/lnorm={lam / st.lnorm:.1e} steps={st.lanczos_steps} res={st.residual:.2e} ms={1e3 * (time.perf_counter() - t0):.1f}", flush=True)
    except Exception as e:
        print(f"n={n} closures={len(ci)} span={span}: {type(e).__name__} steps={P.stats.lanczos_steps} res={P.stats.residual:.2e} ms={1e3 * (time.perf_counter() - t0):.1f}", flush=True)
