#!/usr/bin/env python3
"""Developer probe: one fuzz seed (tools/fuzz_modes.py), modes in a given order, with details."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mac_amd import _lib
s = int(sys.argv[1]); order = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "120")]
rng = np.random.default_rng(1000 + s)
n = int(rng.choice([rng.integers(260, 3072), rng.integers(3072, 16384), rng.integers(16384, 60000)]))
ncl = int(rng.integers(1, max(2, int(n * rng.choice([0.005, 0.05, 0.3])))))
fi = np.arange(n - 1, dtype=np.int32)
fw = 10.0 ** rng.uniform(0, rng.choice([0.5, 2, 3]), n - 1)
a = rng.integers(0, n, ncl); span = int(rng.choice([50, 3000, n]))
b = np.clip(a + rng.integers(-span, span + 1, ncl), 0, n - 1)
keep = np.abs(a - b) > 1
ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
cw = 10.0 ** rng.uniform(0, 2.5, len(ci))
x = rng.random(len(ci)); x[rng.random(len(ci)) < 0.3] = 0.0
P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
P.set_x(x)
ref = None
for mode in order:
    P.set_solver(mode)
    t0 = time.perf_counter()
    try:
        lam, v, _ = P.fiedler()
        st = P.stats
        if ref is None: ref = lam
        print(f"mode {mode}: lam={lam:.12e} dlam/lam={abs(lam-ref)/ref:.1e} dlam/lnorm={abs(lam-ref)/st.lnorm:.1e} steps={st.lanczos_steps} restarts={st.restarts} res={st.residual:.2e} lnorm={st.lnorm:.3g} ms={1e3*(time.perf_counter()-t0):.1f}", flush=True)
    except Exception as e:      # noqa
        print(f"mode {mode}: {type(e).__name__} steps={P.stats.lanczos_steps} res={P.stats.residual:.2e}", flush=True)
