#!/usr/bin/env python3
"""developer tool: one seed of tools/fuzz_handover.py under MACHIP_DEBUG, forced Lanczos (where do the slow steps come from?)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from mac_amd import _lib
s = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(7000 + s)
n = int(rng.integers(3100, 16385))
lo = int(0.13 * n) + 1
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
P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
P.set_x(x); P.set_solver(mode)
ip, ix, da = P.laplacian_csr()
print("n", n, "span", span, "longest row", int(np.diff(ip).max()), "nnz", len(ix), file=sys.stderr)
for rep in range(2):
    t0 = time.perf_counter()
    lam, v, _ = P.fiedler()
    print("rep", rep, "lam", lam, "steps", P.stats.lanczos_steps, "restarts", P.stats.restarts, "gpu_ms", P.stats.gpu_ms, "step_ms", P.stats.step_ms, "steps_timed", P.stats.steps_timed, "wall ms", (time.perf_counter() - t0) * 1e3, file=sys.stderr)
