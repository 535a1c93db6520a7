#!/usr/bin/env python3
"""Gather step vs column-panel step (in-solve us per step) on random graphs of other sizes than the bench's.
usage: panel_size_probe.py n degree [n degree ...]"""
import os, sys
sys.path.insert(0, ".")
import numpy as np
from mac_amd import _lib
args = [int(t) for t in sys.argv[1:]] or [200000, 30]
for n, deg in zip(args[::2], args[1::2]):
    rng = np.random.default_rng(5)
    m0 = n * deg // 2
    a = rng.integers(0, n, m0); b = rng.integers(0, n, m0)
    keep = (a != b) & (np.abs(a - b) != 1)
    key = np.unique(np.minimum(a[keep], b[keep]).astype(np.int64) * n + np.maximum(a[keep], b[keep]))
    ci, cj = (key // n).astype(np.int32), (key % n).astype(np.int32)
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(len(ci)))
    P.set_x(np.ones(len(ci)))
    row = []
    for mode in ("0", "1"):
        os.environ["MACHIP_PANEL"] = mode
        P.assemble(); lam, _, _ = P.fiedler(want_vec=False)
        lam, _, _ = P.fiedler(want_vec=False)
        st = P.stats
        row.append((1e3 * st.step_ms / max(1, st.steps_timed), int(st.lanczos_steps), lam, float(st.gpu_ms)))
    os.environ.pop("MACHIP_PANEL", None)
    print(f"n={n} nnz={int(st.nnz)} ({st.nnz / n:.1f}/row): gather {row[0][0]:.2f} us/step ({row[0][1]} steps, {row[0][3]:.2f} ms)  panel {row[1][0]:.2f} us/step ({row[1][1]} steps, {row[1][3]:.2f} ms)  dlam {abs(row[0][2] - row[1][2]) / row[0][2]:.1e}", flush=True)
    P.close()
