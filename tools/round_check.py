#!/usr/bin/env python3
"""Developer probe: determinism of the C2 trajectory and of the device round_nearest on its end point."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mac_amd import _lib
import oracle
from test_gpu_parity import make_er, reference_start_block
n = 10000
ci, cj = make_er(n, 0.01, 0)
m = len(ci); k = m // 10
fi = np.arange(n - 1, dtype=np.int32)
def run():
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(m))
    P.set_start(reference_start_block(n)[:, 0].copy())
    x0 = np.zeros(m); x0[np.random.default_rng(0).choice(m, k, replace=False)] = 1.0
    P.set_x(x0)
    steps = []
    for it in range(20):
        P.fw_step(k, it); steps.append(int(P.stats.lanczos_steps)); P.fw_commit()
    return P, steps
for rep in range(2):
    P, steps = run()
    w = P.get_x()
    print("run", rep, "x hash", hashlib.sha1(w.tobytes()).hexdigest()[:12], "steps", sum(steps), flush=True)
    sums = []
    for i in range(40):
        r = P.round_nearest(k, decimals=10); sums.append(int(r.sum()))
    ro = oracle.round_nearest(w, k, np.ones(m), 10)
    print("   device sums", sorted(set(sums)), "oracle sum", int(ro.sum()), "equal to oracle:", bool(np.array_equal(r, ro)), flush=True)
    P.close()
