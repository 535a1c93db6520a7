#!/usr/bin/env python3
"""Throughput of machip_eval_batch against one-at-a-time evaluate_objective. usage: batch_probe.py [cfg ...]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
for cfg in (sys.argv[1:] or ["c3", "c5a", "c5b"]):
    w = bench.make_workload(cfg)
    m, k = len(w["cw"]), w["k"]
    rs = np.random.RandomState(1)
    B = 64
    X = np.zeros((B, m))
    for b in range(B):
        X[b, rs.choice(m, k, replace=False)] = 1.0
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    P.eval_batch(X[:8])                       # create the lanes, warm up
    for x in X[:4]:
        P.set_x(x); P.fiedler(want_vec=False)
    t0 = time.perf_counter()
    single = []
    for x in X:
        P.set_x(x); single.append(P.fiedler(want_vec=False)[0])
    t1 = time.perf_counter()
    out = {}
    for lanes in (1, 2, 4, 8, 12):
        os.environ["MACHIP_LANES"] = str(lanes)
        t2 = time.perf_counter()
        lam, st = P.eval_batch(X)
        out[lanes] = B / (time.perf_counter() - t2)
        assert np.array_equal(lam, np.array(single)), "batch differs from single evaluations"
    print(f"== {cfg} (n={w['n']}, m={m}, K={k}): one at a time {B / (t1 - t0):.0f} solves/s; eval_batch " +
          ", ".join(f"{l} lanes {v:.0f}/s" for l, v in out.items()), flush=True)
    P.close()
