#!/usr/bin/env python3
"""Developer probe: MAC.solve(k = 10 % of the closures, NaiveGreedy init, 20 iterations, cold starts as in the
reference) on every g2o file given -- the number to set beside the reference's own timing."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mac_amd.solvers import MAC, NaiveGreedy
from mac_amd.utils.g2o import read_g2o_file, split_edges
for path in sys.argv[1:]:
    edges, n = read_g2o_file(path)
    odom, lc = split_edges(edges)
    k = int(0.1 * len(lc))
    mac = MAC(odom, lc, n)
    w0 = NaiveGreedy(lc).subset(k)
    mac.solve(k, w0, max_iters=20)            # warm-up (graphs, allocations)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); r, u, up = mac.solve(k, w0, max_iters=20); best = min(best, time.perf_counter() - t0)
    print(f"{path.split('/')[-1]} n={n} lc={len(lc)} k={k}: MAC.solve(20 iters) {best:.4f} s  (lambda_2 of the result {mac.evaluate_objective(r):.8g}, {len(mac.trace)} iterations run)", flush=True)
