#!/usr/bin/env python3
"""Aggregate Frank-Wolfe throughput of a concurrent budget sweep (MAC.solve_sweep -> machip_fw_sweep) against the
one-at-a-time loop on the same budgets.  usage: sweep_probe.py [intel|sphere2500|city10000] [budgets] [iters]"""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from conftest import load_golden
from mac_amd.solvers import MAC, NaiveGreedy
from mac_amd.utils.graphs import Edge
nm = sys.argv[1] if len(sys.argv) > 1 else "intel"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 10
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
g = load_golden("g2o_" + nm)
ed = lambda pre: [Edge(int(a), int(b), float(w)) for a, b, w in zip(g[pre + "i"], g[pre + "j"], g[pre + "w"])]
fixed, cand, n = ed("f"), ed("c"), int(g["n"])
m = len(cand)
ks = [int((0.1 + 0.8 * j / max(1, B - 1)) * m) for j in range(B)]
naive = NaiveGreedy(cand)
inits = [naive.subset(k) for k in ks]
mac = MAC(fixed, cand, n)
mac.solve(ks[0], inits[0], max_iters=3)
t0 = time.perf_counter(); its = 0
for k, x0 in zip(ks, inits):
    mac.solve(k, x0, max_iters=iters, relative_duality_gap_tol=0.0, grad_norm_tol=0.0); its += len(mac.trace)
t_seq = time.perf_counter() - t0
for lanes in (1, 2, 4, 8, 12, 16):
    os.environ["MACHIP_LANES"] = str(lanes)
    mac2 = MAC(fixed, cand, n)
    mac2.solve_sweep(ks[:lanes], inits[:lanes], max_iters=2)
    t0 = time.perf_counter()
    mac2.solve_sweep(ks, inits, max_iters=iters, relative_duality_gap_tol=0.0, grad_norm_tol=0.0)
    t = time.perf_counter() - t0
    print(f"{nm}: {B} budgets x {iters} it: one at a time {its / t_seq:8.1f} it/s | {lanes:2d} lanes {B * iters / t:8.1f} it/s  ({t_seq / t:4.2f}x)", flush=True)
