#!/usr/bin/env python3
"""Budget sweep over a g2o pose graph -- the loop of the reference's experiment driver
(examples/g2o_experiment.py:240-336) without plotting / SE-Sync: for 10%..100% of the loop closures,
NaiveGreedy init -> MAC.solve(max_iters=20, rounding="nearest", use_cache=True) -> Madow rounding,
printing lambda_2 of each selection.  One MAC object (one device-resident problem) serves all budgets; by default the
budgets run CONCURRENTLY on the device (MAC.solve_sweep -> machip_fw_sweep), `concurrent=False` runs the reference's
one-after-the-other loop.

    python tools/g2o_sweep.py tests/golden/data/intel.g2o
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from mac_amd.solvers import MAC, NaiveGreedy          # noqa: E402
from mac_amd.utils.g2o import read_g2o_file, split_edges  # noqa: E402
from mac_amd.utils.rounding import round_madow        # noqa: E402


def sweep(path, pcts=(0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0), verbose=True, concurrent=True):
    edges, n = read_g2o_file(path)
    odom, lc = split_edges(edges)
    if verbose:
        print(f"{path}: {n} poses, {len(odom)} odometry edges, {len(lc)} loop closures")
    mac = MAC(odom, lc, n, fiedler_method="tracemin_cholesky")      # reference string, runs on HIP
    naive = NaiveGreedy(lc)
    if verbose:
        print(f"{'pct':>5} {'k':>6} {'naive':>12} {'unrounded':>12} {'nearest':>12} {'madow':>12} {'upper':>12} {'solve_s':>8}")
    rows = []
    ks = [int(pct * len(lc)) for pct in pcts]
    inits = [naive.subset(k) for k in ks]
    pre = None
    if concurrent:
        t0 = time.perf_counter()
        pre = mac.solve_sweep(ks, inits, max_iters=20, rounding="nearest", use_cache=True)
        dt_all = time.perf_counter() - t0
        if verbose:
            print(f"all {len(ks)} budgets solved concurrently in {dt_all:.3f} s")
    for j, pct in enumerate(pcts):
        k, w_init = ks[j], inits[j]
        t0 = time.perf_counter()
        if pre is not None:
            result, unrounded, upper = pre[j]
        else:
            result, unrounded, upper, rtime = mac.solve(k, w_init, max_iters=20, rounding="nearest",
                                                        return_rounding_time=True, use_cache=True)
        dt = time.perf_counter() - t0 if pre is None else dt_all / len(ks)
        madow = round_madow(unrounded, k, seed=np.random.RandomState(42)) if k < len(lc) else result
        # the four evaluations of a budget in one batched call (machip_eval_batch: concurrent evaluation lanes)
        lam = mac.evaluate_objective_batch(np.stack([w_init, unrounded, result, madow]))
        rows.append(dict(pct=pct, k=k, naive=lam[0], unrounded=lam[1], nearest=lam[2], madow=lam[3], upper=upper,
                         solve_s=dt, result=result, madow_x=madow))
        if verbose:
            print(f"{pct:5.1f} {k:6d} {lam[0]:12.8f} {lam[1]:12.8f} {lam[2]:12.8f} {lam[3]:12.8f} {upper:12.8f} {dt:8.3f}")
    return rows


if __name__ == "__main__":
    sweep(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/data/intel.g2o")
