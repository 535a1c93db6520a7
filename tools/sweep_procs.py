#!/usr/bin/env python3
"""Is the concurrent budget sweep bound by the host side (HIP runtime calls of one process) or by the GPU?  Runs the same
sweep in P processes at once, L lanes each, and reports the aggregate.  usage: sweep_procs.py graph procs lanes [budgets] [iters]"""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, "."); sys.path.insert(0, "tests")
    import numpy as np
    from conftest import load_golden
    from mac_amd.solvers import MAC, NaiveGreedy
    from mac_amd.utils.graphs import Edge
    nm, lanes, B, iters, go = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    if lanes > 0: os.environ["MACHIP_LANES"] = str(lanes)      # 0: the library's own choice
    g = load_golden("g2o_" + nm)
    ed = lambda pre: [Edge(int(a), int(b), float(w)) for a, b, w in zip(g[pre + "i"], g[pre + "j"], g[pre + "w"])]
    fixed, cand, n = ed("f"), ed("c"), int(g["n"])
    m = len(cand)
    ks = [int((0.1 + 0.8 * j / max(1, B - 1)) * m) for j in range(B)]
    inits = [NaiveGreedy(cand).subset(k) for k in ks]
    mac = MAC(fixed, cand, n)
    mac.solve_sweep(ks if lanes == 0 else ks[:lanes], inits if lanes == 0 else inits[:lanes], max_iters=2)
    open(go + f".ready{os.getpid()}", "w").close()
    while not os.path.exists(go): time.sleep(0.001)
    t0 = time.perf_counter()
    mac.solve_sweep(ks, inits, max_iters=iters, relative_duality_gap_tol=0.0, grad_norm_tol=0.0)
    print("CHILD", B * iters, time.perf_counter() - t0, flush=True)
    sys.exit(0)
nm, P, lanes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 12
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
go = f"/tmp/sweep_go_{os.getpid()}"
ch = [subprocess.Popen([sys.executable, __file__, "--child", nm, str(lanes), str(B), str(iters), go], stdout=subprocess.PIPE, text=True) for _ in range(P)]
import glob
while len(glob.glob(go + ".ready*")) < P: time.sleep(0.01)
open(go, "w").close()
tot = 0; tmax = 0.0
for c in ch:
    out = c.communicate()[0]
    for l in out.splitlines():
        if l.startswith("CHILD"):
            _, its, t = l.split(); tot += int(its); tmax = max(tmax, float(t))
print(f"{nm}: {P} processes x {lanes} lanes, {B} budgets x {iters} it each: {tot / tmax:8.1f} it/s aggregate")
