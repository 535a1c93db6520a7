#!/usr/bin/env python3
"""Row-partitioned eigen-solve (in-process communicator) against a single handle, value by value.  usage: shard_probe.py cfg R iters"""
import os, sys, threading
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg, R, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
os.environ["MACHIP_PANEL"] = "0"
w = bench.make_workload(cfg)
n, k = w["n"], w["k"]
start = reference_start_block(n)[:, 0].copy()
mk = lambda: _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
def drive(P, out, i):
    P.set_start(start); P.set_x(w["x0"])
    fs = []
    for it in range(iters):
        fs.append(P.fw_step(k, it) + (int(P.stats.lanczos_steps), float(P.stats.residual))); P.fw_commit()
    out[i] = (np.array(fs), P.get_x())
single = [None]; P0 = mk(); drive(P0, single, 0)
again = [None]; drive(P0, again, 0); P0.close()
print("single run twice identical:", np.array_equal(single[0][0], again[0][0]))
Ps = [mk() for _ in range(R)]
_lib.comm_init_local(Ps)
out = [None] * R
th = [threading.Thread(target=drive, args=(Ps[r], out, r)) for r in range(R)]
[t.start() for t in th]; [t.join() for t in th]
for r in range(R):
    a, b = out[r][0], single[0][0]
    print(f"rank {r}: steps {a[:, 3].astype(int).tolist()} vs {b[:, 3].astype(int).tolist()}  max|df| {np.abs(a[:, 0] - b[:, 0]).max():.3e} max|ddual| {np.abs(a[:, 1] - b[:, 1]).max():.3e} x equal {np.array_equal(out[r][1], single[0][1])} res {a[:,4]} {b[:,4]}")
