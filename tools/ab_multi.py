#!/usr/bin/env python3
"""Same-GPU A/B/C... of option settings of libmachip (mac_amd/csrc/options.h), one handle per setting (creation-time options
included: they are applied as process defaults around the handle's creation), alternating passes of 20 Frank-Wolfe iterations.
usage: ab_multi.py cfg rounds "panel=1,chunk=16" "asm_g=16" ...   ("-" = defaults; MACHIP_FOO=1 is read as foo=1)"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg, rounds = sys.argv[1], int(sys.argv[2])
def parse(a):
    out = {}
    for kv in a.split(","):
        k, v = kv.split("=")
        k = k.lower()
        out[k[7:] if k.startswith("machip_") else k] = int(v)
    return out
sets = [parse(a) if a != "-" else {} for a in sys.argv[3:]]
w = bench.make_workload(cfg)
Ps = []
for s in sets:
    with _lib.default_options(**s):
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    Ps.append(P)
res = [[] for _ in sets]; us = [[] for _ in sets]; lam = [None] * len(sets); steps = [0] * len(sets)
for r in range(rounds + 1):
    for i, s in enumerate(sets):
        P = Ps[i]
        P.set_x(w["x0"]); P.synchronize()
        t0 = time.perf_counter()
        rec = bench.run_pass(P, w["k"], 20, w["x0"])
        P.synchronize()
        el = time.perf_counter() - t0
        if r > 0:
            res[i].append(20 / el)
            us[i].append(1e3 * sum(x["step_ms"] for x in rec) / max(1, sum(x["steps_timed"] for x in rec)))
        lam[i] = [x["f"].hex() for x in rec]; steps[i] = sum(x["steps"] for x in rec)
for i, s in enumerate(sets):
    print(f"{cfg} {s or '-'}: {np.median(res[i]):.1f} it/s ({1e3/np.median(res[i]):.3f} ms/it), {np.median(us[i]):.2f} us/step, steps {steps[i]}, "
          f"lambda trajectory {'== first' if lam[i] == lam[0] else 'DIFFERS from first (max rel %.2e)' % max(abs(float.fromhex(a) - float.fromhex(b)) / abs(float.fromhex(b)) for a, b in zip(lam[i], lam[0]))}")
