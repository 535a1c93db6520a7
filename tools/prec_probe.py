#!/usr/bin/env python3
"""fp64 against the mixed mode (machip_set_precision(1)) on the BASELINE graphs: steps, fp32 share, step time,
solve time of cold eigen-solves along a Frank-Wolfe run.  usage: prec_probe.py [cfg ...]"""
import os, sys
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
for cfg in (sys.argv[1:] or ["c3", "c5a", "c5b", "c2"]):
    w = bench.make_workload(cfg)
    res = {}
    for prec in (0, 1):
        print(f"-- {cfg} precision {prec}", flush=True)
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
        P.set_precision(prec)
        P.set_start(reference_start_block(w["n"])[:, 0].copy())
        bench.run_pass(P, w["k"], 3, w["x0"])
        import time
        P.set_x(w["x0"]); P.synchronize()
        t0 = time.perf_counter()
        if os.environ.get("PROBE_DEBUG_IT") and prec == 1:       # trace one eigen-solve of the timed mixed pass
            dbg = int(os.environ["PROBE_DEBUG_IT"])
            rec = bench.run_pass(P, w["k"], dbg, w["x0"])
            os.environ["MACHIP_DEBUG"] = "1"
            f, d, g = P.fw_step(w["k"], dbg); print("traced:", f, P.stats.asdict(), flush=True)
            os.environ.pop("MACHIP_DEBUG")
            sys.exit(0)
        rec = bench.run_pass(P, w["k"], 20, w["x0"])
        P.synchronize()
        el = time.perf_counter() - t0
        res[prec] = (rec, el)
        P.close()
    r0, r1 = res[0][0], res[1][0]
    print(f"== {cfg}: fp64 {20 / res[0][1]:.1f} it/s, mixed {20 / res[1][1]:.1f} it/s; lambda2 rel diff, iterations 0-2 "
          f"(same x on both sides): {max(abs(a['f'] - b['f']) / abs(a['f']) for a, b in zip(r0[:3], r1[:3])):.2e}, all 20 "
          f"(the trajectories may fork at near-ties): {max(abs(a['f'] - b['f']) / abs(a['f']) for a, b in zip(r0, r1)):.2e}")
    for it, (a, b) in enumerate(zip(r0, r1)):
        print(f"   it {it:2d} fp64: steps {a['steps']:5d} {a['gpu_ms']:7.3f} ms {1e3 * a['step_ms'] / max(1, a['steps_timed']):6.2f} us/step | "
              f"mixed: steps {b['steps']:5d} (fp32 {b.get('steps_lowp', -1):5d}) {b['gpu_ms']:7.3f} ms {1e3 * b['step_ms'] / max(1, b['steps_timed']):6.2f} us/step")
