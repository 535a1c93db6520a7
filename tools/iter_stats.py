"""Per-FW-iteration statistics (steps, nnz, solve ms) cold vs warm start. usage: iter_stats.py [cfg] [iters]"""
import sys, time; sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
for warm in (False, True):
    P.set_x(w["x0"]); P.synchronize()
    t0 = time.perf_counter(); tot_steps = 0; rows = []
    for it in range(iters):
        t1 = time.perf_counter()
        f, d, g = P.fw_step(w["k"], it, warm_start=warm and it > 0)
        P.fw_commit()
        t2 = time.perf_counter()
        st = P.stats
        rows.append((it, f, int(st.lanczos_steps), int(st.nnz), st.gpu_ms, 1e3 * (t2 - t1)))
        tot_steps += int(st.lanczos_steps)
    P.synchronize(); el = time.perf_counter() - t0
    print(f"== {cfg} warm={warm}: {iters/el:.1f} it/s, steps/iter {tot_steps/iters:.1f}")
    for r in rows:
        print("   it %2d f=%.10g steps=%4d nnz=%8d eig_ms=%.3f total_ms=%.3f  us/step=%.2f" % (*r, 1e3 * r[4] / max(1, r[2])))
