#!/usr/bin/env python3
"""Per Frank-Wolfe iteration: eigen-solver steps / iterations and device ms, for the solver modes given (MACHIP_SOLVER values).
usage: iter_compare.py <config> <iters> mode [mode ...]   (modes may carry env settings: jacobi:MACHIP_PANEL=1)"""
import os, subprocess, sys, json
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, "."); import bench
    from mac_amd import _lib
    from mac_amd.utils.fiedler import reference_start_block
    cfg, iters = sys.argv[2], int(sys.argv[3])
    w = bench.make_workload(cfg)
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    bench.run_pass(P, w["k"], 2, w["x0"])
    rec = bench.run_pass(P, w["k"], iters, w["x0"])
    rec = bench.run_pass(P, w["k"], iters, w["x0"])
    print("REC", json.dumps([(r["steps"], r["gpu_ms"], r["nnz"], r["f"]) for r in rec]))
    sys.exit(0)
cfg, iters, modes = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {}
for m in modes:
    env = dict(os.environ)
    parts = m.split(":")
    env["MACHIP_SOLVER"] = parts[0]
    for kv in parts[1:]:
        k, v = kv.split("="); env[k] = v
    out = subprocess.run([sys.executable, __file__, "--child", cfg, str(iters)], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("REC")]
    if not line:
        print(m, "FAILED", out.stderr[-800:]); continue
    res[m] = json.loads(line[0][4:])
print("it      nnz  " + "  ".join(f"{m[:28]:>28s}" for m in res))
for i in range(iters):
    row = f"{i:2d} {res[next(iter(res))][i][2]:8d}  "
    for m in res:
        s, ms, _, f = res[m][i]
        row += f"{s:6d} st {ms:7.3f} ms {1e3*ms/max(1,s):6.1f} us  "
    print(row)
print("sum          " + "  ".join(f"{sum(r[0] for r in res[m]):6d} st {sum(r[1] for r in res[m]):7.3f} ms              " for m in res))
