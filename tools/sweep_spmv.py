"""Sweep launch shapes of the fused Lanczos-step kernel on BASELINE configs (run on the GPU box).
usage: python tools/sweep_spmv.py [c2|c4]   -- spawns one subprocess per env setting."""
import json, os, subprocess, sys
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
child = r'''
import sys, os, json; sys.path.insert(0, ".")
import numpy as np
import bench
from mac_amd import _lib
w = bench.make_workload("%s")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
out = {}
for nm, x in [("x0", w["x0"]), ("half", np.where(np.arange(len(w["cw"])) %% 2 == 0, 1.0, 0.0)), ("ones", np.ones(len(w["cw"])))]:
    P.set_x(x); nnz = P.assemble()
    us, by = P.profile_spmv(300)
    out[nm] = (nnz, round(us, 2), round(by / us / 1e3, 1))
print(json.dumps(out))
''' % cfg
settings = [{}]
for cap in (128, 256, 512, 1024):
    for var in ("stream", "vec"):
        settings.append({"MACHIP_MAXGRID": str(cap), "MACHIP_SPMV": var})
for var, key, vals in (("stream", "MACHIP_TPR", (2, 4, 8, 16)), ("vec", "MACHIP_G", (4, 8, 16, 32, 64))):
    for v in vals:
        settings.append({"MACHIP_SPMV": var, key: str(v)})
for s in settings:
    env = dict(os.environ); env.update(s)
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    print(json.dumps(s), r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
