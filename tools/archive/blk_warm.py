"""developer probe (needs the experiments build: MACHIP_BUILD_FLAGS=-DMACHIP_EXPERIMENTS MACHIP_BUILD_OUT=libmachip_exp.so bash mac_amd/csrc/build.sh; MACHIP_LIB=mac_amd/libmachip_exp.so): Frank-Wolfe passes with use_cache=True semantics (warm start), scalar vs block Lanczos"""
import sys, time; sys.path.insert(0, '.')
import bench, numpy as np
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1] if len(sys.argv) > 1 else "c5b"
w = bench.make_workload(cfg)
for setting in ({"blocklan": 0}, {"blocklan": 1}):
    with _lib.default_options(**setting):
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    best = 0; steps = None
    for rep in range(4):
        P.set_x(w["x0"]); P.synchronize()
        t0 = time.perf_counter()
        r = P.fw_run(w["k"], 20, warm_start=True)
        P.synchronize()
        el = time.perf_counter() - t0
        best = max(best, 20 / el)
        steps = [int(s.lanczos_steps) for s in r["stats"][:20]]
    print(cfg, setting, "warm: %.1f it/s" % best, "steps", steps, "lambda2 last %.12g" % r["f"][19], "modes", list(r["modes"][:20]))
