import sys; sys.path.insert(0,'.')
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w=bench.make_workload("c2")
P=_lib.Problem(w["n"],w["fi"],w["fj"],w["fw"],w["ci"],w["cj"],w["cw"])
P.set_start(reference_start_block(w["n"])[:,0].copy())
P.set_x(w["x0"])
for it in range(3):
    print("=== FW it",it, flush=True)
    f,d,g=P.fw_step(w["k"],it); P.fw_commit()
    print(f, P.stats.lanczos_steps, P.stats.restarts, P.stats.gpu_ms, flush=True)
