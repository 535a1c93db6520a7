#!/usr/bin/env python3
"""Developer probe (GPU box): the HIP path's LP vertices on city10000, iteration by iteration, next to the
reference's (tests/golden/city10000_vertices.npz) -> gpurun_out/city_topk.npz + a printed comparison."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
g = np.load("tests/golden/g2o_city10000.npz"); gv = np.load("tests/golden/city10000_vertices.npz")
n, k = int(g["n"]), int(g["k"])
tol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-8
P = _lib.Problem(n, g["fi"].astype(np.int32), g["fj"].astype(np.int32), g["fw"], g["ci"].astype(np.int32), g["cj"].astype(np.int32), g["cw"])
P.set_start(reference_start_block(n)[:, 0].copy())
P.set_x(g["x_init"])
out = {}
for it in range(20):
    f, dual, gn = P.fw_step(k, it, tol=tol)
    s = np.nonzero(P.lp_topk(k))[0]      # g of this iteration is still resident
    ref = gv["ref_s"][it]
    same = s is not None and np.array_equal(s, ref)
    print(it, f"{f:.12g} ref {gv['f_traj'][it]:.12g} rel {abs(f - gv['f_traj'][it]) / f:.2e} steps {P.stats.lanczos_steps} res {P.stats.residual:.1e} vertex==ref {same}"
          + ("" if same or s is None else f" (differs in {len(np.setdiff1d(s, ref))})"), flush=True)
    out[f"s{it}"] = s; out[f"f{it}"] = f
    P.fw_commit()
np.savez_compressed("gpurun_out/city_topk.npz", **out)
