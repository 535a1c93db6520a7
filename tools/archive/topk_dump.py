#!/usr/bin/env python3
"""Developer probe: dump the HIP path's LP vertices s_0, s_1 (top-K index sets) of the C2 run to
gpurun_out/c2_topk.npz so they can be compared offline with an exact dense eigen-solve."""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from mac_amd import _lib
from test_gpu_parity import make_er, reference_start_block
n = 10000
ci, cj = make_er(n, 0.01, 0)
m = len(ci); k = m // 10
fi = np.arange(n - 1, dtype=np.int32)
P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(m))
P.set_start(reference_start_block(n)[:, 0].copy())
x0 = np.zeros(m); x0[np.random.default_rng(0).choice(m, k, replace=False)] = 1.0
P.set_x(x0)
out = {}
for it in range(2):
    P.assemble(); lam, v, _ = P.fiedler()
    g = P.gradient(); s = P.lp_topk(k)
    out[f"s{it}"] = np.nonzero(s)[0]; out[f"lam{it}"] = lam; out[f"v{it}"] = v
    x = P.get_x(); P.set_x(x + 2.0 / (it + 2) * (s - x))
np.savez_compressed("gpurun_out/c2_topk.npz", **out)
print({k_: (v_.shape if hasattr(v_, "shape") else v_) for k_, v_ in out.items()})
