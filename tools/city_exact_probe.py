#!/usr/bin/env python3
"""developer tool: the first <iters> Frank-Wolfe iterations of city10000 (one pass, no warm-up) -- run under tools/kstats.sh with
MACHIP_SOLVER / MACHIP_WB_MAX set to see the kernels of the exact chain + closures mode at a few thousand closures."""
import sys
sys.path.insert(0, ".")
import bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w = bench.make_workload(sys.argv[2] if len(sys.argv) > 2 else "c5b")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
rec = bench.run_pass(P, w["k"], iters, w["x0"])
print([(r["steps"], round(r["gpu_ms"], 3)) for r in rec])
