"""developer probe (needs the experiments build: MACHIP_BUILD_FLAGS=-DMACHIP_EXPERIMENTS MACHIP_BUILD_OUT=libmachip_exp.so bash mac_amd/csrc/build.sh; MACHIP_LIB=mac_amd/libmachip_exp.so): a few Frank-Wolfe iterations of city10000 with the block Lanczos mode forced (eager launches, for kernel traces)"""
import sys; sys.path.insert(0, '.')
import bench, numpy as np
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c5b")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_option("blocklan", 1); P.set_option("graph", int(sys.argv[2]) if len(sys.argv) > 2 else 0)
P.set_x(w["x0"]); rec = bench.run_pass(P, w["k"], 8, w["x0"])
print([(r["steps"], round(1e3 * r["step_ms"] / max(1, r["steps_timed"]), 2)) for r in rec])
