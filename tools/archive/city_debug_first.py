#!/usr/bin/env python3
"""developer tool: MACHIP_DEBUG trace (chunk by chunk: steps done, residual estimate, estimated steps to go) of the first eigen-solves of a
pose-graph workload -- how early does the Lanczos scheduler know that a solve will be long?   usage: city_debug_first.py [config] [iters]"""
import os, sys
sys.path.insert(0, ".")
os.environ["MACHIP_DEBUG"] = "1"
import bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "c5b")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
bench.run_pass(P, w["k"], int(sys.argv[2]) if len(sys.argv) > 2 else 2, w["x0"])
