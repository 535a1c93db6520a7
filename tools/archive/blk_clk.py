"""developer probe (experiments build, MACHIP_LIB=mac_amd/libmachip_exp.so): phase clocks of the block Lanczos step (option debug = 2)"""
import sys; sys.path.insert(0, '.')
import bench, numpy as np
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w = bench.make_workload("c5b")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_option("blocklan", 1); P.set_option("debug", 2)
P.set_x(w["x0"]); rec = bench.run_pass(P, w["k"], 5, w["x0"])
