cd /root/repo
python - <<'PY' 2> gpurun_out/r5_p_dbg.txt
import sys; sys.path.insert(0,'.')
import bench, numpy as np
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w = bench.make_workload("c4")
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"]); rec = bench.run_pass(P, w["k"], 6, w["x0"])
P.set_option("debug", 1)
P.set_x(w["x0"]); rec = bench.run_pass(P, w["k"], 6, w["x0"])
PY
grep -c . gpurun_out/r5_p_dbg.txt; tail -120 gpurun_out/r5_p_dbg.txt | cut -c1-200
