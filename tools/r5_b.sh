#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5g
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q >> $out/pytest.log; tail -5 $out/pytest.log
timeout 600 python tools/ab_multi.py c4 5 "-" "sel_fuse=0" > $out/ab_c4.txt 2>&1; cat $out/ab_c4.txt; timeout 300 python tools/ab_multi.py c2 5 "-" > $out/ab_c2.txt 2>&1; cat $out/ab_c2.txt
true
MACHIP_DEBUG=1 timeout 300 python - > $out/debug_c4.txt 2>&1 <<PY
import sys; sys.path.insert(0,'.')
import bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
w=bench.make_workload('c4')
P=_lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
bench.run_pass(P, w["k"], 20, w["x0"])
PY
grep -B1 "check J=" $out/debug_c4.txt | grep -v "^--" | sed 's/\[machip\]//' | cut -c1-160 > $out/checks_c4.txt; wc -l $out/checks_c4.txt; rm $out/debug_c4.txt
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --warmup 0 --no-cpu --no-pmc --no-warm --min-seconds 0 --max-repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
f=$(find $out/trace -name "t_kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/trace
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:40]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3/20:9.1f} us/iter  avg {float(r['AverageNs'])/1e3:8.2f}")
PY
