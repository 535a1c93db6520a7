#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5l
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $out/pytest.log 2>&1; tail -22 $out/pytest.log
timeout 600 python tools/ab_multi.py c4 5 "-" > $out/ab_c4.txt 2>&1; cat $out/ab_c4.txt
timeout 300 python tools/ab_multi.py c2 5 "-" > $out/ab_c2.txt 2>&1; cat $out/ab_c2.txt
timeout 300 python tools/ab_multi.py c5b 5 "-" > $out/ab_c5b.txt 2>&1; cat $out/ab_c5b.txt
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --warmup 0 --no-cpu --no-pmc --no-warm --min-seconds 0 --max-repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
f=$(find $out/trace -name "t_kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/trace
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
for r in rows[:34]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e3/20:9.1f} us/iter  avg {float(r['AverageNs'])/1e3:8.2f}")
PY
