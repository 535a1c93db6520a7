#!/bin/bash
# developer tool: per-kernel times of the preconditioned eigen-solver mode (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lobprof
MACHIP_SOLVER=lobpcg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lobprof -o lob -- python $GRAFT_REPO_ROOT/tools/lob_probe.py > /tmp/lobprof.log 2>&1
f=$(find /tmp/lobprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:8.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
