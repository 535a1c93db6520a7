#!/bin/bash
# developer tool: rocprofv3 per-kernel stats of an arbitrary command run from the repo root
#   tools/kstats.sh <n_rows> <command...>
rows=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- "$@" > /tmp/kstats.log 2>&1
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1)
python - "$f" "$rows" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>7s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} {r['Percentage']:>6s}%")
PY
