#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --warmup 0 --no-cpu --no-pmc --no-warm --no-roofline --min-seconds 0 --max-repeats 1 --steps 8"
for v in "MACHIP_PANEL=0" "MACHIP_ASM_G=16" "MACHIP_ASM_G=64" "MACHIP_ASM_MAXGRID=1024" "MACHIP_DEBUG=0"; do
  env $v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
  f=$(find $out/trace -name "t_kernel_stats.csv" | head -1); echo "== $v"
  python - "$f" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_asm" in r["Name"] or "k_pan_build" in r["Name"]: print("  ", r["Name"][14:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
  rm -rf $out/trace
done
