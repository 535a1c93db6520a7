#!/bin/bash
# Round 4 (late): why does the gather step take 14.3 us on configs[3] iterate 1 and 10.0 us on iterate 0 with the same number of entries?
# PMC passes of k_pipe_vec: a run of iterate 0 only against a run of iterates 0-1 (per-launch averages; iterate 1 = the difference).
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r4hub
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for it in 1 2; do
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum" \
           "TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum"; do
  i=$((i+1))
  PYTHONPATH=$GRAFT_REPO_ROOT MACHIP_PANEL=0 timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p -o p -- python $GRAFT_REPO_ROOT/tools/city_exact_probe.py $it c4 > $out/log_${it}_${i}.txt 2>&1
  python3 - "$out/p" "$it" >> $out/summary.txt 2>&1 <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pipe_vec" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("iters", sys.argv[2], {c: (v[0], round(v[1] / max(1, v[0]), 1)) for c, v in acc.items()})
PY
  rm -rf $out/p
done
done
cat $out/summary.txt
