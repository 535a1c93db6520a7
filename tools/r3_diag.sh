#!/bin/bash
# Round-3 diagnostics on the GPU box: PMC passes of the column-panel step (k_pan_mul / k_pan_fin) next to the gather
# step (k_pipe_vec) on the same iterates (bench trajectory of config 4, iterations 0-11), phase clocks (tools/ubench6).
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3diag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 12 --warmup 0 --no-cpu --no-roofline --min-seconds 0"
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  MACHIP_PANEL=$mode timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -o p -- $B > $out/pmc_${mode}_$i.log 2>&1
  echo "mode $mode pass $i ($set): exit $?" >> $out/passes.txt
  python3 - "$out/pmc$i" "$mode" >> $out/pmc_summary.txt 2>&1 <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("machip::", "").replace("void ", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in acc:
    if "k_pipe_vec" in k or "k_pan_" in k:
        print("MACHIP_PANEL=" + sys.argv[2], k, {c: (v[0], round(v[1] / v[0], 1)) for c, v in acc[k].items()})
PY
  rm -rf $out/pmc$i
done
done
cat $out/passes.txt
cd $GRAFT_REPO_ROOT && timeout 120 tools/bin/ubench6 12 21 > $out/ubench6_12x21.txt 2>&1
