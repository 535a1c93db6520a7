#!/bin/bash
# Round-2 diagnostics on the GPU box: counter list, per-iteration step times at config 4, PMC passes for the fused step.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2diag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1
cd $GRAFT_REPO_ROOT && timeout 600 python tools/iter_stats.py c4 20 > $out/iter_stats_c4.txt 2>&1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 8 --warmup 0 --no-cpu --no-roofline"
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc$i -o p -- $B > $out/pmc$i.log 2>&1
  echo "pass $i ($set): exit $?" >> $out/passes.txt
  python3 - "$out/pmc$i" >> $out/pmc_summary.txt 2>&1 <<'PY'
import csv, sys, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("machip::", "").replace("void ", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in acc:
    if "k_pipe_vec" in k or "k_asm" in k or "k_grad" in k or "k_sel_pass" in k or "k_fw_final" in k:
        print(k, {c: (v[0], round(v[1] / v[0], 1)) for c, v in acc[k].items()})
PY
  rm -rf $out/pmc$i
done
cat $out/passes.txt
tail -30 $out/iter_stats_c4.txt
