#!/usr/bin/env python3
"""Idle time of the GPU between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (previous kernel -> next kernel).
usage: gap_analysis.py <kernel_trace.csv> [iterations]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"<.*|\(.*", "", r["Kernel_Name"]).replace("void machip::", "").replace("machip::", "")) for r in rows), key=lambda t: t[0])
# keep the timed region: from the first k_x_bits / k_asm_count after warmup... simply everything between the first and last k_fw_final
idx = [i for i, e in enumerate(ev) if e[2] == "k_fw_final"]
lo = next(i for i, e in enumerate(ev) if e[2] == "k_asm_count")
ev = ev[lo: idx[-1] + 1]
busy = sum(e[1] - e[0] for e in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
for a, b in zip(ev, ev[1:]):
    g = b[0] - a[1]
    if g > 0:
        k = (a[2], b[2]); gaps[k][0] += g; gaps[k][1] += 1
print(f"span {span/1e3/iters:.1f} us/iter, kernels busy {busy/1e3/iters:.1f} us/iter, idle {(span-busy)/1e3/iters:.1f} us/iter over {len(ev)} launches")
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {k[0]:22s} -> {k[1]:22s} {t/1e3/iters:8.1f} us/iter  ({c/iters:6.1f} gaps/iter, avg {t/1e3/c:6.2f} us)")
