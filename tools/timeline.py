#!/usr/bin/env python3
"""Timeline of ONE Frank-Wolfe iteration from a rocprofv3 kernel-trace CSV: every kernel with its duration and the idle gap
before it (host round trips show up as gaps of tens of microseconds).  usage: timeline.py trace.csv [iteration index]"""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void machip::", "").replace("machip::", "")[:44]))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_asm_count")]
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
a, b = starts[it], starts[it + 1]
t0 = rows[a][0]
busy = sum(e - s for s, e, _ in rows[a:b]) / 1e3
print(f"iteration {it}: {b - a} kernels, wall {(rows[b][0] - t0) / 1e3:.1f} us, kernels busy {busy:.1f} us")
prev_end = rows[a][0]
agg = {}
for s, e, n in rows[a:b]:
    gap = (s - prev_end) / 1e3
    k = agg.setdefault(n, [0, 0.0, 0.0]); k[0] += 1; k[1] += (e - s) / 1e3; k[2] += gap
    if gap > 8.0 or not n.startswith("k_lan_persist") and not n.startswith("k_pipe_vec") and not n.startswith("k_pan_"):
        print(f"  +{(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  dur {(e - s) / 1e3:7.1f}  {n}")
    prev_end = e
print("per kernel: calls, busy us, gap-before us")
for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"  {n:44s} {c:4d} {d:9.1f} {g:9.1f}")
