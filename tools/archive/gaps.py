"""Gap analysis of a rocprofv3 kernel trace CSV: time between consecutive fused-step kernels."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps = []; durs = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    if "k_pipe_vec" in n0 and ("k_pipe_vec" in n1 or "k_pipe_tail" in n1):
        gaps.append((s1 - e0) / 1e3); durs.append((e0 - s0) / 1e3)
    elif "k_pipe_tail" in n0 and "k_pipe_vec" in n1:
        gaps.append((s1 - e0) / 1e3)
import statistics as st
big = [g for g in gaps if g > 2.0]
print(f"steps {len(durs)} avg dur {st.mean(durs):.2f} us; gaps: n={len(gaps)} sum={sum(gaps):.0f} us avg={st.mean(gaps):.2f}; gaps>2us: n={len(big)} sum={sum(big):.0f} us; hist:",
      collections.Counter(int(g // 5) * 5 for g in big).most_common(8))
