#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_summary.md + .json.
HBM traffic per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes): MI355X_MICROARCH.md section HBM says
FETCH_SIZE on gfx950 counts 128-B requests at 64 B (x2 for wide coalesced reads; other widths and
WRITE_SIZE uncalibrated), both counters come from separate --pmc passes."""
import csv, json, os, sys, collections
tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
out_md = os.path.join("profiles", tag + "_summary.md")
out_js = os.path.join("profiles", tag + "_summary.json")

def short(n):
    n = n.replace("machip::", "").replace("void ", "")
    return n.split("(")[0]

import glob
stats = []
cands = glob.glob(os.path.join(src, "trace", "**", "t_kernel_stats.csv"), recursive=True)
with open(cands[0]) as fh:
    for r in csv.DictReader(fh):
        stats.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]),
                      float(r["Percentage"]), float(r["MinNs"]), float(r["MaxNs"])))

def pmc(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    if not os.path.exists(path):
        path = (glob.glob(os.path.join(os.path.dirname(path), "**", os.path.basename(path)), recursive=True) or [path])[0]
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / max(1, v[0])) for k, v in acc.items()}

fetch = pmc(os.path.join(src, "pmc_fetch", "f_counter_collection.csv"), "FETCH_SIZE")
write = pmc(os.path.join(src, "pmc_write", "w_counter_collection.csv"), "WRITE_SIZE")
bench_line = None
for line in open(os.path.join(src, "trace.log")):
    if line.startswith('{"metric'):
        bench_line = json.loads(line)
rows = []
for (n, calls, tot, avg, pct, mn, mx) in stats:
    f = fetch.get(n, (0, 0.0))[1]; w = write.get(n, (0, 0.0))[1]
    rows.append(dict(kernel=n, calls=calls, total_ms=tot / 1e6, avg_us=avg / 1e3, pct=pct, min_us=mn / 1e3, max_us=mx / 1e3,
                     fetch_kib=f, write_kib=w, hbm_bytes_per_launch=(2.0 * f + w) * 1024.0))
# A Lanczos step = one launch of k_pipe_vec / k_pipe_stream (gather form), or k_pan_mul + k_pan_fin (column-panel form).
heads = [r for r in rows if r["kernel"].startswith(("k_pipe_vec", "k_pipe_stream", "k_pan_mul"))]
parts = [r for r in rows if r["kernel"].startswith(("k_pipe_vec", "k_pipe_stream", "k_pan_mul", "k_pan_fin"))]
calls = sum(r["calls"] for r in heads)
dom = dict(kernel="one Lanczos step: k_pipe_vec (gather form) or k_pan_mul8 + k_pan_finu (column-panel form, 8-byte operand; k_pan_mul + k_pan_fin in record form)", calls=calls,
           avg_us=sum(r["avg_us"] * r["calls"] for r in parts) / max(1, calls),
           hbm_bytes_per_launch=sum(r["hbm_bytes_per_launch"] * r["calls"] for r in parts) / max(1, calls),
           gather_steps=sum(r["calls"] for r in heads if not r["kernel"].startswith("k_pan")),
           panel_steps=sum(r["calls"] for r in heads if r["kernel"].startswith("k_pan")),
           gather_avg_us=sum(r["avg_us"] * r["calls"] for r in parts if not r["kernel"].startswith("k_pan")) / max(1, sum(r["calls"] for r in heads if not r["kernel"].startswith("k_pan"))),
           panel_avg_us=sum(r["avg_us"] * r["calls"] for r in parts if r["kernel"].startswith("k_pan")) / max(1, sum(r["calls"] for r in heads if r["kernel"].startswith("k_pan"))))
if calls == 0:
    # no Lanczos step in the trace (intel: every solve runs in the exact chain + closures mode): the unit is one
    # preconditioned iteration = everything between two k_lob_update / k_lob_fused launches (tridiagonal solve, g / h / w of the
    # Woodbury correction, the product, the update); the inverse (k_gj_step) and the set-up kernels are listed in the table
    it_parts = [r for r in rows if r["kernel"].startswith(("k_lob_update", "k_lob_fused", "k_tri_solve", "k_wb_g", "k_wb_h", "k_wb_w", "k_spmv"))]
    its = sum(r["calls"] for r in rows if r["kernel"].startswith(("k_lob_update", "k_lob_fused")))
    dom = dict(kernel="one preconditioned (exact chain + closures) iteration: k_tri_solve / k_lob_fused + k_wb_g / k_wb_h / k_wb_w + product + update",
               calls=its, avg_us=sum(r["avg_us"] * r["calls"] for r in it_parts) / max(1, its),
               hbm_bytes_per_launch=sum(r["hbm_bytes_per_launch"] * r["calls"] for r in it_parts) / max(1, its),
               gather_steps=0, panel_steps=0, gather_avg_us=0.0, panel_avg_us=0.0,
               inverse_launches=sum(r["calls"] for r in rows if r["kernel"].startswith("k_gj_step")),
               inverse_avg_us=sum(r["avg_us"] * r["calls"] for r in rows if r["kernel"].startswith("k_gj_step")) / max(1, sum(r["calls"] for r in rows if r["kernel"].startswith("k_gj_step"))))
js = dict(tag=tag, bench=bench_line, dominant=dom, kernels=rows)
json.dump(js, open(out_js, "w"), indent=1)
with open(out_md, "w") as fh:
    fh.write(f"# rocprofv3 summary `{tag}`\n\nCommand: `rocprofv3 --kernel-trace --stats -- python bench.py ...` plus two `--pmc` passes "
             "(FETCH_SIZE, WRITE_SIZE); raw CSVs were in `gpurun_out/" + tag + "/` (scratch).\n\n")
    if bench_line:
        fh.write("bench line of the traced run (profiler attached, so slower than the un-profiled number):\n\n```\n" + json.dumps(bench_line) + "\n```\n\n")
    if "inverse_launches" in dom:
        fh.write(f"Dominant work: **{dom['kernel']}**, {dom['calls']} iterations, kernel time per iteration {dom['avg_us']:.2f} us, HBM traffic per "
                 f"iteration (2*FETCH+WRITE) {dom['hbm_bytes_per_launch']/1e6:.3f} MB; the s x s inverse: {dom['inverse_launches']} launches of k_gj_step at "
                 f"{dom['inverse_avg_us']:.2f} us.  No Lanczos step ran in this trace, so the bench line's roofline object (defined per Lanczos step) has "
                 f"nothing to be cross-checked against: the working set lives in L2 / LDS and the solve is a latency chain (DESIGN section 5).\n\n")
    else:
        fh.write(f"Dominant work: **{dom['kernel']}**, {dom['calls']} steps ({dom['gather_steps']} gather-form at {dom['gather_avg_us']:.2f} us, "
                 f"{dom['panel_steps']} panel-form at {dom['panel_avg_us']:.2f} us = k_pan_mul8 + k_pan_finu), kernel time per step {dom['avg_us']:.2f} us, "
                 f"HBM traffic per step (2*FETCH+WRITE) {dom['hbm_bytes_per_launch']/1e6:.2f} MB.\n\n")
    if bench_line and bench_line.get("roofline") and dom["calls"] > 0 and "inverse_launches" not in dom:
        r = bench_line["roofline"]
        frac_trace = r["algorithmic_bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9 / r["peak"]
        fh.write(f"Cross-check of the bench line's roofline object: in-solve step time (hipEvents around the Krylov chunks, "
                 f"tail kernels and inter-kernel gaps included) {r['avg_launch_us']:.2f} us over {r['launches_timed']} launches -> frac "
                 f"{r['frac']:.4f}; rocprofv3 kernel average {dom['avg_us']:.2f} us over {dom['calls']} launches -> frac {frac_trace:.4f} "
                 f"(ratio {r['avg_launch_us'] / dom['avg_us']:.3f}).  With a profiler attached libmachip launches eagerly instead of "
                 f"through hipGraphs (rocprofiler-sdk crashes on short graphs, DESIGN section 5), so on small matrices the in-solve figure "
                 f"of THIS traced run contains launch gaps the un-profiled bench line does not have.\n\n")
    fh.write("| kernel | calls | total ms | avg us | min us | max us | % | FETCH KiB/launch | WRITE KiB/launch |\n|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        fh.write(f"| `{r['kernel'][:60]}` | {r['calls']} | {r['total_ms']:.2f} | {r['avg_us']:.2f} | {r['min_us']:.2f} | {r['max_us']:.2f} | "
                 f"{r['pct']:.2f} | {r['fetch_kib']:.1f} | {r['write_kib']:.1f} |\n")
# hand-written notes of the round (what was tried, before / after tables) ride along: profiles/<tag>_notes.md
notes = os.path.join("profiles", tag + "_notes.md")
if os.path.exists(notes):
    with open(out_md, "a") as fh:
        fh.write(open(notes).read())
print(open(out_md).read()[:3000])
