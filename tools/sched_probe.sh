#!/bin/bash
# Trace of the chunk scheduler (MACHIP_DEBUG=1) over one pass of a bench config: at which step the residual estimate
# crossed the target and at which step the explicit check ran -- how many steps a solve overshoots.
# usage: sched_probe.sh [c4|c2|...] [iters]
cfg=${1:-c4}; it=${2:-20}
mkdir -p gpurun_out
MACHIP_DEBUG=1 python bench.py --config $cfg --steps $it --warmup 0 --max-repeats 1 --min-seconds 0 --no-cpu --no-pmc --no-roofline 2> gpurun_out/sched_$cfg.log | tail -1 > gpurun_out/sched_$cfg.json
grep -c "check J" gpurun_out/sched_$cfg.log
python - "$cfg" <<'PY'
import re, sys
cfg = sys.argv[1]
solves = []; cur = []
for line in open(f"gpurun_out/sched_{cfg}.log"):
    m = re.search(r"\] (\w+) J=(\d+) Jeff=\d+ theta=\S+ est=(\S+) to_go=(\S+) broke=\d pend=(\d+)", line)
    if m:
        cur.append((int(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5)))); continue
    m = re.search(r"check J=(\d+) rq=\S+ res=(\S+) \(tol (\S+)\)", line)
    if m:
        J, res, tol = int(m.group(1)), float(m.group(2)), float(m.group(3))
        if res < tol:
            solves.append((cur, J, res)); cur = []
tot_over = 0; tot = 0
for cur, J, res in solves:
    # estimated crossing step by log-linear interpolation of est between analyses
    import math
    tgt = 1.5e-8
    cross = None
    for (j0, e0, _, _), (j1, e1, _, _) in zip(cur, cur[1:]):
        if e0 >= tgt > e1 and e0 > 0 and e1 > 0:
            cross = j0 + (j1 - j0) * (math.log(e0) - math.log(tgt)) / (math.log(e0) - math.log(e1)); break
    chunks = [b[0] - a[0] for a, b in zip(cur, cur[1:])]
    last = cur[-1]
    print(f"solve: analyses {len(cur)} check at J={J} res={res:.2e} est-crossing ~{cross if cross is None else round(cross,1)} last chunks {chunks[-6:]} pend at check {last[3]}")
    if cross: tot_over += J - cross; tot += J
print("mean overshoot steps/solve", tot_over / max(1, len(solves)), "of", tot / max(1, len(solves)))
PY
