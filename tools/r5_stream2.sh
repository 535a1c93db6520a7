#!/bin/bash
# Streamed records: margin / far-threshold sweep (bench lines), then traces.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],2), 'it/s steps', d.get('lanczos_steps_per_iter'), 'eig ms', round(d.get('eig_ms_per_iter',0),3), 'frac', round(d['roofline']['frac'],4), 'us', round(d['roofline']['avg_launch_us'],3))"; }
for cfg in ${CFGS:-c4 c2 c5b}; do
for v in "0 0 80" "1 0 80" "1 1 80" "1 0 56" "1 0 120"; do set -- $v
  MACHIP_STREAM=$1 MACHIP_STREAM_MARGIN=$2 MACHIP_STREAM_FAR=$3 timeout 300 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 2 2>/dev/null | one "$cfg stream=$1 margin=$2 far=$3:"
done
done
for cfg in ${TRACE:-c4 c2 c5b}; do
MACHIP_DEBUG=1 MACHIP_STREAM=1 timeout 300 python bench.py --config $cfg --steps 20 --warmup 0 --max-repeats 1 --min-seconds 0 --no-cpu --no-pmc --no-roofline --no-warm 2> gpurun_out/stream_trace_$cfg.log | tail -1 > gpurun_out/stream_trace_$cfg.json
python - $cfg <<'PY'
import re, sys
cfg = sys.argv[1]
tot_ran = tot_used = n = 0; fails = 0; sur = []
for line in open(f"gpurun_out/stream_trace_{cfg}.log"):
    m = re.search(r"check J=(\d+) rq=\S+ res=(\S+) \(tol (\S+)\) ran=(\d+)", line)
    if m:
        J, res, tol, ran = int(m.group(1)), float(m.group(2)), float(m.group(3)), int(m.group(4))
        if res < tol: tot_ran += ran; tot_used += J; n += 1; sur.append(ran - J)
        else: fails += 1; print(f"  FAILED check at J={J} ran={ran} res={res:.2e}")
print(cfg, "solves", n, "failed checks", fails, "mean used", tot_used / max(n, 1), "mean ran", tot_ran / max(n, 1), "surplus", (tot_ran - tot_used) / max(n, 1), sur)
PY
done
