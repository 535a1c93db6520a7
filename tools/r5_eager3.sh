#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('$1', round(d['value'],2), 'it/s steps', d.get('lanczos_steps_per_iter'), 'eig ms', round(d.get('eig_ms_per_iter',0),3), 'frac', round(r.get('frac',0),4), 'us', round(r.get('avg_launch_us',0),3))"; }
for rep in 1 2; do
for cfg in ${CFGS:-c4s c4}; do
for g in auto 0 1; do
  if [ $g = auto ]; then unset MACHIP_GRAPH; else export MACHIP_GRAPH=$g; fi
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 2 2>/dev/null | one "$cfg graph=$g:"
done
done
done
