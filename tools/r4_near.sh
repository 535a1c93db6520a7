#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "20 8" "30 8" "40 8" "30 16" "40 16" "50 16" "30 12"; do set -- $v
  MACHIP_NEAR_X10=$1 MACHIP_CHUNK_NEAR=$2 timeout 200 python bench.py --config c4 --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 1.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('near_x10 $1 chunk_near $2:', round(d['value'],2), 'it/s steps', d['lanczos_steps_per_iter'], 'eig ms', round(d['eig_ms_per_iter'],3))"
done
