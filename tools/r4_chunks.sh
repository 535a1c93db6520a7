#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "32 8" "32 4" "32 2" "16 4" "16 8" "64 8" "24 6"; do set -- $v
  MACHIP_CHUNK=$1 MACHIP_CHUNK_NEAR=$2 timeout 200 python bench.py --config c4 --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 1.5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk $1 near $2:', round(d['value'],2), 'it/s steps', d['lanczos_steps_per_iter'], 'eig ms', round(d['eig_ms_per_iter'],3))"
done
