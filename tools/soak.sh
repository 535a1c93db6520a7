#!/bin/bash
# developer tool: repeat the GPU suite and the bench trajectory to catch rare (timing-dependent) failures
n=${1:-10}
fail=0
for i in $(seq 1 $n); do
  r=$(timeout 300 python -m pytest tests -m gpu -x -q --timeout 150 2>&1 | grep -E "passed|failed" | tail -1)
  echo "suite $i: $r"
  case "$r" in *failed*) fail=1;; esac
done
prev=""
for i in $(seq 1 $n); do
  h=$(python bench.py --no-cpu --no-roofline --no-pmc --min-seconds 0 --steps 20 --warmup 1 2>/dev/null | python -c "import sys,json,hashlib; d=json.loads(sys.stdin.read()); print(d['lambda2_first_last'], d['lanczos_steps_per_iter'])")
  echo "bench $i: $h"
  if [ -n "$prev" ] && [ "$h" != "$prev" ]; then echo "NONDETERMINISTIC"; fail=1; fi
  prev="$h"
done
echo "soak fail=$fail"
