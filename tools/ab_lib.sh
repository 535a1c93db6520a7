#!/bin/bash
# Same-box A/B of two builds of the library (MACHIP_LIB override): alternating bench passes.
# usage: ab_lib.sh libA.so libB.so cfg [cfg ...]
A=$1; B=$2; shift 2
for c in "$@"; do
  for r in 1 2 3; do
    for L in $A $B; do
      MACHIP_LIB=$PWD/$L python bench.py --config $c --no-cpu --no-pmc | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', '$L', round(d['value'],1), d.get('lanczos_steps_per_iter'), round(d['roofline']['avg_launch_us'],3))"
    done
  done
done
