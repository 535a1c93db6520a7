#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "er_size_budget or bench or two_ranks" > $out/pytest.log 2>&1; tail -5 $out/pytest.log
for c in c4s c2s; do timeout 900 python bench.py --config $c --steps 20 --warmup 2 > $out/bench_$c.json 2> $out/bench_$c.err; echo "$c rc=$?"; tail -c 400 $out/bench_$c.err; cut -c1-1800 $out/bench_$c.json; done
for c in c3 c5b c5a; do timeout 600 python bench.py --config $c --no-cpu --no-pmc --no-warm > $out/bench_$c.json 2> $out/bench_$c.err; echo "$c rc=$?"; python -c "
import json,sys; d=json.loads(open('$out/bench_$c.json').read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1), r['kernel'][:90], round(r['frac'],4), round(r['avg_launch_us'],2), r['solver_modes'])"; done
