#!/bin/bash
# Round-3 measurement round on the GPU box: bench lines of every config, rocprofv3 summaries (c4, c2), sweep / panel probes.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/r3_bench_c4.json 2> $out/bench_c4.err
for c in c2 c3 c5a c5b; do timeout 600 python bench.py --config $c > $out/r3_bench_$c.json 2> $out/bench_$c.err; done
timeout 300 python bench.py --config c5 > $out/r3_bench_c5.json 2>> $out/bench_c5.err
timeout 300 python bench.py --config c5 --precision 1 > $out/r3_bench_c5_mixed.json 2>> $out/bench_c5.err
bash tools/profile_round.sh r3_c4 --config c4 --warmup 0 > $out/prof_c4.log 2>&1
bash tools/profile_round.sh r3_c2 --config c2 --warmup 0 > $out/prof_c2.log 2>&1
for g in intel sphere2500 city10000; do timeout 300 python tools/sweep_probe.py $g 16 20 2>&1 | tail -6; done > $out/r3_sweep.txt
timeout 900 python tools/sweep_panel.py c4 20 0,1 > $out/r3_sweep_panel_c4.txt 2>&1
timeout 120 tools/bin/ubench6 12 21 > $out/r3_ubench6_phase_clocks.txt 2>&1
tail -3 $out/r3_sweep.txt; tail -4 $out/r3_sweep_panel_c4.txt
for f in $out/r3_bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d.get("roofline", {}).get("frac"), d.get("speedup_vs_cpu"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
