#!/bin/bash
# Round-4 measurement round on the GPU box: bench lines of every config, rocprofv3 summaries (c4, c2, c3), sweeps.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/r4_bench_c4.json 2> $out/bench_c4.err
for c in c2 c3 c5a c5b; do timeout 600 python bench.py --config $c > $out/r4_bench_$c.json 2> $out/bench_$c.err; done
timeout 300 python bench.py --config c5 > $out/r4_bench_c5.json 2>> $out/bench_c5.err
timeout 300 python bench.py --config c5 --precision 1 > $out/r4_bench_c5_mixed.json 2>> $out/bench_c5.err
timeout 300 python bench.py --config c5s > $out/r4_bench_c5s.json 2>> $out/bench_c5.err
bash tools/profile_round.sh r4_c4 --config c4 --warmup 0 > $out/prof_c4.log 2>&1
bash tools/profile_round.sh r4_c2 --config c2 --warmup 0 > $out/prof_c2.log 2>&1
bash tools/profile_round.sh r4_c3 --config c3 --warmup 0 > $out/prof_c3.log 2>&1
# condense on the box (the raw traces exceed what gpurun copies back): summaries -> $out, raw CSVs dropped
mkdir -p profiles_box; for t in r4_c4 r4_c2 r4_c3; do python tools/summarize_profile.py $t > $out/summarize_$t.log 2>&1; f=$(find gpurun_out/$t/trace -name "t_kernel_stats.csv" | head -1); cp "$f" $out/${t}_kernel_stats.csv; cp profiles/${t}_summary.md profiles/${t}_summary.json $out/; rm -rf gpurun_out/$t; done
timeout 600 python tools/g2o_first_budget.py tests/golden/data/*.g2o > $out/r4_first_budget.txt 2>&1
for g in intel sphere2500 city10000; do timeout 300 python tools/sweep_probe.py $g 16 20 2>&1 | tail -6; done > $out/r4_sweep.txt
for f in $out/r4_bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d.get("roofline", {}).get("frac"), d.get("warm_start", {}).get("value"), d.get("lanczos_steps_per_iter"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
