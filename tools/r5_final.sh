#!/bin/bash
# Round-5 measurement round on the GPU box: bench lines of every config, rocprofv3 summaries (c4, c2, c3), sweeps.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r5final2
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/r5_bench_c4.json 2> $out/bench_c4.err
for c in c2 c3 c5a c5b; do timeout 600 python bench.py --config $c > $out/r5_bench_$c.json 2> $out/bench_$c.err; done
timeout 300 python bench.py --config c5 > $out/r5_bench_c5.json 2>> $out/bench_c5.err
timeout 300 python bench.py --config c5 --precision 1 > $out/r5_bench_c5_mixed.json 2>> $out/bench_c5.err
timeout 300 python bench.py --config c5s > $out/r5_bench_c5s.json 2>> $out/bench_c5.err
timeout 600 python bench.py --config c4s > $out/r5_bench_c4s.json 2>> $out/bench_c4s.err
timeout 600 python bench.py --config c2s > $out/r5_bench_c2s.json 2>> $out/bench_c4s.err
MACHIP_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 2 > $out/r5_bench_c4_2ranks_one_gpu.json 2> $out/bench_2r.err
bash tools/profile_round.sh r5_c4 --config c4 --warmup 0 > $out/prof_c4.log 2>&1
bash tools/profile_round.sh r5_c2 --config c2 --warmup 0 > $out/prof_c2.log 2>&1
bash tools/profile_round.sh r5_c3 --config c3 --warmup 0 > $out/prof_c3.log 2>&1
bash tools/profile_round.sh r5_c5b --config c5b --warmup 0 > $out/prof_c5b.log 2>&1
mkdir -p profiles_box; for t in r5_c4 r5_c2 r5_c3 r5_c5b; do python tools/summarize_profile.py $t > $out/summarize_$t.log 2>&1; f=$(find gpurun_out/$t/trace -name "t_kernel_stats.csv" | head -1); cp "$f" $out/${t}_kernel_stats.csv; cp profiles/${t}_summary.md profiles/${t}_summary.json $out/; rm -rf gpurun_out/$t; done
for f in $out/r5_bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d.get("roofline", {}).get("frac"), d.get("warm_start", {}).get("value"), d.get("lanczos_steps_per_iter"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
