#!/bin/bash
# the configs[3] part of tools/r4_final.sh (bench line + rocprofv3 summary), for re-measuring after a change that touches only that path
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $out/r4_bench_c4.json 2> $out/bench_c4.err
bash tools/profile_round.sh r4_c4 --config c4 --warmup 0 > $out/prof_c4.log 2>&1
python tools/summarize_profile.py r4_c4 > $out/summarize_r4_c4.log 2>&1; f=$(find gpurun_out/r4_c4/trace -name "t_kernel_stats.csv" | head -1); cp "$f" $out/r4_c4_kernel_stats.csv; cp profiles/r4_c4_summary.md profiles/r4_c4_summary.json $out/; rm -rf gpurun_out/r4_c4
python -c "import json; d=json.loads(open('$out/r4_bench_c4.json').read().strip().splitlines()[-1]); print(round(d['value'],1), d['roofline']['frac'], d['roofline']['avg_launch_us'], d['eig_ms_per_iter'], d['warm_start']['value'])"
grep -n "Dominant" $out/r4_c4_summary.md | cut -c1-330
