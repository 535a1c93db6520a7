#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 -k "streamed or variants_agree or in_process or ipc_ or teacher_forced_config4 or sweep" 2>&1 | tail -5
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],2), 'it/s steps', d.get('lanczos_steps_per_iter'), 'eig ms', round(d.get('eig_ms_per_iter',0),3), 'frac', round(d['roofline']['frac'],4), 'us', round(d['roofline']['avg_launch_us'],3))"; }
for cfg in c4 c2 c5b; do
  timeout 300 python bench.py --config $cfg --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 2 2>/dev/null | one "$cfg:"
done
MACHIP_GRAPH=0 timeout 300 python bench.py --config c4 --steps 20 --warmup 2 --no-cpu --no-pmc --no-warm --min-seconds 2 2>/dev/null | one "c4 eager:"
bash tools/profile_round.sh r5_c4 --config c4 --warmup 0 > gpurun_out/prof_c4.log 2>&1
python tools/summarize_profile.py r5_c4 > gpurun_out/summarize_r5_c4.log 2>&1
sed -n 1,12p profiles/r5_c4_summary.md | cut -c1-400
cp profiles/r5_c4_summary.md gpurun_out/r5_c4_summary_new.md
