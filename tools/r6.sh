#!/bin/bash
# Round-6 GPU sessions (one gpurun call each): tools/r6.sh <stage>; output under gpurun_out/r6_<stage>/
set -u
stage=${1:-a}
out=gpurun_out/r6_$stage
mkdir -p $out
export TMPDIR=/tmp
case $stage in
a)  # shifted recurrence (panel_u.h): correctness on the forced-panel tests, then the shape sweep at configs[3]
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver_variants_agree or panel_step_multi_round or panel_band_split or column_panel_step_is_bit" > $out/tests.txt 2>&1; tail -5 $out/tests.txt
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "teacher_forced_config4" > $out/tests_c4.txt 2>&1; tail -5 $out/tests_c4.txt
    timeout 1200 python tools/sweep_panel_u.py c4 20 > $out/sweep_c4.txt 2>&1; tail -16 $out/sweep_c4.txt
    ;;
b)  # after moving the prologue into k_pan_mul8's workgroup 0: sweep again, then per-kernel durations of four variants
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver_variants_agree or teacher_forced_config4" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
    timeout 1200 python tools/sweep_panel_u.py c4 20 > $out/sweep_c4.txt 2>&1; tail -13 $out/sweep_c4.txt
    tools/kstats.sh 24 python tools/sweep_panel_u.py c4 20 0,1,4,6 > $out/kstats.txt 2>&1; cat $out/kstats.txt
    ;;
c)  # serpentine tile dealing on / off; streaming bandwidth of this box by working-set size (is a 48 MB matrix served faster than HBM?)
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver_variants_agree or teacher_forced_config4 or panel_step_multi_round or column_panel_step_is_bit" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
    timeout 1200 python tools/sweep_panel_u.py c4 20 0,11,6,12,4,13 > $out/sweep_c4.txt 2>&1; tail -9 $out/sweep_c4.txt
    python - > $out/membench.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, ".")
from mac_amd import _lib
for mb in (16, 32, 48, 64, 96, 128, 192, 256, 512, 1024):
    print(mb, "MiB:", _lib.membench(mb << 20, 20), flush=True)
PY
    cat $out/membench.txt
    ;;
d)  # phase clocks of the matrix kernel (developer build with -DPAN_CLOCKS) on the densest iterate
    MACHIP_LIB=mac_amd/libmachip_clk.so timeout 900 python tools/pan_clocks.py c4 19 panel=1,panel_u=0 panel=1,panel_u=1 panel=1,panel_u=1,panel_np=8,panel_nb=32 panel=1,panel_u=1,panel_np=6,panel_nb=42 > $out/clocks.txt 2>&1; cat $out/clocks.txt
    ;;
e)  # k_pan_mul8 with countable loads (operand into LDS first, chunks multiplied as they land): tests, sweep, clocks
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "solver_variants_agree or teacher_forced_config4 or panel_step_multi_round or column_panel_step_is_bit or panel_band_split" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
    timeout 1200 python tools/sweep_panel_u.py c4 20 0,1,4,5,6 > $out/sweep_c4.txt 2>&1; tail -8 $out/sweep_c4.txt
    MACHIP_LIB=mac_amd/libmachip_clk.so timeout 900 python tools/pan_clocks.py c4 19 panel=1,panel_u=0 panel=1,panel_u=1 panel=1,panel_u=1,panel_np=6,panel_nb=42 > $out/clocks.txt 2>&1; cat $out/clocks.txt
    ;;
f)  # how many groups of chunk loads leave before the panel is in LDS (PAN_U_AHEAD builds)
    for a in 1 2 3 5; do
      echo "== AHEAD $a" >> $out/sweep.txt
      MACHIP_LIB=mac_amd/libmachip_a$a.so timeout 600 python tools/sweep_panel_u.py c4 20 1,4,6 2>&1 | tail -4 >> $out/sweep.txt
    done
    cat $out/sweep.txt
    ;;
g)  # full GPU test suite + default bench line
    timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1; tail -15 $out/tests.txt
    timeout 900 python bench.py > $out/bench_c4.json 2> $out/bench_c4.err; cut -c1-1500 $out/bench_c4.json
    ;;
h)  # advisor fixes + landscape gate: full GPU suite, X-block probe, pose-graph benches, default bench
    timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1; tail -8 $out/tests.txt
    timeout 300 python tools/xblock_probe.py > $out/xblock.txt 2>&1; cat $out/xblock.txt
    for c in c5b c5a c3 c2; do timeout 600 python bench.py --config $c --no-cpu > $out/bench_$c.json 2> $out/bench_$c.err; python -c "import json,sys; d=json.loads(open('$out/bench_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d.get('lanczos_steps_per_iter'), d['roofline']['frac'] if d.get('roofline') else None)"; done
    timeout 900 python bench.py > $out/bench_c4.json 2> $out/bench_c4.err; python -c "import json,sys; d=json.loads(open('$out/bench_c4.json').read().strip().splitlines()[-1]); print('c4', d['value'], d.get('lanczos_steps_per_iter'), d['roofline']['frac'], d['roofline']['traffic'])"
    ;;
j)  # mixed panel mode (late fp32 tiles) + everything since h
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "teacher_forced_config4 or x_block or landscape or mixed or precision" > $out/tests_a.txt 2>&1; tail -6 $out/tests_a.txt
    for p in 0 1; do timeout 600 python bench.py --config c4 --no-cpu --no-warm --no-pmc --no-same-node --precision $p > $out/bench_c4_p$p.json 2> $out/bench_c4_p$p.err; python -c "import json,sys; d=json.loads(open('$out/bench_c4_p$p.json').read().strip().splitlines()[-1]); print('c4 precision $p', d['value'], d.get('lanczos_steps_per_iter'), d['roofline']['frac'], d['roofline']['solver_modes'])"; done
    timeout 2400 python -m pytest tests -x -q -m gpu > $out/tests.txt 2>&1; tail -6 $out/tests.txt
    ;;
final)  # measurement round: bench lines of every config, rocprofv3 summaries (c4, c2, c3, c5b), the mixed leg, two ranks on one GPU
    R=${2:-r6}
    timeout 900 python bench.py > $out/${R}_bench_c4.json 2> $out/bench_c4.err
    for c in c2 c3 c5a c5b; do timeout 600 python bench.py --config $c > $out/${R}_bench_$c.json 2> $out/bench_$c.err; done
    timeout 300 python bench.py --config c5 > $out/${R}_bench_c5.json 2>> $out/bench_c5.err
    timeout 300 python bench.py --config c5 --precision 1 > $out/${R}_bench_c5_mixed.json 2>> $out/bench_c5.err
    timeout 600 python bench.py --config c4 --precision 1 --no-cpu --no-same-node > $out/${R}_bench_c4_mixed.json 2>> $out/bench_c4m.err
    timeout 300 python bench.py --config c5s > $out/${R}_bench_c5s.json 2>> $out/bench_c5.err
    timeout 600 python bench.py --config c4s > $out/${R}_bench_c4s.json 2>> $out/bench_c4s.err
    timeout 600 python bench.py --config c2s > $out/${R}_bench_c2s.json 2>> $out/bench_c4s.err
    MACHIP_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 2 > $out/${R}_bench_c4_2ranks_one_gpu.json 2> $out/bench_2r.err
    for c in c4 c2 c3 c5b; do bash tools/profile_round.sh ${R}_$c --config $c --warmup 0 > $out/prof_$c.log 2>&1; done
    for t in ${R}_c4 ${R}_c2 ${R}_c3 ${R}_c5b; do python tools/summarize_profile.py $t > $out/summarize_$t.log 2>&1; f=$(find gpurun_out/$t/trace -name "t_kernel_stats.csv" | head -1); cp "$f" $out/${t}_kernel_stats.csv; cp profiles/${t}_summary.md profiles/${t}_summary.json $out/; rm -rf gpurun_out/$t; done
    for f in $out/${R}_bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d.get("roofline", {}).get("frac"), d.get("roofline", {}).get("traffic"), d.get("warm_start", {}).get("value"), d.get("lanczos_steps_per_iter"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
    done
    ;;
l)  # row-partitioned panel step between processes
    timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "test_ipc_row_partitioned" > $out/tests.txt 2>&1; grep -E "panel step, us|passed|failed|Error" $out/tests.txt | tail -12
    ;;
esac
