#!/bin/bash
# Round-3 robustness round on the GPU box: solver-mode fuzz, soak (repeated suites + bench determinism), the -m gpu suite
# against an AddressSanitizer build of the host side (mac_amd/libmachip_asan.so, built with -Xarch_host -fsanitize=address).
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3robust
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python tools/fuzz_modes.py 80 0 0 > $out/fuzz_auto.txt 2>&1; tail -1 $out/fuzz_auto.txt
timeout 600 python tools/fuzz_modes.py 30 300 120 > $out/fuzz_modes.txt 2>&1; tail -1 $out/fuzz_modes.txt
bash tools/soak.sh 3 > $out/soak.txt 2>&1; tail -8 $out/soak.txt
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
if [ -f mac_amd/libmachip_asan.so ]; then
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 MACHIP_LIB=$PWD/mac_amd/libmachip_asan.so \
    timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "sweep or advice or in_process or row_partitioned or eval_batch or teacher_forced_city or panel_step or pose_graph_fiedler or solver_variants or hub_rows or padded_fixed_width or random_chain or rounding or ties" > $out/asan.txt 2>&1
  grep -E "passed|failed|ERROR: AddressSanitizer|SUMMARY" $out/asan.txt | tail -5
fi
