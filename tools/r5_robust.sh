#!/bin/bash
# Round-5 robustness round on the GPU box: solver-mode fuzz, soak (repeated suites + bench determinism), GPU tests against an
# AddressSanitizer build of the host side (mac_amd/libmachip_asan.so: hipcc -Xarch_host -fsanitize=address, built beforehand).
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/r5robust
mkdir -p $out
timeout 900 python tools/fuzz_modes.py 60 0 0 > $out/fuzz_auto.txt 2>&1; tail -1 $out/fuzz_auto.txt
timeout 600 python tools/fuzz_modes.py 24 300 120 > $out/fuzz_modes.txt 2>&1; tail -1 $out/fuzz_modes.txt
bash tools/soak.sh 2 > $out/soak.txt 2>&1; tail -6 $out/soak.txt
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
if [ -f mac_amd/libmachip_asan.so ]; then
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 MACHIP_LIB=$PWD/mac_amd/libmachip_asan.so \
    timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x \
    -k "ipc_ or dry_run or sweep or advice or in_process or eval_batch or exact_chain or stiff_chain or pose_graph_fiedler or panel or teacher_forced_config2 or fw_run or option or budget or lanes" > $out/asan.txt 2>&1
  grep -E "passed|failed|ERROR: AddressSanitizer|SUMMARY" $out/asan.txt | tail -8
fi
