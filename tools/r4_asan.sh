#!/bin/bash
# round 4: the -m gpu tests that touch this round's host code against an AddressSanitizer build of the host side
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 MACHIP_LIB=$PWD/mac_amd/libmachip_asan.so \
  timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_integration_patch.py -q -m gpu \
  -k "ipc_ or dry_run or sweep or advice or in_process or eval_batch or exact_chain or stiff_chain or fixed_off_chain or pose_graph_fiedler or solver_variants or panel or teacher_forced_config2 or patched" > gpurun_out/r4_asan.txt 2>&1
grep -E "passed|failed|ERROR: AddressSanitizer|SUMMARY" gpurun_out/r4_asan.txt | tail -8
