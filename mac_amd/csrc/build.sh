#!/bin/bash
# Build libmachip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
ROCM=${ROCM_PATH:-/opt/rocm}
# (linked under a temporary name and renamed: a process that has the old library mapped keeps its own copy)
# -amdgpu-kernarg-preload-count: leading scalar kernel arguments arrive in SGPRs with the wave instead of behind a scalar
# load of the argument block (k_pipe_vec takes its first-load pointers that way, PIPE_ARGS in kernels.h: config 2 +3.5 %)
# MACHIP_BUILD_FLAGS=-DMACHIP_EXPERIMENTS MACHIP_BUILD_OUT=libmachip_exp.so: the developer build with the measured-slower variants (tools/)
OUT=${MACHIP_BUILD_OUT:-libmachip.so}
"$ROCM/bin/hipcc" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC machip.hip -mllvm -amdgpu-kernarg-preload-count=16 ${MACHIP_BUILD_FLAGS:-} \
    -o ../libmachip.build.$$.so -L"$ROCM/lib" -lrccl -ldl -Wl,-rpath,"$ROCM/lib" -Wall -Wno-unused-function
mv -f ../libmachip.build.$$.so ../$OUT
echo "built $(cd .. && pwd)/$OUT"
