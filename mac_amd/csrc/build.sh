#!/bin/bash
# Build libmachip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
ROCM=${ROCM_PATH:-/opt/rocm}
# (linked under a temporary name and renamed: a process that has the old library mapped keeps its own copy)
"$ROCM/bin/hipcc" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC machip.hip \
    -o ../libmachip.build.$$.so -L"$ROCM/lib" -lrccl -lrocsolver -lrocblas -ldl -Wl,-rpath,"$ROCM/lib" -Wall -Wno-unused-function
mv -f ../libmachip.build.$$.so ../libmachip.so
echo "built $(cd .. && pwd)/libmachip.so"
