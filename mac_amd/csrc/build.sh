#!/bin/bash
# Build libmachip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
ROCM=${ROCM_PATH:-/opt/rocm}
"$ROCM/bin/hipcc" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC machip.hip \
    -o ../libmachip.so -L"$ROCM/lib" -lrccl -lrocsolver -lrocblas -ldl -Wl,-rpath,"$ROCM/lib" -Wall -Wno-unused-function
echo "built $(cd .. && pwd)/libmachip.so"
