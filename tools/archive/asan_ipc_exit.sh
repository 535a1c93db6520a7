cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/asan
python - <<'PY'
import sys
sys.path.insert(0,"tests"); sys.path.insert(0,".")
import test_gpu_parity as t
open("/tmp/ipc_worker.py","w").write(t.IPC_WORKER)
PY
ASAN=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
for i in 1 2 3; do
RANK=0 WORLD_SIZE=1 MACHIP_RDZV_KEY=k$i HSA_ENABLE_IPC_MODE_LEGACY=0 LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 MACHIP_LIB=$PWD/mac_amd/libmachip_asan.so timeout 200 python /tmp/ipc_worker.py c4 3 -1 '{}' > gpurun_out/asan/w$i.out 2> gpurun_out/asan/w$i.err; echo "rc=$?"; head -c 300 gpurun_out/asan/w$i.out | cut -c1-100; grep -n "ERROR\|SUMMARY\|#0 \|#1 \|#2 \|#3 \|#4 \|#5 \|#6 \|#7 \|#8 \|#9 " gpurun_out/asan/w$i.err | cut -c1-220 | head -16
done
