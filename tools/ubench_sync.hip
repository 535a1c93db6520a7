// How long after a kernel's end does the host know?  (A) hipStreamSynchronize, (B) hipEventSynchronize, (C) spinning on a word in
// pinned host memory that a trailing one-thread kernel writes, (D) that word written by the work kernel's own last lines.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_sync.hip -o /tmp/ubench_sync && /tmp/ubench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_work(double* a, int n, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double x = a[i % n];
    for (int r = 0; r < reps; ++r) x = x * 1.0000001 + 1e-9;
    a[i % n] = x;
}
__global__ void k_flag(volatile unsigned long long* f, unsigned long long v) { if (threadIdx.x == 0) *f = v; }
__global__ void k_work_flag(double* a, int n, int reps, volatile unsigned long long* f, unsigned long long v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double x = a[i % n];
    for (int r = 0; r < reps; ++r) x = x * 1.0000001 + 1e-9;
    a[i % n] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *f = v;      // one workgroup: the "whole job" is this workgroup
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* a; hipMalloc(&a, 8 << 20);
    hipMemset(a, 0, 8 << 20);
    unsigned long long* hf; hipHostMalloc((void**)&hf, 64, hipHostMallocMapped); *hf = 0;
    unsigned long long* df; hipHostGetDevicePointer((void**)&df, hf, 0);
    hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int N = 2000;
    for (int grid : {1, 256}) {
        for (int reps : {100, 20000}) {
            auto run = [&](int mode) {
                unsigned long long seq = *hf;
                auto t0 = std::chrono::steady_clock::now();
                for (int it = 0; it < N; ++it) {
                    ++seq;
                    if (mode == 3) k_work_flag<<<grid, 256, 0, s>>>(a, 1 << 20, reps, df, seq);
                    else k_work<<<grid, 256, 0, s>>>(a, 1 << 20, reps);
                    if (mode == 0) hipStreamSynchronize(s);
                    else if (mode == 1) { hipEventRecord(ev, s); hipEventSynchronize(ev); }
                    else { if (mode == 2) k_flag<<<1, 64, 0, s>>>(df, seq); while (*(volatile unsigned long long*)hf != seq) __builtin_ia32_pause(); }
                }
                hipStreamSynchronize(s);
                return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            };
            run(0);
            printf("grid %3d reps %5d: launch+wait per iteration: streamSync %.2f us  eventSync %.2f us  flag kernel + spin %.2f us  in-kernel flag + spin %.2f us\n",
                   grid, reps, run(0), run(1), run(2), grid == 1 ? run(3) : 0.0);
        }
    }
    return 0;
}
