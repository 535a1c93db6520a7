#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__device__ __forceinline__ long long wc() { return (long long)__builtin_readcyclecounter(); }
__device__ __forceinline__ long long wall() { long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__global__ void kA(long long* out, int spin_us) { long long t0 = wall(); if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0; while (wall() - t0 < spin_us * 100LL) { __builtin_amdgcn_s_sleep(8); } if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = wall(); }
__global__ void kB(long long* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[2] = wall(); }
int main() {
    long long* d; hipMalloc(&d, 64); long long h[4];
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemsetAsync(d, 0, 64, s);
            hipLaunchKernelGGL(kA, dim3(64), dim3(256), 0, s, d, 50);
            if (mode == 0) hipLaunchKernelGGL(kB, dim3(64), dim3(256), 0, s, d);
            else hipExtLaunchKernelGGL(kB, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d);
            hipStreamSynchronize(s);
            hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
            printf("mode %d: A start 0, A end %.2f us, B start %.2f us  (%s)\n", mode, (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
