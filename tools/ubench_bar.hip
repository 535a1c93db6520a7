// tools/ubench_bar.hip -- developer micro-benchmark: what does a software grid barrier cost on MI355X
// (256 co-resident workgroups), compared with a kernel boundary?  Decides whether a persistent
// multi-step Lanczos kernel could beat one launch per step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_bar.hip -o tools/bin/ubench_bar
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// monotone counter barrier; returns false if it timed out (never hang the box)
__device__ __forceinline__ bool grid_bar(unsigned* ctr, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000) { *err = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(1024) k_bar_only(unsigned* ctr, int steps, int* err) {
    const unsigned P = gridDim.x;
    for (int s = 1; s <= steps; ++s) if (!grid_bar(ctr, P * s, err)) return;
}
__global__ void __launch_bounds__(1024) k_cg_only(int steps) {
    cg::grid_group g = cg::this_grid();
    for (int s = 0; s < steps; ++s) g.sync();
}
// barrier + exchange: every workgroup writes 6 doubles, after the barrier every workgroup reads all P*6 and checks
__global__ void __launch_bounds__(1024) k_bar_xchg(unsigned* ctr, double* part, int steps, int* err, int* bad) {
    const unsigned P = gridDim.x;
    for (int s = 1; s <= steps; ++s) {
        double* buf = part + (size_t)(s & 1) * P * 6;
        if (threadIdx.x < 6) buf[blockIdx.x * 6 + threadIdx.x] = (double)(s * 1000 + blockIdx.x);
        if (!grid_bar(ctr, P * s, err)) return;
        if (threadIdx.x < P) {
            double v = __builtin_nontemporal_load(&buf[threadIdx.x * 6 + 3]);
            if (v != (double)(s * 1000 + threadIdx.x)) atomicAdd(bad, 1);
        }
    }
}
// barrier + vector exchange: n doubles written in slices, after the barrier everyone gathers `g` random entries per thread
__global__ void __launch_bounds__(1024) k_bar_vec(unsigned* ctr, double* vec, const int* idx, int n, int g, int steps, int* err, double* sink) {
    const unsigned P = gridDim.x;
    double acc = 0;
    for (int s = 1; s <= steps; ++s) {
        double* buf = vec + (size_t)(s & 1) * n;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += P * blockDim.x) buf[i] = (double)(s + i);
        if (!grid_bar(ctr, P * s, err)) return;
        const int* my = idx + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * g;
        for (int q = 0; q < g; ++q) acc += buf[my[q]];
    }
    if (acc == 1.2345) *sink = acc;
}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* ctr; int *err, *bad; double* part; double* sink;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&part, 2 * 256 * 6 * 8)); CK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int nb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bar_only, 1024, 0));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("CUs %d, max active 1024-thread blocks/CU %d, cooperativeLaunch %d\n", pr.multiProcessorCount, nb, pr.cooperativeLaunch);
    const int steps = 2000;
    for (int P : {64, 128, 256}) for (int blk : {256, 1024}) {
        CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
        void* args[] = {&ctr, (void*)&steps, &err};
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel((void*)k_bar_only, dim3(P), dim3(blk), args, 0, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); int h; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
        printf("atomic barrier  P=%3d blk=%4d : %6.2f us/barrier  err=%d\n", P, blk, 1e3 * ms / steps, h);
        void* a2[] = {(void*)&steps};
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel((void*)k_cg_only, dim3(P), dim3(blk), a2, 0, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("cg grid.sync    P=%3d blk=%4d : %6.2f us/barrier\n", P, blk, 1e3 * ms / steps);
    }
    {
        int P = 256;
        CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st));
        void* args[] = {&ctr, &part, (void*)&steps, &err, &bad};
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel((void*)k_bar_xchg, dim3(P), dim3(1024), args, 0, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); int h, b; CK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
        printf("barrier + partial exchange P=256 : %6.2f us/step  err=%d stale_reads=%d\n", 1e3 * ms / steps, h, b);
    }
    for (int n : {10000, 100000}) for (int g : {2, 8}) {
        int P = 256; double* vec; int* idx; CK(hipMalloc(&vec, 2 * (size_t)n * 8)); CK(hipMalloc(&idx, (size_t)P * 1024 * g * 4));
        std::vector<int> h((size_t)P * 1024 * g); for (auto& x : h) x = rand() % n;
        CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
        void* args[] = {&ctr, &vec, &idx, &n, &g, (void*)&steps, &err, &sink};
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel((void*)k_bar_vec, dim3(P), dim3(1024), args, 0, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); int e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("barrier + write n=%6d + %d gathers/thread (%.0fk gathers) : %6.2f us/step err=%d\n", n, g, P * 1024.0 * g / 1e3, 1e3 * ms / steps, e);
        CK(hipFree(vec)); CK(hipFree(idx));
    }
    return 0;
}
