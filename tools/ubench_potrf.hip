// tools/ubench_potrf.hip -- developer micro-benchmark: rocSOLVER Cholesky factorisation + inverse of an s x s SPD matrix
// (the capacitance matrix of a Woodbury-style exact chain + closures preconditioner), fp64, MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_potrf.hip -o tools/bin/ubench_potrf -lrocsolver -lrocblas
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    rocblas_handle h;
    auto t0 = std::chrono::steady_clock::now();
    rocblas_create_handle(&h);
    hipStream_t st; CK(hipStreamCreate(&st)); rocblas_set_stream(h, st);
    printf("handle creation %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    for (int s : {64, 160, 400, 800, 1600, 2400}) {
        std::vector<double> A((size_t)s * s);
        for (int i = 0; i < s; ++i) for (int j = 0; j < s; ++j) A[(size_t)i * s + j] = (i == j ? s + 1.0 : 1.0 / (1.0 + abs(i - j)));
        double* d; int* info; CK(hipMalloc(&d, sizeof(double) * s * s)); CK(hipMalloc(&info, 4));
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        float best_f = 1e9f, best_i = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemcpy(d, A.data(), sizeof(double) * s * s, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, st));
            rocsolver_dpotrf(h, rocblas_fill_lower, s, d, s, info);
            CK(hipEventRecord(e1, st));
            rocsolver_dpotri(h, rocblas_fill_lower, s, d, s, info);
            CK(hipEventRecord(e2, st)); CK(hipEventSynchronize(e2));
            float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
            if (rep) { best_f = a < best_f ? a : best_f; best_i = b < best_i ? b : best_i; }
        }
        printf("s=%5d  potrf %8.3f ms   potri %8.3f ms\n", s, best_f, best_i);
        CK(hipFree(d)); CK(hipFree(info));
    }
    return 0;
}
