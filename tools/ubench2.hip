// tools/ubench2.hip -- what bounds the CSR SpMV at config-4 size: the val/col stream or the gathers?
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../mac_amd/csrc/kernels.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: stream only (no gather)  1: gather 8 B  2: gather 16 B (Z2)  3: pure contiguous stream (grid-stride over nnz)
template <int G, int MODE, int UNR>
__global__ __launch_bounds__(1024) void k_t(CsrView A, const double* __restrict__ x, const Z2* __restrict__ Z, double* __restrict__ y) {
    if (MODE == 3) {
        double acc = 0.0;
        const long nnz = A.rowptr[A.n];
        for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += (long)gridDim.x * blockDim.x) acc += A.val[p] * (double)A.col[p];
        if (acc == 1.2345) y[0] = acc;
        return;
    }
    const int GPB = blockDim.x / G;
    const int lane = threadIdx.x % G, g = threadIdx.x / G;
    for (int r = blockIdx.x * GPB + g; r < A.n; r += gridDim.x * GPB) {
        const int b = A.rowptr[r], e = A.rowptr[r + 1];
        double s0 = 0.0, s1 = 0.0;
        int p = b + lane;
        for (; p + (UNR - 1) * G < e; p += UNR * G) {
            double vv[UNR]; int cc[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) { vv[q] = A.val[p + q * G]; cc[q] = A.col[p + q * G]; }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                if (MODE == 0) s0 += vv[q] * (double)cc[q];
                else if (MODE == 1) s0 += vv[q] * x[cc[q]];
                else { const Z2 z = Z[cc[q]]; s0 += vv[q] * z.t; s1 += vv[q] * z.v; }
            }
        }
        for (; p < e; p += G) {
            const double vv = A.val[p]; const int cc = A.col[p];
            if (MODE == 0) s0 += vv * (double)cc;
            else if (MODE == 1) s0 += vv * x[cc];
            else { const Z2 z = Z[cc]; s0 += vv * z.t; s1 += vv * z.v; }
        }
        s0 = group_sum<G>(s0 + s1);
        if (lane == 0) y[r] = s0;
    }
}
template <class F>
double time_us(F&& launch, int reps, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}
template <int G, int MODE, int UNR>
void run(const char* nm, CsrView A, double* x, Z2* Z, double* y, int block, int grid, hipStream_t s, long nnz) {
    const double us = time_us([&] { k_t<G, MODE, UNR><<<grid, block, 0, s>>>(A, x, Z, y); }, 200, s);
    printf("  %-28s G=%-2d unr=%d blk=%-4d grid=%-4d : %7.2f us  %6.0f GB/s(12B/nnz)\n", nm, G, UNR, block, grid, us, nnz * 12.0 / us / 1e3);
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 100000, deg = 40;
    std::mt19937_64 rng(42);
    std::vector<int> rowptr(n + 1, 0), col; std::vector<double> val;
    for (int r = 0; r < n; ++r) {
        std::vector<int> cs{r};
        for (int k = 0; k < deg; ++k) cs.push_back((int)(rng() % n));
        std::sort(cs.begin(), cs.end());
        for (int c : cs) { col.push_back(c); val.push_back(1.0); }
        rowptr[r + 1] = (int)col.size();
    }
    const long nnz = (long)col.size();
    int *drp, *dcol; double *dval, *x, *y; Z2* Z;
    CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 8));
    CK(hipMemcpy(drp, rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dval, val.data(), nnz * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&Z, n * 16)); CK(hipMemset(x, 0, n * 8)); CK(hipMemset(Z, 0, n * 16));
    CsrView A{n, drp, dcol, dval};
    printf("n=%d nnz=%ld (%.1f MB CSR)\n", n, nnz, nnz * 12.0 / 1e6);
    for (int grid : {256, 512, 1024, 2048}) {
        run<16, 3, 1>("contiguous stream", A, x, Z, y, 1024, grid, s, nnz);
        run<16, 3, 1>("contiguous stream", A, x, Z, y, 256, grid * 4, s, nnz);
    }
    for (int grid : {256, 512, 2048}) for (int block : {256, 1024}) {
        run<16, 0, 2>("row stream, no gather", A, x, Z, y, block, grid, s, nnz);
        run<16, 1, 2>("8 B gather", A, x, Z, y, block, grid, s, nnz);
        run<16, 2, 2>("16 B gather", A, x, Z, y, block, grid, s, nnz);
        run<8, 2, 2>("16 B gather", A, x, Z, y, block, grid, s, nnz);
        run<8, 2, 4>("16 B gather", A, x, Z, y, block, grid, s, nnz);
        run<64, 2, 1>("16 B gather", A, x, Z, y, block, grid, s, nnz);
    }
    return 0;
}
