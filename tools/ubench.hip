// tools/ubench.hip -- micro-benchmark of the fused Lanczos-step kernel (developer tool, not part of
// the library): back-to-back launches on synthetic random-graph Laplacians of the BASELINE sizes,
// for the launch shapes plan_pipe() chooses between.  Also times an empty kernel and a plain SpMV
// so the fixed launch cost and the price of the fused vector work can be read off.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o tools/bin/ubench && tools/bin/ubench
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../mac_amd/csrc/kernels.h"

namespace machip { thread_local std::string g_err; }
using namespace machip;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }

template <class F>
double time_us(F&& launch, int reps, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 1e3 * ms / reps;
}

template <int BLOCK, int G, int UNR, bool DED = false>
void run_pipe(const CsrView& A, PipeView L, int cap, hipStream_t s, long nnz) {
    const int gpb = (DED ? BLOCK - 64 : BLOCK) / G;
    const int grid = std::min(cap, (A.n + gpb - 1) / gpb);
    L.P = grid;
    const double us = time_us([&] { k_pipe_vec<BLOCK, G, UNR, DED><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, 0)); }, 400, s);
    printf("  k_pipe_vec blk=%-4d G=%-2d unr=%d ded=%d grid=%-4d : %7.2f us  (%6.0f GB/s on 12*nnz+56*n)\n", BLOCK, G, UNR, (int)DED, grid, us,
           (nnz * 12.0 + 56.0 * A.n) / us / 1e3);
}
template <int G>
void run_plain(const CsrView& A, const double* x, double* y, int cap, hipStream_t s, long nnz) {
    const int gpb = kBlock / G;
    const int grid = std::min(cap, (A.n + gpb - 1) / gpb);
    OpPlain op{y};
    const double us = time_us([&] { k_spmv_vec<G, OpPlain><<<grid, kBlock, 0, s>>>(A, x, op); }, 400, s);
    printf("  plain spmv (8 B gather)   G=%-2d grid=%-4d       : %7.2f us  (%6.0f GB/s on 12*nnz)\n", G, grid, us, nnz * 12.0 / us / 1e3);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int ns[] = {10000, 100000};
    for (int n : ns) {
        const int degs10k[] = {12, 52, 102}, degs100k[] = {6, 20, 40};
        for (int di = 0; di < 3; ++di) {
            const int deg = n == 10000 ? degs10k[di] : degs100k[di];
            std::mt19937_64 rng(42);
            std::vector<int> rowptr(n + 1, 0), col; std::vector<double> val;
            for (int r = 0; r < n; ++r) {
                std::vector<int> cs{r};
                for (int k = 0; k < deg; ++k) cs.push_back((int)(rng() % n));
                std::sort(cs.begin(), cs.end());
                for (int c : cs) { col.push_back(c); val.push_back(c == r ? (double)deg : -1.0); }
                rowptr[r + 1] = (int)col.size();
            }
            const long nnz = (long)col.size();
            int *drp, *dcol; double *dval, *x, *y;
            CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 8));
            CK(hipMemcpy(drp, rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dval, val.data(), nnz * 8, hipMemcpyHostToDevice));
            CsrView A{n, drp, dcol, dval};
            CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMemset(x, 0, n * 8));
            PipeView L{}; L.n = n;
            CK(hipMalloc(&L.st, sizeof(LanState))); CK(hipMemset(L.st, 0, sizeof(LanState)));
            CK(hipMalloc(&L.Z0, n * sizeof(Z2))); CK(hipMalloc(&L.Z1, n * sizeof(Z2))); CK(hipMemset(L.Z0, 0, n * sizeof(Z2)));
            CK(hipMalloc(&L.V, (size_t)n * 8 * 4)); CK(hipMalloc(&L.tri, 8 * 64));
            CK(hipMalloc(&L.part, 16 * kNP * kMaxGrid)); CK(hipMemset(L.part, 0, 16 * kNP * kMaxGrid));
            L.htri = nullptr; L.hflag = nullptr; L.P = 256;
            printf("== n=%d mean row %.1f nnz=%ld (%.2f MB csr)\n", n, (double)nnz / n, nnz, nnz * 12.0 / 1e6);
            printf("  empty kernel, 256 workgroups                  : %7.2f us\n", time_us([&] { k_empty<<<256, kBlock, 0, s>>>(nullptr); }, 400, s));
            run_plain<4>(A, x, y, 1024, s, nnz); run_plain<16>(A, x, y, 1024, s, nnz);
            for (int cap : {256}) {
                run_pipe<256, 4, 1>(A, L, cap, s, nnz); run_pipe<256, 4, 1, true>(A, L, cap, s, nnz);
                run_pipe<512, 4, 1>(A, L, cap, s, nnz); run_pipe<512, 4, 1, true>(A, L, cap, s, nnz);
                run_pipe<512, 8, 2>(A, L, cap, s, nnz); run_pipe<512, 8, 2, true>(A, L, cap, s, nnz);
                run_pipe<1024, 8, 2>(A, L, cap, s, nnz); run_pipe<1024, 8, 2, true>(A, L, cap, s, nnz);
                run_pipe<1024, 4, 1>(A, L, cap, s, nnz); run_pipe<1024, 4, 1, true>(A, L, cap, s, nnz);
                run_pipe<1024, 16, 2>(A, L, cap, s, nnz); run_pipe<1024, 16, 2, true>(A, L, cap, s, nnz);
            }
            CK(hipFree(drp)); CK(hipFree(dcol)); CK(hipFree(dval)); CK(hipFree(x)); CK(hipFree(y));
            CK(hipFree(L.st)); CK(hipFree(L.Z0)); CK(hipFree(L.Z1)); CK(hipFree(L.V)); CK(hipFree(L.tri)); CK(hipFree(L.part));
        }
    }
    return 0;
}
