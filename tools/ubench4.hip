// tools/ubench4.hip -- does the cache policy of the scattered 16-byte gathers change their cost?  (plain / nt / sc1 / sc0 sc1)
// Row-group SpMV of tools/ubench3.hip with the operand load issued by inline asm; operand rewritten between launches by a
// second kernel so that it is L2-cold at kernel start like in a real Lanczos step.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../mac_amd/csrc/kernels.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int POL>
__device__ __forceinline__ v2d gload(const Z2* p) {
    v2d r;
    if (POL == 0) { const Z2 z = *p; r.x = z.t; r.y = z.v; return r; }
    if (POL == 1) return __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// two loads in flight, then one wait (the asm forms above wait per load, which would handicap them)
template <int POL>
__device__ __forceinline__ void gload2(const Z2* p0, const Z2* p1, v2d& r0, v2d& r1) {
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(p0), "v"(p1) : "memory");
    else if (POL == 3) asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(p0), "v"(p1) : "memory");
    else if (POL == 4) asm volatile("global_load_dwordx4 %0, %2, off sc0\n\tglobal_load_dwordx4 %1, %3, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(r0), "=&v"(r1) : "v"(p0), "v"(p1) : "memory");
    else { r0 = gload<POL>(p0); r1 = gload<POL>(p1); }
}

template <int G, int UNR, int POL>
__global__ __launch_bounds__(1024) void k_rowgroup(CsrView A, const Z2* __restrict__ Z, double* __restrict__ y) {
    const int GPB = blockDim.x / G;
    const int lane = threadIdx.x % G, g = threadIdx.x / G;
    for (int r = blockIdx.x * GPB + g; r < A.n; r += gridDim.x * GPB) {
        const int b = A.rowptr[r], e = A.rowptr[r + 1];
        double s0 = 0.0, s1 = 0.0;
        int p = b + lane;
        for (; p + (UNR - 1) * G < e; p += UNR * G) {
            double vv[UNR]; int cc[UNR]; v2d zz[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) { vv[q] = A.val[p + q * G]; cc[q] = A.col[p + q * G]; }
            if (UNR == 2) gload2<POL>(Z + cc[0], Z + cc[1], zz[0], zz[1]);
            else {
#pragma unroll
                for (int q = 0; q < UNR; ++q) zz[q] = gload<POL>(Z + cc[q]);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) { s0 += vv[q] * zz[q].x; s1 += vv[q] * zz[q].y; }
        }
        for (; p < e; p += G) { const double vv = A.val[p]; const v2d z = gload<POL>(Z + A.col[p]); s0 += vv * z.x; s1 += vv * z.y; }
        s0 = group_sum<G>(s0); s1 = group_sum<G>(s1);
        if (lane == 0) { y[2 * r] = s0; y[2 * r + 1] = s1; }
    }
}
__global__ void k_touch(Z2* Z, int n, double eps) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { Z2 z = Z[i]; z.t += eps; Z[i] = z; }
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (double deg : {6.0, 26.0, 40.0}) {
        const int n = 100000;
        std::mt19937_64 rng(7);
        std::vector<std::vector<int>> adj((size_t)n);
        for (int i = 0; i + 1 < n; ++i) { adj[i].push_back(i + 1); adj[i + 1].push_back(i); }
        for (long k = 0; k < (long)(deg * n / 2); ++k) { int a = (int)(rng() % n), b = (int)(rng() % n); if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } }
        std::vector<int> rp(n + 1, 0), col; 
        for (int r = 0; r < n; ++r) { auto& v = adj[r]; v.push_back(r); std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); for (int c : v) col.push_back(c); rp[r + 1] = (int)col.size(); }
        const long nnz = (long)col.size();
        std::vector<double> hv((size_t)nnz, 1.0);
        int *drp, *dcol; double *dval, *y; Z2* Z;
        CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 8)); CK(hipMalloc(&y, (size_t)n * 16)); CK(hipMalloc(&Z, (size_t)n * 16));
        CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dval, hv.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemset(Z, 0, (size_t)n * 16));
        CsrView A{n, drp, dcol, dval};
        printf("== n=%d nnz=%ld mean %.1f\n", n, nnz, (double)nnz / n);
        auto timeit = [&](auto&& launch, bool cold) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 3; ++i) launch();
            double tot = 0; const int reps = 50;
            for (int i = 0; i < reps; ++i) {
                if (cold) k_touch<<<256, 256, 0, s>>>(Z, n, 1e-9);
                CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms;
            }
            CK(hipGetLastError());
            return 1e3 * tot / reps;
        };
#define RUN(G, U, POL, NM) { double w = timeit([&] { k_rowgroup<G, U, POL><<<256, 1024, 0, s>>>(A, Z, y); }, false); double c = timeit([&] { k_rowgroup<G, U, POL><<<256, 1024, 0, s>>>(A, Z, y); }, true); \
        printf("   G=%-2d unr=%d %-10s warm %7.2f us   operand rewritten before each launch %7.2f us\n", G, U, NM, w, c); fflush(stdout); }
        RUN(16, 2, 0, "plain") RUN(16, 2, 1, "nt") RUN(16, 2, 2, "sc1") RUN(16, 2, 3, "sc0 sc1") RUN(16, 2, 4, "sc0")
        RUN(8, 2, 0, "plain") RUN(8, 2, 1, "nt") RUN(4, 2, 0, "plain") RUN(4, 2, 1, "nt")
        CK(hipFree(drp)); CK(hipFree(dcol)); CK(hipFree(dval)); CK(hipFree(y)); CK(hipFree(Z));
    }
    return 0;
}
