// tools/ubench.hip -- ablation micro-benchmark for the fused Lanczos-step kernel (developer tool,
// not part of the library).  Times back-to-back launches of kernel variants on a synthetic
// random-graph Laplacian so the cost of each ingredient (launch, prologue reduction, epilogue
// reduction, gather width, unrolling, grid size) can be read off.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
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

// MODE 0: x[c] gather (8 B); 1: Z3 gather + on-the-fly combination (1 accumulator);
// MODE 2: Z3 gather, 3 raw accumulators combined at the end.
template <int G, int MODE, bool PRO, bool EPI, int UNR>
__global__ __launch_bounds__(kBlock) void k_var(CsrView A, PipeView L, const double* __restrict__ x,
                                                double* __restrict__ y, int jrel) {
    __shared__ double smw[kMaxWaves * kNP];
    constexpr int GPB = kBlock / G;
    const int lane = threadIdx.x % G, g = threadIdx.x / G;
    PipeCoef c; c.alpha = 0.3; c.betap = 0.2; c.mu = 0.0; c.beta = 1.0; c.inv = 1.0; c.l1prev = 0;
    int jA = 0;
    __shared__ double scoef[8];
    if (PRO) {
        if (threadIdx.x < 64) { int jd; (void)pipe_prologue_wave0(L, jrel, -1, scoef, &jd); }
        __syncthreads();
        jA = (int)scoef[4];
    }
    const Z3* __restrict__ Zc = L.Z0;
    Z3* __restrict__ Zn = L.Z1;
    double* vj = L.V + (size_t)(jA + jrel) * L.n;
    PipeRow pr; pr.clear();
    for (int r0 = blockIdx.x * GPB; r0 < A.n; r0 += gridDim.x * GPB) {
        const int r = r0 + g;
        if (r >= A.n) continue;
        const int b = A.rowptr[r], e = A.rowptr[r + 1];
        double sw = 0.0, s1 = 0.0, s2 = 0.0;
        int p = b + lane;
        if (UNR > 1) {
            for (; p + (UNR - 1) * G < e; p += UNR * G) {
                double vv[UNR]; int cc[UNR];
#pragma unroll
                for (int q = 0; q < UNR; ++q) { vv[q] = A.val[p + q * G]; cc[q] = A.col[p + q * G]; }
                if (MODE == 0) {
                    double xx[UNR];
#pragma unroll
                    for (int q = 0; q < UNR; ++q) xx[q] = x[cc[q]];
#pragma unroll
                    for (int q = 0; q < UNR; ++q) sw += vv[q] * xx[q];
                } else {
                    Z3 zz[UNR];
#pragma unroll
                    for (int q = 0; q < UNR; ++q) zz[q] = Zc[cc[q]];
#pragma unroll
                    for (int q = 0; q < UNR; ++q) {
                        if (MODE == 1) sw += vv[q] * ((((zz[q].w - c.alpha * zz[q].v1) - c.betap * zz[q].v2) - c.mu) * c.inv);
                        else { sw += vv[q] * zz[q].w; s1 += vv[q] * zz[q].v1; s2 += vv[q] * zz[q].v2; }
                    }
                }
            }
        }
        for (; p < e; p += G) {
            const double vv = A.val[p];
            const int cc = A.col[p];
            if (MODE == 0) sw += vv * x[cc];
            else {
                const Z3 z = Zc[cc];
                if (MODE == 1) sw += vv * ((((z.w - c.alpha * z.v1) - c.betap * z.v2) - c.mu) * c.inv);
                else { sw += vv * z.w; s1 += vv * z.v1; s2 += vv * z.v2; }
            }
        }
        sw = group_sum<G>(sw);
        if (MODE == 2) { s1 = group_sum<G>(s1); s2 = group_sum<G>(s2); }
        if (lane == 0) {
            if (MODE == 0) y[r] = sw;
            else {
                const Z3 z = Zc[r];
                if (MODE == 1) { s1 = 0; s2 = 0; }
                pr.finish(c.alpha, c.betap, c.mu, c.inv, z, sw, s1, s2, vj, Zn, r);
            }
        }
    }
    if (EPI) pr.template store<kBlock>(L, jrel, smw);
}

struct Dev {
    CsrView A; PipeView L; double *x, *y;
};

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

template <int G, int MODE, bool PRO, bool EPI, int UNR>
void run(const char* name, const Dev& d, int grid_cap, hipStream_t s, long nnz) {
    constexpr int GPB = kBlock / G;
    const int grid = std::min(grid_cap, (d.A.n + GPB - 1) / GPB);
    PipeView L = d.L; L.P = grid;
    const double us = time_us([&] { k_var<G, MODE, PRO, EPI, UNR><<<grid, kBlock, 0, s>>>(d.A, L, d.x, d.y, 0); }, 400, s);
    printf("  %-34s G=%-2d grid=%-4d unr=%d : %7.2f us  (%6.0f GB/s on nnz*12)\n", name, G, grid, UNR, us, nnz * 12.0 / us / 1e3);
}

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int ns[] = {10000, 100000};
    for (int n : ns) {
        const int degs10k[] = {12, 52, 102}, degs100k[] = {6, 20, 40};
        for (int di = 0; di < 3; ++di) {
            const int deg = n == 10000 ? degs10k[di] : degs100k[di];
            // random symmetric-ish pattern: each row gets `deg` random columns + diagonal (sorted)
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
            Dev d;
            int *drp, *dcol; double* dval;
            CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 8));
            CK(hipMemcpy(drp, rowptr.data(), (n + 1) * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dval, val.data(), nnz * 8, hipMemcpyHostToDevice));
            d.A = CsrView{n, drp, dcol, dval};
            CK(hipMalloc(&d.x, n * 8)); CK(hipMalloc(&d.y, n * 8));
            CK(hipMemset(d.x, 0, n * 8));
            PipeView L; L.n = n;
            CK(hipMalloc(&L.st, sizeof(LanState))); CK(hipMemset(L.st, 0, sizeof(LanState)));
            CK(hipMalloc(&L.Z0, n * sizeof(Z3))); CK(hipMalloc(&L.Z1, n * sizeof(Z3)));
            CK(hipMemset(L.Z0, 0, n * sizeof(Z3)));
            CK(hipMalloc(&L.V, (size_t)n * 8 * 4)); CK(hipMalloc(&L.tri, 8 * 64)); CK(hipMalloc(&L.cb, 8 * (kMaxChunk + 2)));
            CK(hipMemset(L.cb, 0, 8 * (kMaxChunk + 2)));
            CK(hipMalloc(&L.part, 16 * kNP * kMaxGrid)); CK(hipMemset(L.part, 0, 16 * kNP * kMaxGrid));
            L.P = 256;
            d.L = L;
            printf("== n=%d mean row %.1f nnz=%ld (%.2f MB csr)\n", n, (double)nnz / n, nnz, nnz * 12.0 / 1e6);
            printf("  %-34s                    : %7.2f us\n", "empty kernel x256 blocks",
                   time_us([&] { k_empty<<<256, kBlock, 0, s>>>(nullptr); }, 400, s));
            const int caps[] = {256};
            for (int cap : caps) {
                run<4, 0, false, false, 1>("spmv 8B", d, cap, s, nnz);
                run<16, 0, false, false, 1>("spmv 8B", d, cap, s, nnz);
                run<4, 0, false, false, 4>("spmv 8B unrolled", d, cap, s, nnz);
                run<16, 0, false, false, 4>("spmv 8B unrolled", d, cap, s, nnz);
                run<4, 1, false, false, 1>("spmv Z3 1acc", d, cap, s, nnz);
                run<16, 1, false, false, 1>("spmv Z3 1acc", d, cap, s, nnz);
                run<4, 1, false, false, 4>("spmv Z3 1acc unrolled", d, cap, s, nnz);
                run<16, 1, false, false, 4>("spmv Z3 1acc unrolled", d, cap, s, nnz);
                run<4, 1, true, false, 1>("Z3 1acc + prologue", d, cap, s, nnz);
                run<4, 1, false, true, 1>("Z3 1acc + epilogue", d, cap, s, nnz);
                run<4, 1, true, true, 1>("Z3 1acc + pro + epi (fused v1)", d, cap, s, nnz);
                run<16, 1, true, true, 1>("Z3 1acc + pro + epi (fused v1)", d, cap, s, nnz);
                run<4, 1, true, true, 4>("fused v1 unrolled", d, cap, s, nnz);
                run<16, 1, true, true, 4>("fused v1 unrolled", d, cap, s, nnz);
                run<4, 2, true, true, 1>("Z3 3acc + pro + epi", d, cap, s, nnz);
                {
                    for (int blk : {256, 1024}) for (int G : {4, 16}) {
                        const int gpb = blk / G;
                        const int grid = std::min(cap, (n + gpb - 1) / gpb);
                        if (grid > 256) continue;
                        PipeView L2 = d.L; L2.P = grid;
                        auto f = [&] {
                            if (blk == 256 && G == 4) k_pipe_vec<256, 4><<<grid, 256, 0, s>>>(d.A, L2, 0);
                            else if (blk == 256) k_pipe_vec<256, 16><<<grid, 256, 0, s>>>(d.A, L2, 0);
                            else if (G == 4) k_pipe_vec<1024, 4><<<grid, 1024, 0, s>>>(d.A, L2, 0);
                            else k_pipe_vec<1024, 16><<<grid, 1024, 0, s>>>(d.A, L2, 0);
                        };
                        const double us = time_us(f, 400, s);
                        printf("  %-34s G=%-2d grid=%-4d blk=%d : %7.2f us  (%6.0f GB/s on nnz*12)\n", "PRODUCTION k_pipe_vec", G, grid, blk, us, nnz * 12.0 / us / 1e3);
                    }
                }
            }
            CK(hipFree(drp)); CK(hipFree(dcol)); CK(hipFree(dval)); CK(hipFree(d.x)); CK(hipFree(d.y));
            CK(hipFree(L.st)); CK(hipFree(L.Z0)); CK(hipFree(L.Z1)); CK(hipFree(L.V)); CK(hipFree(L.tri)); CK(hipFree(L.cb)); CK(hipFree(L.part));
        }
    }
    return 0;
}
