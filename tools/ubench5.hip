// tools/ubench5.hip -- where does a fused Lanczos step spend its time?  k_pipe_vec compiled with PIPE_CLOCKS (100 MHz wall
// clock stamps per workgroup: entry, prologue done, first tile's loads+reduce done, barrier passed, last tile done, epilogue
// done), run as REAL consecutive steps (operand rewritten by the previous launch) on config-4-like matrices.
#define PIPE_CLOCKS 1
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../mac_amd/csrc/kernels.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BLOCK, int G, int UNR>
void run(const char* nm, CsrView A, PipeView L, hipStream_t s, long nnz, std::vector<double>& u0h) {
    const int gpb = (BLOCK - 64) / G;
    const int grid = std::min(256, (A.n + gpb - 1) / gpb);
    L.P = grid;
    double* u0; CK(hipMalloc(&u0, A.n * 8)); CK(hipMemcpy(u0, u0h.data(), A.n * 8, hipMemcpyHostToDevice));
    k_pipe_init<<<grid, kBlock, 0, s>>>(L, u0, 1);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int steps = 40;
    for (int j = 0; j < 8; ++j) k_pipe_vec<BLOCK, G, UNR, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, j));   // jA stays 0: j = jrel (V columns 0..47 exist)
    CK(hipEventRecord(e0, s));
    for (int j = 8; j < 8 + steps; ++j) k_pipe_vec<BLOCK, G, UNR, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, j));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    std::vector<long long> c((size_t)grid * 8);
    CK(hipMemcpy(c.data(), L.clk, c.size() * 8, hipMemcpyDeviceToHost));
    long long t0 = c[0];
    for (int b = 0; b < grid; ++b) t0 = std::min(t0, c[(size_t)b * 8]);
    auto stat = [&](int i, const char* what) {
        double mn = 1e30, mx = 0, av = 0;
        for (int b = 0; b < grid; ++b) { const double v = (c[(size_t)b * 8 + i] - t0) * 0.01; mn = std::min(mn, v); mx = std::max(mx, v); av += v; }
        printf("      %-46s min %6.2f  mean %6.2f  max %6.2f us after the first workgroup's entry\n", what, mn, av / grid, mx);
    };
    printf("   %s blk=%d G=%d unr=%d grid=%d: %.2f us per step back-to-back (nnz %ld)\n", nm, BLOCK, G, UNR, grid, 1e3 * ms / steps, nnz);
    stat(0, "workgroup entry (wave 0)"); stat(2, "worker wave entry"); stat(7, "prologue: partials loaded (wave 0)"); stat(1, "prologue done (wave 0)"); stat(3, "prologue: partial loads issued (wave 0)");
    stat(4, "barrier passed"); stat(5, "last tile finished"); stat(6, "epilogue (partials stored)");
    CK(hipFree(u0));
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (double deg : {6.0, 26.0, 40.0}) {
        const int n = 100000;
        std::mt19937_64 rng(7);
        std::vector<std::vector<int>> adj((size_t)n);
        for (int i = 0; i + 1 < n; ++i) { adj[i].push_back(i + 1); adj[i + 1].push_back(i); }
        for (long k = 0; k < (long)(deg * n / 2); ++k) { int a = (int)(rng() % n), b = (int)(rng() % n); if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } }
        std::vector<int> rp(n + 1, 0), col; std::vector<double> val;
        for (int r = 0; r < n; ++r) {
            auto& v = adj[r]; std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
            col.push_back(r); val.push_back((double)v.size());            // diagonal first, like the assembled L(x)
            for (int c : v) { col.push_back(c); val.push_back(-1.0); }
            rp[r + 1] = (int)col.size();
        }
        const long nnz = (long)col.size();
        int *drp, *dcol; double* dval;
        CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 8));
        CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dval, val.data(), nnz * 8, hipMemcpyHostToDevice));
        CsrView A{n, drp, dcol, dval};
        PipeView L{}; L.n = n;
        CK(hipMalloc(&L.st, sizeof(LanState))); CK(hipMemset(L.st, 0, sizeof(LanState)));
        CK(hipMalloc(&L.Z0, n * sizeof(Z2))); CK(hipMalloc(&L.Z1, n * sizeof(Z2)));
        CK(hipMalloc(&L.V, (size_t)n * 8 * 64)); CK(hipMalloc(&L.tri, 8 * 3 * 80));
        CK(hipMalloc(&L.part, 16 * kNP * kMaxGrid)); CK(hipMemset(L.part, 0, 16 * kNP * kMaxGrid));
        CK(hipMalloc(&L.clk, 8 * 8 * kMaxGrid)); CK(hipMemset(L.clk, 0, 8 * 8 * kMaxGrid));
        L.htri = nullptr; L.hflag = nullptr; L.P = 256;
        std::vector<double> u0((size_t)n); for (int i = 0; i < n; ++i) u0[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
        printf("== n=%d mean row %.1f nnz=%ld\n", n, (double)nnz / n, nnz);
        run<1024, 16, 2>("current", A, L, s, nnz, u0);
        run<1024, 8, 2>("current", A, L, s, nnz, u0);
        run<1024, 4, 2>("current", A, L, s, nnz, u0);
        run<512, 8, 4>("small wg", A, L, s, nnz, u0);
        CK(hipFree(drp)); CK(hipFree(dcol)); CK(hipFree(dval));
    }
    return 0;
}
