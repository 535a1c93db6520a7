// tools/ubench_persist.hip -- developer micro-benchmark of the single-workgroup Lanczos kernel
// (mac_amd/csrc/persist.h) on a synthetic chain + closures Laplacian: time per step, and shader-clock
// stamps of the phases of one step (PERSIST_CLOCKS build flag).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPERSIST_CLOCKS tools/ubench_persist.hip -o tools/bin/ubench_persist
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>
#include "../mac_amd/csrc/persist.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1728, ncl = argc > 2 ? atoi(argv[2]) : 157;
    std::mt19937_64 rng(3);
    std::vector<std::map<int, double>> rows(n);
    auto add = [&](int i, int j, double w) { rows[i][j] -= w; rows[j][i] -= w; rows[i][i] += w; rows[j][j] += w; };
    for (int i = 0; i + 1 < n; ++i) add(i, i + 1, 100.0 + (rng() % 200));
    for (int c = 0; c < ncl; ++c) { int a = rng() % n, b = rng() % n; if (abs(a - b) > 1) add(a, b, 100.0 + (rng() % 100)); }
    std::vector<int> rp(n + 1, 0), col; std::vector<double> val;
    for (int r = 0; r < n; ++r) { for (auto& kv : rows[r]) { col.push_back(kv.first); val.push_back(kv.second); } rp[r + 1] = (int)col.size(); }
    const long nnz = (long)col.size();
    int *drp, *dcol; double* dval;
    CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, nnz * 4)); CK(hipMalloc(&dval, nnz * 8));
    CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcol, col.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dval, val.data(), nnz * 8, hipMemcpyHostToDevice));
    CsrView A{n, drp, dcol, dval};
    PersistPack P;
    CK(hipMalloc(&P.band, 5 * kPersistPad * 8)); CK(hipMalloc(&P.cc, 2 * kPersistPad * 4)); CK(hipMalloc(&P.crow, (kPersistPad + 1) * 4));
    CK(hipMalloc(&P.ccol, kPersistPackEntries * 4)); CK(hipMalloc(&P.cval, kPersistPackEntries * 8));
    PersistView L; L.n = n;
    const int cap = 4096;
    CK(hipMalloc(&L.st, sizeof(LanState))); CK(hipMalloc(&L.u, n * 8)); CK(hipMalloc(&L.vprev, n * 8)); CK(hipMalloc(&L.V, (size_t)n * cap * 8));
    CK(hipMalloc(&L.tri, 3 * (cap + 2) * 8)); CK(hipMalloc(&L.htri, 3 * (cap + 2) * 8)); CK(hipMalloc((void**)&L.hflag, 64));
    long long* dclk = nullptr; CK(hipMalloc(&dclk, 64 * 8)); CK(hipMemset(dclk, 0, 64 * 8));
#ifdef PERSIST_CLOCKS
    L.clk = dclk;
#endif
    std::vector<double> u0(n); for (auto& x : u0) x = (double)(rng() % 1000) / 500.0 - 1.0;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("n=%d nnz=%ld fits=%d\n", n, nnz, (int)persist_fits(n, nnz - n - 2L * (n - 1)));
    for (int steps : {16, 128, 256}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemcpyAsync(L.u, u0.data(), n * 8, hipMemcpyHostToDevice, s));
            k_persist_begin<<<8, 256, 0, s>>>(L, 1);
            k_persist_pack<<<1, 1024, 0, s>>>(A, P);
            CK(hipEventRecord(e0, s));
            switch ((n + 2 * kPersistThreads - 1) / (2 * kPersistThreads)) {   // rows per thread, rounded up to 2
                case 1: k_lan_persist<2><<<1, kPersistThreads, 0, s>>>(P, L, steps); break;
                case 2: k_lan_persist<4><<<1, kPersistThreads, 0, s>>>(P, L, steps); break;
                default: k_lan_persist<6><<<1, kPersistThreads, 0, s>>>(P, L, steps); break;
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("steps=%3d : %8.2f us total, %6.3f us/step\n", steps, 1e3 * ms, 1e3 * ms / steps);
        }
    }
    long long h[64]; CK(hipMemcpy(h, dclk, 64 * 8, hipMemcpyDeviceToHost));
    printf("clock stamps (shader clocks since kernel start): setup %lld | step0: R1 %lld norm %lld bar %lld spmv %lld R2 %lld upd %lld | step1 start %lld | end %lld (wall 100MHz ticks %lld)\n",
           h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[0], h[9] - h[0], h[11] - h[10]);
    return 0;
}
