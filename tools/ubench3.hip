// tools/ubench3.hip -- round 2: row-group SpMV (the shipped mapping) against a wave-tile stream mapping, on
// matrices shaped like the config-4 Frank-Wolfe iterates (chain + Poisson degrees, optionally hub-concentrated).
//   row-group: G lanes own a row: rowptr -> col/val -> gather is one dependent chain per row, lanes idle on short rows
//   wave-tile: a wave owns R consecutive rows, streams their nnz in coalesced chunks of 64*UNR through wave-private
//              LDS (products), then lane = row sums its LDS segment: no dependence of the stream on rowptr,
//              perfect coalescing, balanced by construction inside a tile
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../mac_amd/csrc/kernels.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int G, int UNR>
__global__ __launch_bounds__(1024) void k_rowgroup(CsrView A, const Z2* __restrict__ Z, double* __restrict__ y) {
    const int GPB = blockDim.x / G;
    const int lane = threadIdx.x % G, g = threadIdx.x / G;
    for (int r = blockIdx.x * GPB + g; r < A.n; r += gridDim.x * GPB) {
        const int b = A.rowptr[r], e = A.rowptr[r + 1];
        double s0 = 0.0, s1 = 0.0;
        int p = b + lane;
        for (; p + (UNR - 1) * G < e; p += UNR * G) {
            double vv[UNR]; int cc[UNR]; Z2 zz[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) { vv[q] = A.val[p + q * G]; cc[q] = A.col[p + q * G]; }
#pragma unroll
            for (int q = 0; q < UNR; ++q) zz[q] = Z[cc[q]];
#pragma unroll
            for (int q = 0; q < UNR; ++q) { s0 += vv[q] * zz[q].t; s1 += vv[q] * zz[q].v; }
        }
        for (; p < e; p += G) { const double vv = A.val[p]; const Z2 z = Z[A.col[p]]; s0 += vv * z.t; s1 += vv * z.v; }
        s0 = group_sum<G>(s0); s1 = group_sum<G>(s1);
        if (lane == 0) { y[2 * r] = s0; y[2 * r + 1] = s1; }
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// R rows per wave tile, LPR = 64 / R lanes per row in the reduce phase, chunks of 64*UNR nnz.
template <int R, int UNR, int NW>
__global__ __launch_bounds__(NW * 64) void k_wavetile(CsrView A, const Z2* __restrict__ Z, double* __restrict__ y) {
    constexpr int C = 64 * UNR;
    constexpr int LPR = 64 / R;
    __shared__ double lds[NW][2][C];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* pt = lds[wv][0];
    double* pv = lds[wv][1];
    const int ntiles = (A.n + R - 1) / R;
    const int row_l = lane / LPR, sub = lane % LPR;
    for (int tile = blockIdx.x * NW + wv; tile < ntiles; tile += gridDim.x * NW) {
        const int r0 = tile * R;
        const int r = r0 + row_l;
        int b = 0, e = 0;
        if (r < A.n) { b = A.rowptr[r]; e = A.rowptr[r + 1]; }
        const int q0 = __builtin_amdgcn_readfirstlane(b);
        const int rl = min(R, A.n - r0) - 1;
        const int q1 = __builtin_amdgcn_readlane(e, rl * LPR);
        double s0 = 0.0, s1 = 0.0;
        for (int base = q0; base < q1; base += C) {
            double vv[UNR]; int cc[UNR]; Z2 zz[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * 64 + lane;
                const bool ok = idx < q1;
                vv[u] = ok ? A.val[idx] : 0.0; cc[u] = ok ? A.col[idx] : 0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) zz[u] = Z[cc[u]];
#pragma unroll
            for (int u = 0; u < UNR; ++u) { pt[u * 64 + lane] = vv[u] * zz[u].t; pv[u * 64 + lane] = vv[u] * zz[u].v; }
            wave_lds_sync();
            const int lo = max(b, base) - base, hi = min(e, base + C) - base;
            for (int i = lo + sub; i < hi; i += LPR) { s0 += pt[i]; s1 += pv[i]; }
            wave_lds_sync();
        }
        if (LPR > 1) { s0 = group_sum<LPR>(s0); s1 = group_sum<LPR>(s1); }
        if (r < A.n && sub == 0) { y[2 * r] = s0; y[2 * r + 1] = s1; }
    }
}

// Same, software-pipelined: col/val of chunk k+1 are requested before the gathers of chunk k are consumed.
template <int R, int UNR, int NW>
__global__ __launch_bounds__(NW * 64) void k_wavetile_pf(CsrView A, const Z2* __restrict__ Z, double* __restrict__ y) {
    constexpr int C = 64 * UNR;
    constexpr int LPR = 64 / R;
    __shared__ double lds[NW][2][C];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* pt = lds[wv][0];
    double* pv = lds[wv][1];
    const int ntiles = (A.n + R - 1) / R;
    const int row_l = lane / LPR, sub = lane % LPR;
    for (int tile = blockIdx.x * NW + wv; tile < ntiles; tile += gridDim.x * NW) {
        const int r0 = tile * R;
        const int r = r0 + row_l;
        int b = 0, e = 0;
        if (r < A.n) { b = A.rowptr[r]; e = A.rowptr[r + 1]; }
        const int q0 = __builtin_amdgcn_readfirstlane(b);
        const int rl = min(R, A.n - r0) - 1;
        const int q1 = __builtin_amdgcn_readlane(e, rl * LPR);
        double s0 = 0.0, s1 = 0.0;
        double vv[UNR]; int cc[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = q0 + u * 64 + lane;
            const bool ok = idx < q1;
            vv[u] = ok ? A.val[idx] : 0.0; cc[u] = ok ? A.col[idx] : 0;
        }
        for (int base = q0; base < q1; base += C) {
            Z2 zz[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) zz[u] = Z[cc[u]];
            double vn[UNR]; int cn[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + C + u * 64 + lane;
                const bool ok = idx < q1;
                vn[u] = ok ? A.val[idx] : 0.0; cn[u] = ok ? A.col[idx] : 0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) { pt[u * 64 + lane] = vv[u] * zz[u].t; pv[u * 64 + lane] = vv[u] * zz[u].v; }
            wave_lds_sync();
            const int lo = max(b, base) - base, hi = min(e, base + C) - base;
            for (int i = lo + sub; i < hi; i += LPR) { s0 += pt[i]; s1 += pv[i]; }
            wave_lds_sync();
#pragma unroll
            for (int u = 0; u < UNR; ++u) { vv[u] = vn[u]; cc[u] = cn[u]; }
        }
        if (LPR > 1) { s0 = group_sum<LPR>(s0); s1 = group_sum<LPR>(s1); }
        if (r < A.n && sub == 0) { y[2 * r] = s0; y[2 * r + 1] = s1; }
    }
}

template <class F>
double time_us(F&& launch, int reps, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return 1e3 * ms / reps;
}

struct Mat { int n; long nnz; int *rp, *col; double* val; std::vector<int> hrp, hcol; };

Mat make(int n, double mean_deg, double hub_frac, double hub_share, unsigned seed) {
    // symmetric pattern: chain + random edges; hub_frac of the nodes receive hub_share of the random edge endpoints
    std::mt19937_64 rng(seed);
    const long E = (long)(mean_deg * n / 2);
    std::vector<std::vector<int>> adj((size_t)n);
    for (int i = 0; i + 1 < n; ++i) { adj[i].push_back(i + 1); adj[i + 1].push_back(i); }
    const int nh = std::max(1, (int)(hub_frac * n));
    std::vector<int> hubs(nh);
    for (int i = 0; i < nh; ++i) hubs[i] = (int)(rng() % n);
    for (long k = 0; k < E; ++k) {
        int a = (hub_share > 0 && (rng() % 1000) < hub_share * 1000) ? hubs[rng() % nh] : (int)(rng() % n);
        int b = (int)(rng() % n);
        if (a == b) continue;
        adj[a].push_back(b); adj[b].push_back(a);
    }
    Mat M; M.n = n; M.hrp.assign(n + 1, 0);
    for (int r = 0; r < n; ++r) {
        auto& v = adj[r]; v.push_back(r);
        std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        for (int c : v) M.hcol.push_back(c);
        M.hrp[r + 1] = (int)M.hcol.size();
    }
    M.nnz = (long)M.hcol.size();
    std::vector<double> hv((size_t)M.nnz, 1.0);
    for (long i = 0; i < M.nnz; ++i) hv[i] = 0.5 + (double)(rng() % 1000) / 1000.0;
    CK(hipMalloc(&M.rp, (n + 1) * 4)); CK(hipMalloc(&M.col, M.nnz * 4)); CK(hipMalloc(&M.val, M.nnz * 8));
    CK(hipMemcpy(M.rp, M.hrp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(M.col, M.hcol.data(), M.nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(M.val, hv.data(), M.nnz * 8, hipMemcpyHostToDevice));
    return M;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Cfg { int n; double deg, hf, hs; const char* nm; };
    const Cfg cfgs[] = {{100000, 6, 0, 0, "c4 it0-like (mean 7)"}, {100000, 6, 0.03, 0.5, "c4 it1-like (mean 7, hubs)"},
                        {100000, 26, 0, 0, "c4 mid (mean 27)"}, {100000, 26, 0.05, 0.3, "c4 mid hubs"}, {100000, 40, 0, 0, "c4 late (mean 41)"},
                        {10000, 12, 0, 0, "c2 it0-like (mean 13)"}, {10000, 12, 0.06, 0.4, "c2 it1-like hubs"}, {10000, 90, 0.06, 0.3, "c2 late (mean 90, hubs)"}};
    for (const Cfg& c : cfgs) {
        Mat M = make(c.n, c.deg, c.hf, c.hs, 7);
        const int n = c.n;
        int maxlen = 0; for (int r = 0; r < n; ++r) maxlen = std::max(maxlen, M.hrp[r + 1] - M.hrp[r]);
        Z2* Z; double *y, *y2;
        CK(hipMalloc(&Z, (size_t)n * 16)); CK(hipMalloc(&y, (size_t)n * 16)); CK(hipMalloc(&y2, (size_t)n * 16));
        std::vector<Z2> hz((size_t)n); for (int i = 0; i < n; ++i) { hz[i].t = 1.0 + i % 7; hz[i].v = 0.25 * (i % 5); }
        CK(hipMemcpy(Z, hz.data(), (size_t)n * 16, hipMemcpyHostToDevice));
        CsrView A{n, M.rp, M.col, M.val};
        printf("== %s: n=%d nnz=%ld mean %.1f maxlen %d\n", c.nm, n, M.nnz, (double)M.nnz / n, maxlen);
        auto report = [&](const char* nm, double us) { printf("   %-44s %7.2f us   %.3f ns/nnz\n", nm, us, 1e3 * us / M.nnz); fflush(stdout); };
        // reference result
        k_rowgroup<16, 2><<<256, 1024, 0, s>>>(A, Z, y); CK(hipStreamSynchronize(s));
        std::vector<double> hy((size_t)2 * n), hy2((size_t)2 * n);
        CK(hipMemcpy(hy.data(), y, (size_t)n * 16, hipMemcpyDeviceToHost));
        auto check = [&](const char* nm) {
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(hy2.data(), y2, (size_t)n * 16, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < hy.size(); ++i) md = std::max(md, std::fabs(hy[i] - hy2[i]) / (1.0 + std::fabs(hy[i])));
            if (md > 1e-12) printf("   !! %s mismatch %.3e\n", nm, md);
        };
#define RG(G, U, BLK, GRID) report("rowgroup G=" #G " unr=" #U " blk=" #BLK " grid=" #GRID, time_us([&] { k_rowgroup<G, U><<<GRID, BLK, 0, s>>>(A, Z, y2); }, 100, s)); check("rg");
        RG(4, 1, 1024, 256) RG(4, 2, 1024, 256) RG(8, 2, 1024, 256) RG(16, 2, 1024, 256) RG(8, 2, 1024, 512) RG(16, 2, 1024, 512) RG(8, 4, 512, 1024) RG(4, 2, 256, 2048)
#define WT(K, R, U, NW, GRID) report(#K " R=" #R " unr=" #U " waves/wg=" #NW " grid=" #GRID, time_us([&] { K<R, U, NW><<<GRID, NW * 64, 0, s>>>(A, Z, y2); }, 100, s)); check(#K);
        WT(k_wavetile, 64, 2, 8, 256) WT(k_wavetile, 64, 4, 8, 256) WT(k_wavetile, 32, 2, 8, 256) WT(k_wavetile, 32, 4, 8, 256) WT(k_wavetile, 16, 2, 8, 256) WT(k_wavetile, 16, 2, 16, 256)
        WT(k_wavetile, 32, 2, 16, 256) WT(k_wavetile, 32, 2, 4, 512) WT(k_wavetile, 16, 2, 4, 1024) WT(k_wavetile, 8, 2, 8, 256) WT(k_wavetile, 8, 1, 16, 256)
        WT(k_wavetile_pf, 64, 2, 8, 256) WT(k_wavetile_pf, 32, 2, 8, 256) WT(k_wavetile_pf, 32, 4, 8, 256) WT(k_wavetile_pf, 16, 2, 16, 256) WT(k_wavetile_pf, 32, 2, 16, 256) WT(k_wavetile_pf, 16, 2, 8, 512)
        WT(k_wavetile_pf, 8, 2, 16, 256) WT(k_wavetile_pf, 8, 1, 16, 256)
        CK(hipFree(Z)); CK(hipFree(y)); CK(hipFree(y2)); CK(hipFree(M.rp)); CK(hipFree(M.col)); CK(hipFree(M.val));
    }
    return 0;
}
