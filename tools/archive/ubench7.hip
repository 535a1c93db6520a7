// tools/ubench7.hip -- round 4: phase clocks of the ONE-LAUNCH column-panel step k_pan_step (panel.h), compiled with
// PAN_CLOCKS (100 MHz wall-clock stamps per workgroup), run as REAL consecutive steps (k_pan_mul + k_pan_fin) on
// config-4-like matrices; the panel form is built by the library's own k_pan_count/scan/fill.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench6.hip -o /tmp/ubench6 && /tmp/ubench6 [NP NB]
#define PIPE_CLOCKS 1
#define PAN_CLOCKS 1
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../mac_amd/csrc/panel.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static int FUSED = 1, BAND = 0;
template <int RPT>
void step(PanView& P, PipeView& L, hipStream_t s, int j) {
    if (FUSED == 2) k_pan_step<RPT, true><<<8 * ((P.NB + 7) / 8) * P.NP, kPanThreads, 0, s>>>(PAN_STEP_ARGS(P, L, j));     // a row block's workgroups on one XCD
    else if (FUSED) k_pan_step<RPT><<<P.NB * P.NP, kPanThreads, 0, s>>>(PAN_STEP_ARGS(P, L, j));
    else { k_pan_mul<RPT><<<P.NB * P.NP, kPanThreads, 0, s>>>(PAN_MUL_ARGS(P, L, j)); k_pan_fin<512><<<196, 512, 0, s>>>(PAN_FIN_ARGS(P, L, j)); }
}
template <int RPT>
void run(PanView P, PipeView L, hipStream_t s, long nnz, std::vector<double>& u0h) {
    const int g1 = P.NB * P.NP, g2 = 256;
    L.P = FUSED ? g1 : g2;
    double* u0; CK(hipMalloc(&u0, P.n * 8)); CK(hipMemcpy(u0, u0h.data(), P.n * 8, hipMemcpyHostToDevice));
    k_pipe_init<<<g2, kBlock, 0, s>>>(L, u0, 1);
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int steps = 40;
    CK(hipMemsetAsync(P.tick, 0, 4 * 256, s)); CK(hipMemsetAsync(P.claim, 0, 4 * 4096, s));
    for (int j = 0; j < 8; ++j) step<RPT>(P, L, s, j);
    CK(hipEventRecord(e0, s));
    for (int j = 8; j < 8 + steps; ++j) step<RPT>(P, L, s, j);
    CK(hipEventRecord(e1, s));
    CK(hipEventRecord(e2, s)); CK(hipEventSynchronize(e2));
    float ms, ms2; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&ms2, e1, e2));
    CK(hipGetLastError());
    const int gl = FUSED == 2 ? 8 * ((P.NB + 7) / 8) * P.NP : g1;
    {   // cross-variant check: the records after the last step, bit for bit
        std::vector<Z2> z((size_t)P.n);
        CK(hipMemcpy(z.data(), ((8 + steps) & 1) ? L.Z1 : L.Z0, (size_t)P.n * sizeof(Z2), hipMemcpyDeviceToHost));
        unsigned long long h = 1469598103934665603ull;
        for (int i = 0; i < P.n; ++i) { unsigned long long a, b2; memcpy(&a, &z[i].t, 8); memcpy(&b2, &z[i].v, 8); h = (h ^ a) * 1099511628211ull; h = (h ^ b2) * 1099511628211ull; }
        printf("   records after %d steps: hash %016llx  z[12345] = {%.17g, %.17g}\n", 8 + steps, h, z[12345].t, z[12345].v);
    }
    std::vector<long long> c((size_t)kMaxGrid * 16);
    CK(hipMemcpy(c.data(), P.clk, (size_t)gl * 16 * 8, hipMemcpyDeviceToHost));
    long long t0 = c[0];
    for (int b = 0; b < gl; ++b) if (c[(size_t)b * 16]) t0 = std::min(t0, c[(size_t)b * 16]);
    auto stat = [&](int i, const char* what) {
        double mn = 1e30, mx = 0, av = 0;
        for (int b = 0; b < gl; ++b) { if (!c[(size_t)b * 16]) continue; const double v = (c[(size_t)b * 16 + i] - t0) * 0.01; mn = std::min(mn, v); mx = std::max(mx, v); av += v; }
        printf("      %-52s min %6.2f  mean %6.2f  max %6.2f us\n", what, mn, av / g1, mx);
    };
    (void)ms2;
    printf("   NP=%d C=%d NB=%d NTB=%d RPT=%d grid=%d fused=%d spin=%d: %.2f us per step (nnz %ld)\n", P.NP, P.C, P.NB, P.NTB, RPT, g1, FUSED, P.spin_ticks, 1e3 * ms / steps, nnz);
    stat(0, "workgroup entry (wave 0)"); stat(1, "wave 1 entry"); stat(2, "records + tile heads arrived (wave 1)"); stat(3, "prologue done (wave 0)");
    stat(4, "barrier 1 passed"); stat(5, "panel in LDS, barrier 2 passed"); stat(6, "first round multiplied (wave 1)");
    stat(7, "tiles accumulated (wave 1)");
    if (FUSED) { stat(8, "partials stored and drained (wave 1)"); stat(9, "ticket / wait over"); stat(10, "slices finished (wave 1)"); stat(11, "last wave done"); }
    else { stat(8, "wave 1 done"); stat(9, "wave 15 done"); }
    if (FUSED != 2) {   // who is late?  completion (slot 9) by XCD (blockIdx mod 8), by panel and by row block
        double bx[8] = {0}, bp[64] = {0}, bb[256] = {0}; int nx[8] = {0}, np_[64] = {0}, nb_[256] = {0};
        for (int b = 0; b < g1; ++b) { const double v = (c[(size_t)b * 16 + (FUSED ? 11 : 9)] - t0) * 0.01; bx[b % 8] += v; nx[b % 8]++; bp[b % P.NP] += v; np_[b % P.NP]++; bb[b / P.NP] += v; nb_[b / P.NP]++; }
        printf("      done by XCD:  "); for (int i = 0; i < 8; ++i) printf(" %5.2f", bx[i] / std::max(1, nx[i])); printf("\n");
        printf("      done by panel:"); for (int i = 0; i < P.NP; ++i) printf(" %5.2f", bp[i] / std::max(1, np_[i])); printf("\n");
        printf("      done by block:"); for (int i = 0; i < P.NB; ++i) printf(" %5.2f", bb[i] / std::max(1, nb_[i])); printf("\n");
    }
    {   // are the SAME workgroups late every time?  the 16 slowest (block index: time) of this run's last step
        std::vector<std::pair<double, int>> v;
        for (int b = 0; b < gl; ++b) if (c[(size_t)b * 16]) v.push_back({(c[(size_t)b * 16 + (FUSED ? 11 : 9)] - t0) * 0.01, b});
        std::sort(v.begin(), v.end()); std::reverse(v.begin(), v.end());
        printf("      slowest:"); for (int i = 0; i < 16 && i < (int)v.size(); ++i) printf(" %d:%.1f", v[i].second, v[i].first); printf("\n");
        printf("      fastest:"); for (int i = 0; i < 8 && i < (int)v.size(); ++i) printf(" %d:%.1f", v[v.size() - 1 - i].second, v[v.size() - 1 - i].first); printf("\n");
    }
    if (FUSED) {   // hardware XCC id of every workgroup: do the workgroups w = x (mod 8) share one?
        int bad = 0; long long first[8];
        for (int w = 0; w < gl; ++w) { const long long x = c[(size_t)w * 16 + 12]; if (w < 8) first[w] = x; else if (x != first[w & 7]) ++bad; }
        printf("      XCC id of workgroups 0..7:"); for (int w = 0; w < 8; ++w) printf(" %lld", first[w]); printf("   workgroups off their residue class: %d of %d\n", bad, gl);
    }
    CK(hipFree(u0));
}

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int NPa = argc > 1 ? atoi(argv[1]) : 12, NBa = argc > 2 ? atoi(argv[2]) : 21;
    FUSED = argc > 3 ? atoi(argv[3]) : 1;
    const int spin_us = argc > 4 ? atoi(argv[4]) : 20;
    BAND = argc > 5 ? atoi(argv[5]) : 0;      // two-launch form only: band kept out of the tiles (k_pan_fin adds it)
    for (double deg : {26.0, 40.0}) {
        const int n = 100000;
        std::mt19937_64 rng(7);
        std::vector<std::vector<int>> adj((size_t)n);
        for (int i = 0; i + 1 < n; ++i) { adj[i].push_back(i + 1); adj[i + 1].push_back(i); }
        for (long k = 0; k < (long)(deg * n / 2); ++k) { int a = (int)(rng() % n), b = (int)(rng() % n); if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } }
        std::vector<int> rp(n + 1, 0), col; std::vector<double> val;
        for (int r = 0; r < n; ++r) {
            auto& v = adj[r]; std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
            col.push_back(r); val.push_back((double)v.size());
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
        PanView P; P.n = n; P.NP = NPa; P.C = (n + P.NP - 1) / P.NP; P.NP = (n + P.C - 1) / P.C;
        const int groups = (n + 63) / 64;
        P.NTB = std::min((groups + NBa - 1) / NBa, kPanWork * kPanTW); P.NB = (groups + P.NTB - 1) / P.NTB; P.TWW = (P.NTB + kPanWork - 1) / kPanWork;
        const size_t NT = (size_t)P.NB * P.NP * kPanWork * P.TWW;
        const size_t ecap = (size_t)nnz + (size_t)P.NB * P.NP * 64 * 128 + kPanSlack;
        CK(hipMalloc(&P.tptr, (NT + 1) * 4)); CK(hipMalloc(&P.tcount, NT * 4)); CK(hipMalloc(&P.thead, NT * 64 * 2));
        CK(hipMalloc(&P.bval, ecap * 8)); CK(hipMalloc(&P.bcol, ecap * 2)); CK(hipMalloc(&P.ypart, (size_t)P.NP * (n + 2) * 8)); CK(hipMalloc(&P.ps, (size_t)(P.NP + 1) * n * 4));
        CK(hipMalloc(&P.ovf, 4)); CK(hipMemset(P.ovf, 0, 4)); P.CELLS = 1; P.band = BAND; CK(hipMalloc(&P.bd, 3 * (size_t)n * 8)); CK(hipMalloc(&P.bpk, (size_t)n * 4)); CK(hipMalloc(&P.tick, 4 * 256)); CK(hipMalloc(&P.claim, 4 * 4096)); P.spin_ticks = spin_us * 100;
        CK(hipMalloc(&P.coef, 64)); CK(hipMalloc(&P.clk, 16 * 8 * kMaxGrid)); CK(hipMemset(P.clk, 0, 16 * 8 * kMaxGrid));
        hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
        CK(hipEventRecord(a0, s));
        k_pan_rows<<<(n + kBlock - 1) / kBlock, kBlock, 0, s>>>(A, P); k_pan_count<<<pan_build_grid(P.NB, P.NP), kPanThreads, 0, s>>>(A, P); k_pan_scan<<<1, 1024, 0, s>>>(P); k_pan_fill<<<pan_build_grid(P.NB, P.NP), kPanThreads, 0, s>>>(A, P);
        CK(hipEventRecord(a1, s)); CK(hipEventSynchronize(a1));
        float bms; CK(hipEventElapsedTime(&bms, a0, a1));
        std::vector<double> u0((size_t)n); for (int i = 0; i < n; ++i) u0[i] = (double)((i * 2654435761u) % 1000) / 500.0 - 1.0;
        int tot = 0; CK(hipMemcpy(&tot, P.tptr + NT, 4, hipMemcpyDeviceToHost));
        printf("== n=%d mean row %.1f nnz=%ld   panel form built in %.1f us, %d entries with padding (+%.1f %%)\n", n, (double)nnz / n, nnz, 1e3 * bms, tot, 100.0 * (tot - nnz) / nnz);
        const int RPT = (P.C + kPanWorkThreads - 1) / kPanWorkThreads;
        switch (RPT) {
            case 5: run<5>(P, L, s, nnz, u0); break; case 6: run<6>(P, L, s, nnz, u0); break; case 7: run<7>(P, L, s, nnz, u0); break;
            case 8: run<8>(P, L, s, nnz, u0); break; case 9: run<9>(P, L, s, nnz, u0); break; case 10: run<10>(P, L, s, nnz, u0); break;
            case 11: run<11>(P, L, s, nnz, u0); break; case 12: run<12>(P, L, s, nnz, u0); break; case 13: run<13>(P, L, s, nnz, u0); break;
            default: printf("RPT %d not instantiated\n", RPT); break;
        }
        CK(hipFree(drp)); CK(hipFree(dcol)); CK(hipFree(dval));
    }
    return 0;
}
