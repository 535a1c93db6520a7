// tools/ubench_gj.hip -- round 4: the hand-written blocked Gauss-Jordan inverse (woodbury.h, k_gj_step, v_mfma_f64_16x16x4_f64)
// that replaced rocSOLVER's dpotrf + dpotri on the exact chain + closures preconditioner: time and accuracy per size.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_gj.hip -o tools/bin/ubench_gj
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../mac_amd/csrc/woodbury.h"
namespace machip { thread_local std::string g_err; }
using namespace machip;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int VAR>
static float time_var(double* d0, double* d1, int ld, int* bad, hipStream_t st) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st));
        double *src = d0, *dst = d1;
        for (int kb = 0; kb < ld; kb += kGjB) { k_gj_step<VAR><<<dim3(ld / kGjT, ld / kGjT), 256, 0, st>>>(src, dst, ld, kb, bad); std::swap(src, dst); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, ms);
    }
    return best;
}
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int s : {43, 157, 400, 645, 800, 1100, 1600, 2137, 2400, 4096}) {
        const int ld = (s + kGjT - 1) / kGjT * kGjT;
        std::mt19937_64 rng(s);
        std::uniform_real_distribution<double> U(-1.0, 1.0), E(-2.0, 6.0);
        // SPD, badly scaled like a capacitance matrix (D^-1 spans many decades) and NOT a symmetric pattern of values a
        // transposed tile would reproduce: C = G G^T / m + diag(10^e), padded with the identity
        const int m = 2 * s;
        std::vector<double> G((size_t)s * m), C((size_t)ld * ld, 0.0);
        for (auto& g : G) g = U(rng);
        for (int i = 0; i < s; ++i)
            for (int j = 0; j <= i; ++j) {
                double a = 0.0;
                for (int k = 0; k < m; ++k) a += G[(size_t)i * m + k] * G[(size_t)j * m + k];
                a /= m;
                if (i == j) a += std::pow(10.0, E(rng));
                C[(size_t)i * ld + j] = a; C[(size_t)j * ld + i] = a;
            }
        for (int i = s; i < ld; ++i) C[(size_t)i * ld + i] = 1.0;
        double *d0, *d1; int* bad;
        CK(hipMalloc(&d0, sizeof(double) * ld * ld)); CK(hipMalloc(&d1, sizeof(double) * ld * ld)); CK(hipMalloc(&bad, 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        double* res = nullptr;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemcpy(d0, C.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice));
            CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0, st));
            double *src = d0, *dst = d1;
            for (int kb = 0; kb < ld; kb += kGjB) { k_gj_step<0><<<dim3(ld / kGjT, ld / kGjT), 256, 0, st>>>(src, dst, ld, kb, bad); std::swap(src, dst); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) best = std::min(best, ms);
            res = src;
        }
        std::vector<double> Ci((size_t)ld * ld);
        int hbad = 0;
        CK(hipMemcpy(Ci.data(), res, sizeof(double) * ld * ld, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        // ||C Ci - I||_max on a sample of rows (all rows up to s = 800), asymmetry of the result
        double err = 0.0, asym = 0.0, big = 0.0;
        const int stride = s <= 800 ? 1 : s / 200;
        for (int i = 0; i < s; i += stride)
            for (int j = 0; j < s; ++j) {
                double a = 0.0;
                for (int k = 0; k < s; ++k) a += C[(size_t)i * ld + k] * Ci[(size_t)k * ld + j];
                err = std::max(err, std::fabs(a - (i == j ? 1.0 : 0.0)));
            }
        for (int i = 0; i < s; ++i) for (int j = 0; j < i; ++j) { asym = std::max(asym, std::fabs(Ci[(size_t)i * ld + j] - Ci[(size_t)j * ld + i])); big = std::max(big, std::fabs(Ci[(size_t)i * ld + j])); }
        printf("s=%5d (ld %5d, %3d launches)  inverse %8.3f ms  (%.1f GFLOP/s of 2 s^3)   max|C Cinv - I| = %.2e   asymmetry %.1e of %.1e   bad=%d\n",
               s, ld, ld / kGjB, best, 2.0 * s * (double)s * s / (best * 1e6), err, asym, big, hbad);
        {   // bit reproducibility of the inverse: 40 more runs against the first
            std::vector<double> again((size_t)ld * ld);
            int diff = 0;
            for (int rep = 0; rep < 40; ++rep) {
                CK(hipMemcpy(d0, C.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice));
                double *src = d0, *dst = d1;
                for (int kb = 0; kb < ld; kb += kGjB) { k_gj_step<0><<<dim3(ld / kGjT, ld / kGjT), 256, 0, st>>>(src, dst, ld, kb, bad); std::swap(src, dst); }
                CK(hipMemcpyAsync(again.data(), src, sizeof(double) * ld * ld, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                if (memcmp(again.data(), Ci.data(), sizeof(double) * ld * ld)) ++diff;
            }
            printf("          40 repeats: %d differ from the first run bit for bit\n", diff);
        }
        {   // look-ahead form (the workgroup holding the next pivot block inverts it for the next launch): same bits? time?
            double* piv; CK(hipMalloc(&piv, sizeof(double) * 2 * kGjB * kGjB));
            std::vector<double> again((size_t)ld * ld);
            float lbest = 1e9f; int diff = 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemcpy(d0, C.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice));
                CK(hipEventRecord(e0, st));
                double *src = d0, *dst = d1;
                for (int kb = 0, k = 0; kb < ld; kb += kGjB, ++k) {
                    k_gj_step<0><<<dim3(ld / kGjT, ld / kGjT), 256, 0, st>>>(src, dst, ld, kb, bad, k ? piv + (size_t)(k & 1) * kGjB * kGjB : nullptr, piv + (size_t)((k + 1) & 1) * kGjB * kGjB);
                    std::swap(src, dst);
                }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) lbest = std::min(lbest, ms);
                CK(hipMemcpy(again.data(), src, sizeof(double) * ld * ld, hipMemcpyDeviceToHost));
                if (memcmp(again.data(), Ci.data(), sizeof(double) * ld * ld)) ++diff;
            }
            CK(hipGetLastError());
            printf("          look-ahead pivot inversion: inverse %8.3f ms (%.2f us per launch), %d of 6 runs differ from the plain form bit for bit\n", lbest, 1e3 * lbest / (ld / kGjB), diff);
            CK(hipFree(piv));
        }
        printf("          per launch: full %.2f us | without the pivot-block inversion %.2f us | loads + stores only %.2f us\n", 1e3 * best / (ld / kGjB),
               1e3 * time_var<1>(d0, d1, ld, bad, st) / (ld / kGjB), 1e3 * time_var<2>(d0, d1, ld, bad, st) / (ld / kGjB));
        CK(hipFree(d0)); CK(hipFree(d1)); CK(hipFree(bad));
    }
    return 0;
}
