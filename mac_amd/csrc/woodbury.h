// mac_amd/csrc/woodbury.h -- exact preconditioner for "odometry chain + a few hundred closures".
//
// The tridiagonal preconditioner of precond.h leaves the loop closures to the iteration: 1 400-2 300
// LOBPCG iterations per cold solve on ais2klinik, and no convergence at all on chains whose weak
// links are bridged by closures.  With s active closures  L + sigma I = T + U D U^T  where
// T = (chain Laplacian) + sigma I is tridiagonal SPD, U = [e_i - e_j] (n x s), D = diag(x_k w_k), so
//     (L + sigma I)^-1 r = y - Z C^-1 U^T y,   y = T^-1 r,  Z = T^-1 U,  C = D^-1 + U^T Z   (s x s, SPD)
// (Sherman-Morrison-Woodbury).  Per solve: s tridiagonal solves in ONE launch (one workgroup each,
// the kernel of precond.h with a two-entry right-hand side), C assembled by gathers, factored and
// inverted by rocSOLVER (dpotrf + dpotri: 0.3 ms at s = 160, 1.2 ms at s = 400, 5 ms at s = 1 600 --
// the dense factorisation is a library call, everything else is hand-written), then every
// application costs the tridiagonal solve + an s x s GEMV + an n x s GEMV.  LOBPCG with this M^-1
// converges in 5-18 iterations where the tridiagonal one needed hundreds to thousands (NumPy
// prototype; the same counts on the device).
#pragma once
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "precond.h"

namespace machip {

constexpr int kWbMaxS = 2048;     // closures beyond this: the dense factorisation (7 ms at 2 048) costs more than it saves

struct WbView {
    int s;                 // active closures (upper off-band entries of L)
    size_t cap;            // column stride of Zt = c * 1024
    const int *ui, *uj;    // closure endpoints
    const double* uc;      // closure weights x_k w_k (> 0)
    double* Zt;            // T^-1 U, chunk-transposed columns
    double* Cm;            // capacitance matrix / its inverse, column-major s x s
    double *g, *h;         // s-vectors
};

// ---- closure list from the assembled CSR (one workgroup, rows in chunks like the tridiagonal solver) ----
// counts[0] <- s; entries in row order (deterministic).  bad <- 1 on a non-negative off-band entry.
__global__ __launch_bounds__(kTriThreads) void k_wb_extract(CsrView A, int c, int cap_s, int* ui, int* uj, double* uc,
                                                            int* counts, int* bad) {
    __shared__ int s_scan[kTriThreads];
    const int t = threadIdx.x, n = A.n;
    int cnt = 0;
    for (int i = 0; i < c; ++i) {
        const int e = t * c + i;
        if (e < n)
            for (int p = A.rowptr[e]; p < A.rowptr[e + 1]; ++p) cnt += A.col[p] > e + 1;
    }
    s_scan[t] = cnt;
    __syncthreads();
    for (int o = 1; o < kTriThreads; o <<= 1) {
        const int add = t >= o ? s_scan[t - o] : 0;
        __syncthreads();
        s_scan[t] += add;
        __syncthreads();
    }
    int q = s_scan[t] - cnt;
    if (t == kTriThreads - 1) counts[0] = s_scan[t];
    int isbad = 0;
    for (int i = 0; i < c; ++i) {
        const int e = t * c + i;
        if (e < n)
            for (int p = A.rowptr[e]; p < A.rowptr[e + 1]; ++p) {
                const int col = A.col[p];
                if (col > e + 1) {
                    const double v = A.val[p];
                    if (!(v < 0.0)) isbad = 1;
                    if (q < cap_s) { ui[q] = e; uj[q] = col; uc[q] = -v; }
                    ++q;
                }
            }
    }
    if (isbad) *bad = 1;
}

// ---- Z = T^-1 U: one workgroup per closure, right-hand side e_i - e_j ----
template <int CMAX>
__global__ __launch_bounds__(kTriThreads) void k_wb_cols(LobView L, WbView W) {
    __shared__ double sA[16], sB[16];
    __shared__ double s_pa[CMAX * kTriThreads];
    const int b = blockIdx.x;
    const int pi = tri_perm(W.ui[b], L.c, L.stride), pj = tri_perm(W.uj[b], L.c, L.stride);
    tri_solve_body<CMAX>(L, [pi, pj](int k) { return (k == pi ? 1.0 : 0.0) - (k == pj ? 1.0 : 0.0); },
                         W.Zt + (size_t)b * W.cap, sA, sB, s_pa);
}

// n > 16 384: the multi-workgroup solve of precond.h, one right-hand side per blockIdx.y.  Z's own column is the
// sweep scratch, `pas` a second n x s buffer, `maps` = 4 doubles per (column, workgroup).
struct WbBig { double* pas; double* maps; };
__device__ __forceinline__ TriBigBuf wb_big_buf(const LobView& L, const WbView& W, const WbBig& Bg) {
    const size_t col = (size_t)blockIdx.y * W.cap;
    double* m = Bg.maps + (size_t)blockIdx.y * 4 * gridDim.x;
    return TriBigBuf{W.Zt + col, Bg.pas + col, m, m + gridDim.x, m + 2 * gridDim.x, m + 3 * gridDim.x, W.Zt + col};
}
__global__ __launch_bounds__(kTriThreads) void k_wb_big_fwd(LobView L, WbView W, WbBig Bg) {
    __shared__ double sA[16], sB[16];
    const int b = blockIdx.y;
    const int pi = tri_perm(W.ui[b], L.c, L.stride), pj = tri_perm(W.uj[b], L.c, L.stride);
    tri_big_fwd(L, wb_big_buf(L, W, Bg), [pi, pj](int k) { return (k == pi ? 1.0 : 0.0) - (k == pj ? 1.0 : 0.0); }, sA, sB);
}
__global__ __launch_bounds__(kTriThreads) void k_wb_big_mid(LobView L, WbView W, WbBig Bg) {
    __shared__ double sA[16], sB[16];
    tri_big_mid(L, wb_big_buf(L, W, Bg), sA, sB);
}
__global__ __launch_bounds__(kTriThreads) void k_wb_big_fin(LobView L, WbView W, WbBig Bg) {
    __shared__ double sA[16], sB[16];
    tri_big_fin(L, wb_big_buf(L, W, Bg), sA, sB);
}

// ---- C = D^-1 + U^T Z ----
__global__ __launch_bounds__(kBlock) void k_wb_cap(LobView L, WbView W) {
    const long s = W.s;
    for (long idx = (long)blockIdx.x * kBlock + threadIdx.x; idx < s * s; idx += (long)gridDim.x * kBlock) {
        const int a = (int)(idx % s), b = (int)(idx / s);
        const double* z = W.Zt + (size_t)b * W.cap;
        double v = z[tri_perm(W.ui[a], L.c, L.stride)] - z[tri_perm(W.uj[a], L.c, L.stride)];
        if (a == b) v += 1.0 / W.uc[a];
        W.Cm[idx] = v;
    }
}
// dpotri leaves the inverse in the lower triangle: mirror it so the GEMV can read contiguous columns
__global__ __launch_bounds__(kBlock) void k_wb_sym(WbView W) {
    const long s = W.s;
    for (long idx = (long)blockIdx.x * kBlock + threadIdx.x; idx < s * s; idx += (long)gridDim.x * kBlock) {
        const int a = (int)(idx % s), b = (int)(idx / s);      // element (row a, column b)
        if (a < b) W.Cm[idx] = W.Cm[(size_t)b + (size_t)a * s];
    }
}

// ---- application: w <- y - Z C^-1 U^T y with y = T^-1 r already in wT ----
__global__ __launch_bounds__(kBlock) void k_wb_g(LobView L, WbView W) {
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < W.s; a += gridDim.x * kBlock)
        W.g[a] = L.wT[tri_perm(W.ui[a], L.c, L.stride)] - L.wT[tri_perm(W.uj[a], L.c, L.stride)];
}
__global__ __launch_bounds__(kBlock) void k_wb_h(WbView W) {      // h = Cinv g: one wave per entry, column a of the symmetric Cinv
    const int lane = threadIdx.x & 63;
    const int s = W.s;
    for (int a = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); a < s; a += gridDim.x * (kBlock / 64)) {
        const double* col = W.Cm + (size_t)a * s;
        double acc = 0.0;
        for (int b = lane; b < s; b += 64) acc += col[b] * W.g[b];
        acc = wave_total(acc);
        if (lane == 0) W.h[a] = acc;
    }
}
__global__ __launch_bounds__(kBlock) void k_wb_w(LobView L, WbView W) {
    const size_t cap = W.cap;
    const int s = W.s;
    for (size_t k = (size_t)blockIdx.x * kBlock + threadIdx.x; k < cap; k += (size_t)gridDim.x * kBlock) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = 0;
        for (; b + 3 < s; b += 4) {
            a0 += W.Zt[k + (size_t)b * cap] * W.h[b];
            a1 += W.Zt[k + (size_t)(b + 1) * cap] * W.h[b + 1];
            a2 += W.Zt[k + (size_t)(b + 2) * cap] * W.h[b + 2];
            a3 += W.Zt[k + (size_t)(b + 3) * cap] * W.h[b + 3];
        }
        for (; b < s; ++b) a0 += W.Zt[k + (size_t)b * cap] * W.h[b];
        L.wT[k] -= (a0 + a1) + (a2 + a3);
    }
}

}  // namespace machip
