// mac_amd/csrc/woodbury.h -- exact preconditioner for "odometry chain + a few hundred closures".
//
// The tridiagonal preconditioner of precond.h leaves the loop closures to the iteration: 1 400-2 300
// LOBPCG iterations per cold solve on ais2klinik, and no convergence at all on chains whose weak
// links are bridged by closures.  With s active closures  L + sigma I = T + U D U^T  where
// T = (chain Laplacian) + sigma I is tridiagonal SPD, U = [e_i - e_j] (n x s), D = diag(x_k w_k), so
//     (L + sigma I)^-1 r = y - Z C^-1 U^T y,   y = T^-1 r,  Z = T^-1 U,  C = D^-1 + U^T Z   (s x s, SPD)
// (Sherman-Morrison-Woodbury).  Per solve: s tridiagonal solves in ONE launch (one workgroup each,
// the kernel of precond.h with a two-entry right-hand side), C assembled by gathers and inverted by a
// hand-written blocked Gauss-Jordan elimination on the matrix cores (k_gj_step below, v_mfma_f64_16x16x4_f64;
// rounds 1-3 called rocSOLVER's dpotrf + dpotri here: 0.3 ms at s = 160, 1.2 ms at s = 400, 5 ms at s = 1 600),
// then every application costs the tridiagonal solve + an s x s GEMV + an n x s GEMV.  LOBPCG with this M^-1
// converges in 5-18 iterations where the tridiagonal one needed hundreds to thousands (NumPy
// prototype; the same counts on the device).
#pragma once
#include "precond.h"

namespace machip {

constexpr int kWbMaxS = 3072;     // closures beyond this: the dense inverse (1.7 ms at 2 137, 4.5 ms at 3 072, 9.2 ms at 4 096) costs more than it saves

struct WbView {
    int s;                 // active closures (upper off-band entries of L)
    size_t cap;            // column stride of Zt = c * 1024
    const int *ui, *uj;    // closure endpoints
    const double* uc;      // closure weights x_k w_k (> 0)
    double* Zt;            // T^-1 U, chunk-transposed columns
    int ld;                // leading dimension of Cm: s rounded up to whole 64 x 64 tiles (identity beyond s)
    double* Cm;            // capacitance matrix / its inverse, ld x ld
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
    const long s = W.s, ld = W.ld;
    for (long idx = (long)blockIdx.x * kBlock + threadIdx.x; idx < ld * ld; idx += (long)gridDim.x * kBlock) {
        const int a = (int)(idx % ld), b = (int)(idx / ld);
        double v = a == b ? 1.0 : 0.0;                     // identity beyond s: whole tiles for k_gj_step, inverse unchanged
        if (a < s && b < s) {
            const double* z = W.Zt + (size_t)b * W.cap;
            v = z[tri_perm(W.ui[a], L.c, L.stride)] - z[tri_perm(W.uj[a], L.c, L.stride)];
            if (a == b) v += 1.0 / W.uc[a];
        }
        W.Cm[idx] = v;
    }
}

// ---- C^-1 by blocked Gauss-Jordan elimination on the matrix cores (round 4: replaces rocSOLVER's dpotrf + dpotri, the one
// vendor-library call the path had; nx:79-107 `_LUSolver` semantics: an exact solve per application) ----
// In-place Gauss-Jordan without pivoting (C is symmetric positive definite: every pivot block is a Schur complement, SPD
// itself), pivot blocks of 32: step k maps  A -> A'  with  P = A_KK^-1,
//     A'_ij = A_ij - A_iK P A_Kj (i, j not in K),   A'_iK = -A_iK P,   A'_Kj = P A_Kj,   A'_KK = P,
// and after all ld/32 steps A' = C^-1 -- the inverse comes out directly, so applying it stays a GEMV (two triangular solves
// per application would be a chain of s dependent steps).  One launch per step, ping-pong between two buffers (a tile reads
// the old pivot row / column blocks that other workgroups rewrite).  Workgroup = one 64 x 64 tile of A', 4 waves:
//   * wave 0 inverts the 32 x 32 pivot block (scalar Gauss-Jordan, the block in registers as 4 x 4 sub-blocks per lane, only
//     row / column p through LDS) while waves 1-3 stage A_iK / A_Kj; every workgroup does it redundantly -- a separate
//     launch would cost more than the few us it takes;
//   * -(A_iK P) by v_mfma_f64_16x16x4_f64 (rows inside K: P itself), then the tile as ONE uniform product
//     A' = base + (-A_iK P) * B  with  B = A_Kj (columns inside K: identity columns)  and  base = A_ij (0 on pivot rows /
//     columns): 32 MFMAs per wave; lane layout of the f64 MFMA: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
//     D[row = (l >> 4) + 4 reg][col = l & 15].
// 2 s^3 flops in all (six times a Cholesky factorisation, at matrix-core rate): 22 launches, ~0.15 ms at s = 645.
// bad <- 1 on a non-positive scalar pivot (C not positive definite numerically: the caller falls back to Lanczos).
constexpr int kGjB = 32;
constexpr int kGjT = 64;
typedef double gj_d4 __attribute__((ext_vector_type(4)));

// P = A_KK^-1 by ONE wave: scalar Gauss-Jordan, the 32 x 32 block in registers (lane (br, bc) keeps the 4 x 4 sub-block (4 br.., 4 bc..)).
// Per pivot p only row p and column p travel: their owners park them in LDS (rowp[32], colp[32]), every lane reads the 4 + 4 entries
// its sub-block needs (a wave's LDS traffic is ordered: no barrier), 16 multiply-adds per lane.  (First build: all 256 threads on an LDS
// copy, one barrier per pivot -- 0.53 us per pivot, 17 of the kernel's 28 us.)  Returns 1 on a non-positive pivot.
__device__ __forceinline__ int gj_invert_block(double (&m)[4][4], double* rowp, double* colp, int br, int bc) {
    int bad = 0;
    for (int pb = 0; pb < kGjB / 4; ++pb) {
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) {
            const int p = 4 * pb + pj;
            if (br == pb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) rowp[4 * bc + j] = m[pj][j];
            }
            if (bc == pb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) colp[4 * br + i] = m[i][pj];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double piv = rowp[p];
            double rp[4], cp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) rp[j] = rowp[4 * bc + j];
#pragma unroll
            for (int i = 0; i < 4; ++i) cp[i] = colp[4 * br + i];
            if (!(piv > 0.0) || !(piv < 1e300)) bad = 1;
            double ip = __builtin_amdgcn_rcp(piv);     // hardware reciprocal + two Newton steps (no IEEE division on the chain)
            ip = ip * __builtin_fma(-piv, ip, 2.0);
            ip = ip * __builtin_fma(-piv, ip, 2.0);
            // generic update for all 16 entries, then the pivot row / column / element by their owners (pj is a compile-time
            // index: static registers, lane predicates only -- the per-entry compare-and-select form was 40 % more instructions)
            double sc[4], nc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[j] = rp[j] * ip;
#pragma unroll
            for (int i = 0; i < 4; ++i) nc[i] = -cp[i] * ip;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = __builtin_fma(-cp[i], sc[j], m[i][j]);
            if (bc == pb) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i][pj] = nc[i];
            }
            if (br == pb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) m[pj][j] = sc[j];
                if (bc == pb) m[pj][pj] = ip;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();          // (everybody has read row / column p before the next pivot's owners overwrite them)
        }
    }
    return bad;
}

template <int VAR = 0>     // (VAR != 0: timing builds of tools/ubench_gj.hip -- 1: no pivot-block inversion, 2: nor the products)
__global__ __launch_bounds__(256) void k_gj_step(const double* __restrict__ src, double* __restrict__ dst, int ld, int kb, int* bad,
                                                 const double* __restrict__ pin = nullptr, double* __restrict__ pout = nullptr) {
    __shared__ double sP[2][kGjB][kGjB + 1];
    __shared__ double sA[kGjT][kGjB + 1];      // A_iK: rows of the tile x pivot columns
    __shared__ double sT[kGjT][kGjB + 1];      // -(A_iK P); rows inside the pivot block: P
    __shared__ double sB[kGjB][kGjT + 1];      // A_Kj: pivot rows x columns of the tile; columns inside the pivot block: identity
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Look-ahead (large matrices, solver.h): with `pin` the inverse of THIS step's pivot block comes from the previous launch -- the
    // workgroup whose tile holds the NEXT pivot block inverts it once its tile is updated and leaves it in `pout` --, so only one
    // workgroup per step runs the ~7 us scalar inversion instead of all of them in front of their products (1 156 workgroups at
    // s = 2 137: 40 -> ~20 us per step).  That workgroup is dispatched first: the tile grid is rotated so that (0, 0) is its tile.
    const int tiles = gridDim.x, tnext = pout ? min((kb + kGjB) / kGjT, tiles - 1) : 0;
    const int ty = (blockIdx.y + tnext) % tiles, tx = (blockIdx.x + tnext) % tiles;
    const int r0 = ty * kGjT, c0 = tx * kGjT;
    const int li = lane & 15, lk = lane >> 4;
    // wave 0 inverts the pivot block (below); waves 1-3 stage A_iK and A_Kj in LDS meanwhile
    const int br = lane >> 3, bc = lane & 7;         // wave 0: lane (br, bc) keeps the 4 x 4 sub-block (4 br.., 4 bc..) of A_KK in registers
    double m[4][4];
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = pin ? pin[(4 * br + i) * kGjB + 4 * bc + j] : src[(size_t)(kb + 4 * br + i) * ld + kb + 4 * bc + j];
    } else {
        // all 22 loads of a thread in flight at once (unconditional, clamped index; a loop with a run-time trip count is
        // compiled into one load -> wait -> LDS store round trip per element: 11 dependent cold misses, 7 of the first
        // build's 10 us of "loads")
        constexpr int NE = kGjT * kGjB, NQ = (NE + 191) / 192;
        double va[NQ], vb[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = min(tid - 64 + 192 * q, NE - 1);
            va[q] = src[(size_t)(r0 + e / kGjB) * ld + kb + e % kGjB];
            vb[q] = src[(size_t)(kb + e / kGjT) * ld + c0 + e % kGjT];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int e = tid - 64 + 192 * q;
            if (e < NE) {
                sA[e / kGjB][e % kGjB] = va[q];
                const int r = e / kGjT, gc = c0 + e % kGjT;
                const bool in_k = gc >= kb && gc < kb + kGjB;
                sB[r][e % kGjT] = in_k ? (gc - kb == r ? 1.0 : 0.0) : vb[q];
            }
        }
    }
    // the tile itself, already in the MFMA's result layout (zero on pivot rows / columns)
    const int wr = wv >> 1, wc = wv & 1;
    gj_d4 acc[2][2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = r0 + 32 * wr + 16 * bi + lk + 4 * q, col = c0 + 32 * wc + 16 * bj + li;
                const bool piv = (row >= kb && row < kb + kGjB) || (col >= kb && col < kb + kGjB);
                acc[bi][bj][q] = piv ? 0.0 : src[(size_t)row * ld + col];
            }
    // ---- P = A_KK^-1: scalar Gauss-Jordan by ONE wave, the block in registers (first build: all 256 threads on an LDS copy,
    // one barrier per pivot -- 0.53 us per pivot, 17 of the kernel's 28 us).  Per pivot p only row p and column p travel:
    // their owners park them in LDS, every lane reads the 4 + 4 entries its sub-block needs (a wave's LDS traffic is ordered:
    // no barrier), 16 multiply-adds per lane. ----
    int isbad = 0;
    if (wv == 0 && !VAR && !pin) isbad = gj_invert_block(m, &sP[1][0][0], &sP[1][0][0] + kGjB, br, bc);     // (scratch: sP[1] is otherwise unused)
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sP[0][4 * br + i][4 * bc + j] = m[i][j];
    }
    __syncthreads();
    if (isbad && tid == 0) *bad = 1;          // (every workgroup that inverted the block saw the same pivots)
    // ---- sT = -(A_iK P): wave w owns rows 16 w .. 16 w + 15, both 16-column halves ----
    if (VAR < 2) {
        gj_d4 t[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) { t[cb][0] = 0.0; t[cb][1] = 0.0; t[cb][2] = 0.0; t[cb][3] = 0.0; }
#pragma unroll
        for (int kk = 0; kk < kGjB / 4; ++kk) {
            const double a = sA[16 * wv + li][4 * kk + lk];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) t[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, sP[0][4 * kk + lk][16 * cb + li], t[cb], 0, 0, 0);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rl = 16 * wv + lk + 4 * q, cl = 16 * cb + li, row = r0 + rl;
                const bool in_k = row >= kb && row < kb + kGjB;
                sT[rl][cl] = in_k ? sP[0][row - kb][cl] : -t[cb][q];
            }
    }
    __syncthreads();
    // ---- the tile: base + sT * sB, a 32 x 32 quadrant per wave ----
#pragma unroll
    for (int kk = 0; kk < (VAR < 2 ? kGjB / 4 : 0); ++kk) {
        double a[2], b[2];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) a[bi] = sT[32 * wr + 16 * bi + li][4 * kk + lk];
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) b[bj] = sB[4 * kk + lk][32 * wc + 16 * bj + li];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj) acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], b[bj], acc[bi][bj], 0, 0, 0);
    }
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = r0 + 32 * wr + 16 * bi + lk + 4 * q, col = c0 + 32 * wc + 16 * bj + li;
                dst[(size_t)row * ld + col] = acc[bi][bj][q];
            }
    // ---- look-ahead: the next pivot block sits in this tile (quadrant (o, o)): its owner wave parks it in LDS, wave 0 inverts it ----
    const int kn = kb + kGjB;
    if (pout && kn < ld && r0 == (kn / kGjT) * kGjT && c0 == r0) {       // (uniform per workgroup)
        const int o = kn - r0;                       // 0 or 32
        double (*stage)[kGjB + 1] = sA;              // (A_iK is consumed: the products are done -- barrier below orders the reuse)
        __syncthreads();
        if (wr == o / 32 && wc == o / 32) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) stage[16 * bi + lk + 4 * q][16 * bj + li] = acc[bi][bj][q];
        }
        __syncthreads();
        if (wv == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = stage[4 * br + i][4 * bc + j];
            if (gj_invert_block(m, &sP[1][0][0], &sP[1][0][0] + kGjB, br, bc) && lane == 0) *bad = 1;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pout[(4 * br + i) * kGjB + 4 * bc + j] = m[i][j];
        }
    }
}

// ---- update + tridiagonal solve + g in ONE single-workgroup launch (round 4; n <= 16 384) ------------------------------------
// Between two products L w the preconditioned iteration runs k_lob_update (Rayleigh-Ritz on the 15 sums, x / p / Lx / Lp / r),
// k_tri_solve (y = T^-1 r, one workgroup) and k_wb_g (g = U^T y): three dependent launches of ~5-7 us each around ~1 us of work on
// a pose graph.  Here thread t of one 1 024-thread workgroup does all of it for the CMAX unknowns it owns in the solver's
// chunk layout: update (the residual stays in registers), the two sweeps of the LU solve with their scans, y into wT, then --
// behind a barrier: the same CU wrote it -- the s differences g.  Same arithmetic per unknown as the three kernels; the
// 1-norm of the residual is reduced by this one workgroup (partR slot 0, the other slots zeroed).
template <int CMAX>
__global__ __launch_bounds__(kTriThreads) void k_lob_fused(LobView L, WbView W, int jrel) {
    __shared__ double sc[8];
    __shared__ double sA[16], sB[16], sL[16];
    __shared__ double s_pa[CMAX * kTriThreads];
    const int t = threadIdx.x, n = L.n;
    const int itn = L.st->it0 + jrel + 1;          // index of the iterate this launch produces
    if (t >= 64 && t < 128) {                      // record of the iterate entering this launch (cf. k_lob_update)
        double a = 0.0;
        const double* pr = L.partR + (size_t)((itn - 1) & 1) * kMaxGrid;
        for (int i = t - 64; i < L.P_a; i += 64) a += pr[i];
        a = wave_total(a);
        if (t == 64) {
            L.hrec[4 * (size_t)(itn - 1)] = L.st->theta;
            L.hrec[4 * (size_t)(itn - 1) + 1] = a;
            L.hrec[4 * (size_t)(itn - 1) + 2] = lob_tag(L.st->epoch, itn - 1);
        }
    }
    if (t < 64) {
        double s[kLobNS];
#pragma unroll
        for (int q = 0; q < kLobNS; ++q) s[q] = 0.0;
        for (int base = 0; base < L.P_c; base += 128) {    // (two columns of partials in flight: 1 024 threads leave 128 VGPRs per lane)
            double v[kLobNS][2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < kLobNS; ++q) v[q][c] = L.part[(size_t)q * kMaxGrid + base + t + 64 * c];   // < kMaxGrid
#pragma unroll
            for (int q = 0; q < kLobNS; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) s[q] += (base + t + 64 * c < L.P_c) ? v[q][c] : 0.0;
        }
        wave_total_n<kLobNS>(s);
        if (t == 0) {
            const LobCoef co = lob_rayleigh_ritz(s, L.n, L.st->havep0 != 0 || jrel > 0);
            sc[0] = co.z0; sc[1] = co.z1; sc[2] = co.z2; sc[3] = co.theta; sc[4] = co.mx; sc[5] = co.mw; sc[6] = co.mp;
            sc[7] = co.bad ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    if (t == 0) {
        L.st->theta = sc[3];
        if (sc[7] != 0.0) L.st->bad = 1;
    }
    const double z0 = sc[0], z1 = sc[1], z2 = sc[2], th = sc[3], mx = sc[4], mw = sc[5], mp = sc[6];
    double rr[CMAX];
    double l1 = 0.0;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        const int e = t * CMAX + i, k = i * kTriThreads + t;
        rr[i] = 0.0;
        if (e < n) {
            const double w = L.wT[k] - mw;
            const double pn = z1 * w + z2 * (L.p[e] - mp);
            const double lpn = z1 * L.Lw[e] + z2 * L.Lp[e];
            const double xn = z0 * (L.x[e] - mx) + pn;
            const double lxn = z0 * L.Lx[e] + lpn;
            L.p[e] = pn; L.Lp[e] = lpn; L.x[e] = xn; L.Lx[e] = lxn;
            const double res = lxn - th * xn;
            rr[i] = res;
            L.rT[k] = res;
            l1 += fabs(res);
        }
    }
    {   // ||r||_1: wave totals, then 16 values
        l1 = wave_total(l1);
        if ((t & 63) == 0) sL[t >> 6] = l1;
        __syncthreads();
        if (t == 0) {
            double a = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) a += sL[q];
            L.partR[(size_t)(itn & 1) * kMaxGrid] = a;
        }
        if (t >= 1 && t < L.P_a) L.partR[(size_t)(itn & 1) * kMaxGrid + t] = 0.0;
    }
    // ---- y = T^-1 r (tri_solve_body with the right-hand side in registers) ----
    double y[CMAX];
    double run = 0.0, prod = 1.0;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        const int k = i * kTriThreads + t;
        const double l = L.tl[k];
        run = rr[i] - l * run;
        prod = -l * prod;
        y[i] = run; s_pa[k] = prod;
    }
    const double carry = affine_carry_in<false>(prod, run, sA, sB);
    double xr = 0.0, pb = 1.0;
#pragma unroll
    for (int i = CMAX - 1; i >= 0; --i) {
        const int k = i * kTriThreads + t;
        const double cu = L.tcu[k];
        xr = (y[i] + s_pa[k] * carry) * L.tdinv[k] - cu * xr;
        pb = -cu * pb;
        y[i] = xr; s_pa[k] = pb;
    }
    const double carry2 = affine_carry_in<true>(pb, xr, sA, sB);
#pragma unroll
    for (int i = 0; i < CMAX; ++i) L.wT[i * kTriThreads + t] = y[i] + s_pa[i * kTriThreads + t] * carry2;
    // ---- g = U^T y ----
    if (W.s > 0) {
        __threadfence_block();
        __syncthreads();
        for (int a = t; a < W.s; a += kTriThreads)
            W.g[a] = L.wT[tri_perm(W.ui[a], L.c, L.stride)] - L.wT[tri_perm(W.uj[a], L.c, L.stride)];
    }
}

// ---- application: w <- y - Z C^-1 U^T y with y = T^-1 r already in wT ----
__global__ __launch_bounds__(kBlock) void k_wb_g(LobView L, WbView W) {
    for (int a = blockIdx.x * kBlock + threadIdx.x; a < W.s; a += gridDim.x * kBlock)
        W.g[a] = L.wT[tri_perm(W.ui[a], L.c, L.stride)] - L.wT[tri_perm(W.uj[a], L.c, L.stride)];
}
__global__ __launch_bounds__(kBlock) void k_wb_h(WbView W) {      // h = Cinv g: one wave per entry, row a of Cinv (contiguous)
    const int lane = threadIdx.x & 63;
    const int s = W.s;
    for (int a = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); a < s; a += gridDim.x * (kBlock / 64)) {
        const double* col = W.Cm + (size_t)a * (size_t)W.ld;
        double acc = 0.0;
        for (int b = lane; b < s; b += 64) acc += col[b] * W.g[b];
        acc = wave_total(acc);
        if (lane == 0) W.h[a] = acc;
    }
}
// w <- y - Z h: the n x s product.  A workgroup owns 16 rows (a 128-byte segment of every column of Z), its 1 024 threads are
// 16 rows x 64 column groups: thread (r, c) walks the columns c, c + 64, ... (8 loads in flight), the 64 group sums of a row are
// added in group order through LDS -- deterministic.  cap / 16 workgroups (128 on intel: rounds 1-3 ran one thread per row, 8
// workgroups, 33 us per application at s = 645; the first round-4 form, 64-row tiles x 16 waves = 32 workgroups, 8.9 us).
constexpr int kWbwThreads = 1024;
constexpr int kWbwRows = 16;
__global__ __launch_bounds__(kWbwThreads) void k_wb_w(LobView L, WbView W) {
    __shared__ double sred[kWbwThreads / kWbwRows][kWbwRows + 1];
    const size_t cap = W.cap;
    const int s = W.s, r = threadIdx.x & (kWbwRows - 1), cg = threadIdx.x / kWbwRows;
    constexpr int NG = kWbwThreads / kWbwRows;      // 64 column groups
    for (size_t k0 = (size_t)blockIdx.x * kWbwRows; k0 < cap; k0 += (size_t)gridDim.x * kWbwRows) {
        const size_t k = k0 + r;             // (cap is a multiple of 1024)
        double acc = 0.0;
        int b = cg;
        for (; b + 7 * NG < s; b += 8 * NG) {
            double z[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) z[q] = W.Zt[k + (size_t)(b + q * NG) * cap];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_fma(z[q], W.h[b + q * NG], acc);
        }
        for (; b < s; b += NG) acc = __builtin_fma(W.Zt[k + (size_t)b * cap], W.h[b], acc);
        sred[cg][r] = acc;
        __syncthreads();
        if (cg == 0) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) t += sred[g][r];
            L.wT[k] -= t;
        }
        __syncthreads();
    }
}

}  // namespace machip
