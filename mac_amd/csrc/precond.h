// mac_amd/csrc/precond.h -- preconditioned eigen-solver mode for chain-dominated graphs.
//
// The reference offers three flavours of networkx's TraceMIN (mac/utils/fiedler.py:38-42); its
// 'tracemin_pcg' one solves the inner systems with a preconditioned CG (nx:22-76).  Pose graphs
// (a long odometry chain plus few loop closures) are stiff -- lambda_2/lambda_max down to 1e-8 --
// and an un-preconditioned Krylov method pays ~sqrt(lambda_max/gap) dependent SpMVs there (10^4
// launches on ais2klinik).  This mode is the preconditioned alternative: LOBPCG with block size 1
// (three-term locally optimal recurrence on span{x, w, p}) where w = T^-1 r and T is the
// tridiagonal part of L(x) (the odometry chain with the full diagonal) plus a tiny shift.
//
// Per iteration three kernels:
//   k_tri_solve   w = T^-1 r           one workgroup, 1024 threads: every thread owns a chunk of
//                                      <= 16 consecutive unknowns in registers, the two first-order
//                                      recurrences of the LU solve are cut at the chunk borders and
//                                      stitched by a scan of affine maps (2 barriers each);
//                                      r / w / factors live chunk-transposed so all loads coalesce
//   k_spmv_vec<OpLob>  Lw = L w        + the 15 inner products of the Rayleigh-Ritz step
//   k_lob_update  3x3 Rayleigh-Ritz (every workgroup, identical arithmetic), x, p, Lx, Lp, r
// The host only watches (theta, ||r||_1) records in pinned memory, as in the Lanczos path.
#pragma once
#include "kernels.h"
#include "panel.h"

namespace machip {

constexpr int kTriThreads = 1024;
constexpr int kTriCMax = 16;                       // unknowns per thread held in registers
constexpr int kTriMaxN = kTriThreads * kTriCMax;   // 16384: above it the chunk lives in global scratch (k_tri_*_big)
constexpr int kTriBigMaxN = 1 << 21;
constexpr int kTriBigC = 4;                        // unknowns per thread of the multi-workgroup solver (n > 16 384)
constexpr int kLobNS = 15;                         // sums per Rayleigh-Ritz step
constexpr int kLobMaxChunk = 32;

struct LobState {
    double theta;      // Ritz value of the current x
    int it0;           // iterations completed before the running chunk
    int havep0;        // a search direction p exists at the start of the running chunk
    int bad;           // Rayleigh-Ritz breakdown seen
    unsigned int epoch;
};

struct LobView {
    int n, c, stride;                  // c = unknowns per thread of the tridiagonal solver, stride = threads (padded)
    double *x, *Lx, *p, *Lp, *Lw;      // natural order
    double *rT, *wT;                   // chunk-transposed: element e = t*c + i sits at i*stride + t
    double *tl, *tdinv, *tcu;          // LU of T, chunk-transposed (zero padded)
    double *ys, *pas;                  // chunk scratch of the big-n solver (n > 16 384), same layout
    double *mapA, *mapB, *mapA2, *mapB2;   // its per-WORKGROUP affine maps (forward / backward sweep)
    double* part;                      // [kLobNS][kMaxGrid] partial sums of the SpMV kernel
    double* partR;                     // [kMaxGrid] ||r||_1 partials of the update kernel
    int P_c, P_a;
    LobState* st;
    double* hrec;                      // pinned, device-mapped: (theta, ||r||_1, tag, -) per iteration; tag =
                                       // lob_tag(epoch, it) lets the host tell a landed record from stale memory
    unsigned long long* hflag;
};

__device__ __forceinline__ int tri_perm(int e, int c, int stride) { return (e % c) * stride + e / c; }
__host__ __device__ inline double lob_tag(unsigned int epoch, int it) { return (double)(epoch & 0xfffffu) * 16777216.0 + (double)it; }

// ---- scans over the 1024 threads of the solver workgroup --------------------------------------
// Affine maps f(y) = A y + B, composed in thread order (REV: in reverse thread order).  Returns
// the value the composition of all earlier maps gives to 0, i.e. the carry entering this thread.
template <bool REV>
__device__ __forceinline__ double affine_carry_in(double A, double B, double* sA, double* sB, double c0 = 0.0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double A2 = REV ? __shfl_down(A, o, 64) : __shfl_up(A, o, 64);
        const double B2 = REV ? __shfl_down(B, o, 64) : __shfl_up(B, o, 64);
        const bool ok = REV ? (lane + o < 64) : (lane >= o);
        if (ok) { B = A * B2 + B; A = A * A2; }     // self is the later map
    }
    // inclusive result of the neighbour = exclusive result of this lane
    double pA = REV ? __shfl_down(A, 1, 64) : __shfl_up(A, 1, 64);
    double pB = REV ? __shfl_down(B, 1, 64) : __shfl_up(B, 1, 64);
    if (REV ? lane == 63 : lane == 0) { pA = 1.0; pB = 0.0; }
    if (REV ? lane == 0 : lane == 63) { sA[w] = A; sB[w] = B; }
    __syncthreads();
    double carry = c0;                               // value entering this wave (c0 enters the first one)
    if (REV) { for (int k = 15; k > w; --k) carry = sA[k] * carry + sB[k]; }
    else     { for (int k = 0; k < w; ++k) carry = sA[k] * carry + sB[k]; }
    __syncthreads();
    return pA * carry + pB;
}

// Moebius maps u -> (m0 u + m1) / (m2 u + m3) in thread order; returns the homogeneous pair
// (p, q) the earlier maps give to (1, 0), i.e. u = infinity.  Entries are rescaled by powers of
// two so the products never overflow.
struct Mob { double m0, m1, m2, m3; };
__device__ __forceinline__ Mob mob_norm(Mob a) {
    const double m = fmax(fmax(fabs(a.m0), fabs(a.m1)), fmax(fabs(a.m2), fabs(a.m3)));
    if (m > 0.0) {
        const double s = ldexp(1.0, -ilogb(m));
        a.m0 *= s; a.m1 *= s; a.m2 *= s; a.m3 *= s;
    }
    return a;
}
__device__ __forceinline__ Mob mob_mul(const Mob& l, const Mob& e) {   // later o earlier
    Mob r;
    r.m0 = l.m0 * e.m0 + l.m1 * e.m2; r.m1 = l.m0 * e.m1 + l.m1 * e.m3;
    r.m2 = l.m2 * e.m0 + l.m3 * e.m2; r.m3 = l.m2 * e.m1 + l.m3 * e.m3;
    return mob_norm(r);
}
__device__ __forceinline__ Mob mob_shfl_up(const Mob& a, int o) {
    Mob r;
    r.m0 = __shfl_up(a.m0, o, 64); r.m1 = __shfl_up(a.m1, o, 64);
    r.m2 = __shfl_up(a.m2, o, 64); r.m3 = __shfl_up(a.m3, o, 64);
    return r;
}
__device__ __forceinline__ void mob_carry_in(Mob M, Mob* sM, double* p_in, double* q_in) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const Mob e = mob_shfl_up(M, o);
        if (lane >= o) M = mob_mul(M, e);
    }
    Mob prev = mob_shfl_up(M, 1);
    if (lane == 0) prev = Mob{1.0, 0.0, 0.0, 1.0};
    if (lane == 63) sM[w] = M;
    __syncthreads();
    double p = 1.0, q = 0.0;
    for (int k = 0; k < w; ++k) {
        const Mob t = sM[k];
        const double p2 = t.m0 * p + t.m1 * q, q2 = t.m2 * p + t.m3 * q;
        const double m = fmax(fabs(p2), fabs(q2));
        const double s = m > 0.0 ? ldexp(1.0, -ilogb(m)) : 1.0;
        p = p2 * s; q = q2 * s;
    }
    __syncthreads();
    *p_in = prev.m0 * p + prev.m1 * q;
    *q_in = prev.m2 * p + prev.m3 * q;
}

// ---- T = tridiag(L) + sigma I = L U, pivots by the continued fraction u_e = b_e - a_e^2/u_{e-1}
// (a_e = L[e, e-1]), parallelised as a scan of Moebius maps.  One workgroup of 1024 threads.
// bad[0] <- 1 when a pivot is not positive (T not positive definite: caller falls back).
// Tridiagonal band of L, row-parallel over the whole chip (the factorisation below is one workgroup: letting it
// search the CSR rows itself cost 42 us per solve): sub-diagonal, diagonal, super-diagonal in the solver's layout.
__global__ __launch_bounds__(kBlock) void k_tri_band(CsrView A, int c, int stride, double* __restrict__ ba,
                                                     double* __restrict__ bd, double* __restrict__ bu) {
    for (int e = blockIdx.x * kBlock + threadIdx.x; e < A.n; e += gridDim.x * kBlock) {
        double a = 0.0, d = 0.0, u = 0.0;
        for (int p = A.rowptr[e]; p < A.rowptr[e + 1]; ++p) {
            const int col = A.col[p];
            const double x = A.val[p];
            if (col == e) d += x;
            else if (col == e - 1) a += x;
            else if (col == e + 1) u += x;
        }
        const int k = tri_perm(e, c, stride);
        ba[k] = a; bd[k] = d; bu[k] = u;
    }
}

// chain_only: T = (odometry-chain Laplacian) + sigma I, i.e. the diagonal counts the two band neighbours
// only -- the closures are then added back exactly by the Woodbury correction (woodbury.h).
template <int C>
__global__ __launch_bounds__(kTriThreads) void k_tri_factor(int n, const double* __restrict__ ba, const double* __restrict__ bd,
                                                            const double* __restrict__ bu, double sigma, double* tl,
                                                            double* tdinv, double* tcu, int* bad, int chain_only) {
    __shared__ Mob sM[16];
    __shared__ double s_afirst[kTriThreads + 1];
    const int t = threadIdx.x;   // thread t owns the unknowns e = t*C .. t*C + C-1 (zero padded past n)
    double av[C], bv[C];
    Mob M{1.0, 0.0, 0.0, 1.0};
    {   // all 2 C loads in flight at once (always in bounds: the band arrays hold C x 1024 entries), masked afterwards --
        // inside `if (e < n)` the compiler had emitted one predicated load + wait per value, 2 C dependent round trips
        const double* __restrict__ bsel = chain_only ? bu : bd;
        double la[C], lb[C];
#pragma unroll
        for (int i = 0; i < C; ++i) { const int k = i * kTriThreads + t; la[i] = ba[k]; lb[i] = bsel[k]; }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            av[i] = 0.0; bv[i] = 1.0;
            const int e = t * C + i;
            if (e < n) {
                const double a = la[i];
                const double b = chain_only ? -(a + lb[i]) : lb[i];
                av[i] = a; bv[i] = b + sigma;
                M = mob_mul(Mob{bv[i], -a * a, 1.0, 0.0}, M);
            }
        }
    }
    s_afirst[t] = av[0];
    if (t == 0) s_afirst[kTriThreads] = 0.0;
    double p_in, q_in;
    mob_carry_in(M, sM, &p_in, &q_in);     // contains the barriers that publish s_afirst
    double rinv = (t == 0 || p_in == 0.0) ? 0.0 : q_in / p_in;   // 1 / u_{e-1}
    int isbad = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int e = t * C + i, k = i * kTriThreads + t;
        double l = 0.0, dinv = 0.0;
        if (e < n) {
            l = av[i] * rinv;
            const double u = bv[i] - av[i] * av[i] * rinv;
            if (!(u > 0.0)) isbad = 1;
            rinv = 1.0 / u;
            dinv = rinv;
        }
        tl[k] = l; tdinv[k] = dinv;
        bv[i] = dinv;
    }
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const int e = t * C + i, k = i * kTriThreads + t;
        double anext = 0.0;   // a_{e+1}: next element of the chunk (constant index after unrolling) or of the next thread
        if (e + 1 < n) anext = (i + 1 < C) ? av[i + 1 < C ? i + 1 : C - 1] : s_afirst[t + 1];
        tcu[k] = (e < n) ? anext * bv[i] : 0.0;
    }
    if (isbad) *bad = 1;
}

// ---- n > 16 384: the same algorithm over several workgroups ----------------------------------------
// Chunks of kTriBigC = 4 unknowns per thread, Q = ceil(n/4) threads in Q/1024 workgroups, the chunk in
// global scratch (chunk-transposed with stride = padded Q, so still coalesced).  Three launches per
// solve: (1) forward sweeps with a zero carry; every workgroup also publishes the composition of its
// 1 024 chunk maps as ONE affine map; (2) every workgroup composes the maps of all EARLIER workgroups
// itself (<= a few hundred, identical arithmetic everywhere, no inter-workgroup signalling), scans its
// own chunks, fixes them up and runs the backward sweeps; (3) the same from the other end.
// ~13 n doubles of traffic spread over n/4096 CUs.  The LU is computed once per solve by a single
// workgroup (k_tri_factor_big) straight into this layout.
__global__ __launch_bounds__(kTriThreads) void k_tri_factor_big(CsrView A, int stride, double sigma, double* tl, double* tdinv,
                                                                double* tcu, double* as, double* bs, int* bad, int chain_only) {
    __shared__ Mob sM[16];
    const int t = threadIdx.x, n = A.n;
    const int c = (n + kTriThreads - 1) / kTriThreads;     // unknowns per thread HERE (private scratch layout i*1024 + t)
    Mob M{1.0, 0.0, 0.0, 1.0};
    for (int i = 0; i < c; ++i) {
        const int e = t * c + i, k = i * kTriThreads + t;
        double a = 0.0, b = 1.0;
        if (e < n) {
            b = 0.0;
            double up = 0.0;
            for (int p = A.rowptr[e]; p < A.rowptr[e + 1]; ++p) {
                const int col = A.col[p];
                if (col == e) b += A.val[p];
                else if (col == e - 1) a += A.val[p];
                else if (col == e + 1) up += A.val[p];
            }
            if (chain_only) b = -(a + up);
            b += sigma;
            M = mob_mul(Mob{b, -a * a, 1.0, 0.0}, M);
        }
        as[k] = a; bs[k] = b;
    }
    double p_in, q_in;
    mob_carry_in(M, sM, &p_in, &q_in);
    double rinv = (t == 0 || p_in == 0.0) ? 0.0 : q_in / p_in;
    int isbad = 0;
    for (int i = 0; i < c; ++i) {
        const int e = t * c + i, k = i * kTriThreads + t;
        if (e < n) {
            const double a = as[k];
            const double u = bs[k] - a * a * rinv;
            if (!(u > 0.0)) isbad = 1;
            const int ko = tri_perm(e, kTriBigC, stride);
            tl[ko] = a * rinv;
            rinv = 1.0 / u;
            tdinv[ko] = rinv;
            bs[k] = rinv;
        }
    }
    __syncthreads();                     // as[] of the neighbouring thread (same CU: L1 is shared, stores drained)
    for (int i = 0; i < c; ++i) {
        const int e = t * c + i, k = i * kTriThreads + t;
        if (e < n) {
            const int e1 = e + 1;
            const double anext = e1 < n ? as[(e1 % c) * kTriThreads + e1 / c] : 0.0;
            tcu[tri_perm(e, kTriBigC, stride)] = anext * bs[k];
        }
    }
    if (isbad) *bad = 1;
}

// Product of one value per thread over the 1 024 threads (shuffle tree + 16 LDS words).
__device__ __forceinline__ double block_prod_1024(double a, double* sP) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a *= __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sP[threadIdx.x >> 6] = a;
    __syncthreads();
    double p = 1.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) p *= sP[k];
    __syncthreads();
    return p;
}

// The three phases as device functions over explicit buffers, so that the batched form (one right-hand side per
// blockIdx.y: woodbury.h) shares them: ys / pas = chunk scratch (chunk-transposed, stride S), mA/mB/mA2/mB2 =
// one affine map per workgroup and direction, out = the solution.
struct TriBigBuf { double *ys, *pas, *mA, *mB, *mA2, *mB2, *out; };

template <class Rhs>
__device__ __forceinline__ void tri_big_fwd(const LobView& L, const TriBigBuf& B, Rhs rhs, double* sA, double* sB) {
    const int q = blockIdx.x * kTriThreads + threadIdx.x, S = L.stride;
    double run = 0.0, prod = 1.0;
#pragma unroll
    for (int i = 0; i < kTriBigC; ++i) {
        const int k = i * S + q;
        const double l = L.tl[k];
        run = rhs(k) - l * run;
        prod = -l * prod;
        B.ys[k] = run; B.pas[k] = prod;
    }
    // this workgroup's 1 024 chunk maps as one map (A, B): B = value given to 0, A = product of the A's
    const double cin = affine_carry_in<false>(prod, run, sA, sB);
    const double Atot = block_prod_1024(prod, sA);
    if (threadIdx.x == kTriThreads - 1) { B.mA[blockIdx.x] = Atot; B.mB[blockIdx.x] = prod * cin + run; }
}
__device__ __forceinline__ void tri_big_mid(const LobView& L, const TriBigBuf& B, double* sA, double* sB) {
    const int q = blockIdx.x * kTriThreads + threadIdx.x, S = L.stride;
    double c0 = 0.0;                                   // carry entering this workgroup
    for (int g = 0; g < (int)blockIdx.x; ++g) c0 = B.mA[g] * c0 + B.mB[g];
    // own chunk maps are recomputed from the stored sweeps: A = pas at the chunk end, B = ys at the chunk end
    const int kl = (kTriBigC - 1) * S + q;
    const double carry = affine_carry_in<false>(B.pas[kl], B.ys[kl], sA, sB, c0);
    double xr = 0.0, pb = 1.0;
#pragma unroll
    for (int i = kTriBigC - 1; i >= 0; --i) {
        const int k = i * S + q;
        const double cu = L.tcu[k];
        xr = (B.ys[k] + B.pas[k] * carry) * L.tdinv[k] - cu * xr;
        pb = -cu * pb;
        B.ys[k] = xr; B.pas[k] = pb;
    }
    const double cin = affine_carry_in<true>(pb, xr, sA, sB);
    const double Atot = block_prod_1024(pb, sA);
    if (threadIdx.x == 0) { B.mA2[blockIdx.x] = Atot; B.mB2[blockIdx.x] = pb * cin + xr; }
}
__device__ __forceinline__ void tri_big_fin(const LobView& L, const TriBigBuf& B, double* sA, double* sB) {
    const int q = blockIdx.x * kTriThreads + threadIdx.x, S = L.stride;
    double c0 = 0.0;
    for (int g = (int)gridDim.x - 1; g > (int)blockIdx.x; --g) c0 = B.mA2[g] * c0 + B.mB2[g];
    const double carry2 = affine_carry_in<true>(B.pas[q], B.ys[q], sA, sB, c0);   // chunk start = row 0 of the layout
#pragma unroll
    for (int i = 0; i < kTriBigC; ++i) {
        const int k = i * S + q;
        B.out[k] = B.ys[k] + B.pas[k] * carry2;
    }
}
__device__ __forceinline__ TriBigBuf tri_big_buf(const LobView& L) {
    return TriBigBuf{L.ys, L.pas, L.mapA, L.mapB, L.mapA2, L.mapB2, L.wT};
}
__global__ __launch_bounds__(kTriThreads) void k_tri_big_fwd(LobView L) {
    __shared__ double sA[16], sB[16];
    const double* __restrict__ r = L.rT;
    tri_big_fwd(L, tri_big_buf(L), [r](int k) { return r[k]; }, sA, sB);
}
__global__ __launch_bounds__(kTriThreads) void k_tri_big_mid(LobView L) {
    __shared__ double sA[16], sB[16];
    tri_big_mid(L, tri_big_buf(L), sA, sB);
}
__global__ __launch_bounds__(kTriThreads) void k_tri_big_fin(LobView L) {
    __shared__ double sA[16], sB[16];
    tri_big_fin(L, tri_big_buf(L), sA, sB);
}

__global__ __launch_bounds__(kBlock) void k_lob_perm_cols(const int* __restrict__ col, long nnz, int c, int stride, int* __restrict__ colT) {
    for (long p = (long)blockIdx.x * kBlock + threadIdx.x; p < nnz; p += (long)gridDim.x * kBlock) colT[p] = tri_perm(col[p], c, stride);
}

// Reduce the ||r||_1 partials of iteration `it` (ping-pong halves of partR by parity: the update
// kernel that writes iteration it+1's partials is the one that reads these) and publish
// (theta, ||r||_1).  One full wave.
__device__ __forceinline__ void lob_publish(const LobView& L, int it) {
    double a = 0.0;
    const double* pr = L.partR + (size_t)(it & 1) * kMaxGrid;
    for (int i = threadIdx.x; i < L.P_a; i += 64) a += pr[i];
    a = wave_total(a);
    if (threadIdx.x == 0) {
        L.hrec[4 * (size_t)it] = L.st->theta;
        L.hrec[4 * (size_t)it + 1] = a;
        L.hrec[4 * (size_t)it + 2] = lob_tag(L.st->epoch, it);
    }
}

// ---- w = T^-1 r ------------------------------------------------------------------------------
// w = T^-1 rhs for one workgroup; rhs(k) supplies the right-hand side in the chunk-transposed layout.
template <int CMAX, class Rhs>
__device__ __forceinline__ void tri_solve_body(const LobView& L, Rhs rhs, double* __restrict__ out, double* sA, double* sB,
                                               double* s_pa) {
    const int t = threadIdx.x;   // CMAX == L.c: every thread owns exactly CMAX unknowns (zero padded past n)
    double y[CMAX];
    double run = 0.0, prod = 1.0;
#pragma unroll
    for (int i = 0; i < CMAX; ++i) {
        const int k = i * kTriThreads + t;
        const double l = L.tl[k];
        run = rhs(k) - l * run;
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
    for (int i = 0; i < CMAX; ++i) out[i * kTriThreads + t] = y[i] + s_pa[i * kTriThreads + t] * carry2;
}

template <int CMAX>
__global__ __launch_bounds__(kTriThreads) void k_tri_solve(LobView L, int jrel) {
    __shared__ double sA[16], sB[16];
    __shared__ double s_pa[CMAX * kTriThreads];   // carry coefficients (CMAX = 16: 128 of the 160 KB of LDS)
    (void)jrel;
    const double* __restrict__ r = L.rT;
    tri_solve_body<CMAX>(L, [r](int k) { return r[k]; }, L.wT, sA, sB, s_pa);
}

// ---- Lw = L w and the inner products -----------------------------------------------------------
// sums: 0 xx 1 xw 2 xp 3 ww 4 wp 5 pp | 6 xLx 7 xLw 8 xLp 9 wLw 10 wLp 11 pLp | 12 Sx 13 Sw 14 Sp
struct OpLob {
    LobView L;
    double s[kLobNS];
    __device__ __forceinline__ void begin(double*) {
#pragma unroll
        for (int q = 0; q < kLobNS; ++q) s[q] = 0.0;
    }
    __device__ __forceinline__ double gather(const double* __restrict__ x, int c) const { return x[c]; }
    __device__ __forceinline__ void row(int r, double lw) {
        const double x = L.x[r], p = L.p[r], lx = L.Lx[r], lp = L.Lp[r];
        const double w = L.wT[tri_perm(r, L.c, L.stride)];
        L.Lw[r] = lw;
        s[0] += x * x; s[1] += x * w; s[2] += x * p; s[3] += w * w; s[4] += w * p; s[5] += p * p;
        s[6] += x * lx; s[7] += x * lw; s[8] += x * lp; s[9] += w * lw; s[10] += w * lp; s[11] += p * lp;
        s[12] += x; s[13] += w; s[14] += p;
    }
    __device__ __forceinline__ void end(double*) {
        __shared__ double red[kBlock / 64][kLobNS];
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int q = 0; q < kLobNS; ++q) {
            const double tq = wave_total(s[q]);
            if ((threadIdx.x & 63) == 0) red[w][q] = tq;
        }
        __syncthreads();
        if (threadIdx.x < kLobNS) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kBlock / 64; ++k) a += red[k][threadIdx.x];
            L.part[(size_t)threadIdx.x * kMaxGrid + blockIdx.x] = a;
        }
    }
};

#ifdef MACHIP_EXPERIMENTS
// Column-panel form of the product (panel.h, round 4: the diagonally preconditioned mode on large random graphs): k_pan_mul<RPT, RAW>
// has left one partial product per (row, panel); this kernel adds them in panel order -- Lw[r] -- and takes the 15 inner products
// exactly as the fused SpMV kernels do (OpLob::row / end).  256 threads per workgroup (OpLob's scratch), partials per workgroup.
__global__ __launch_bounds__(kBlock) void k_pan_find(OpLob op, const double* __restrict__ ypart, int NP) {
    op.begin(nullptr);
    const int n = op.L.n;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        double w = 0.0;
        for (int p0 = 0; p0 < NP; p0 += 16) {     // sixteen panels in flight; added in panel order
            double y[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = ypart[(size_t)min(p0 + q, NP - 1) * n + r];
#pragma unroll
            for (int q = 0; q < 16; ++q) w += (p0 + q < NP) ? y[q] : 0.0;
        }
        op.row(r, w);
    }
    op.end(nullptr);
}
#endif

// ---- 3x3 Rayleigh-Ritz on span{x, w - mean, p - mean} -------------------------------------------
struct LobCoef { double z0, z1, z2, theta, mx, mw, mp; int bad; };

// Fallback eigen-solver (cyclic Jacobi) for the rare case the Rayleigh-quotient iteration below did
// not land on the smallest eigenvalue.  Fully unrolled: every index is a compile-time constant, so the
// matrices stay in registers (a kernel that touches scratch memory pays for it at every launch).
template <int A, int B>
__device__ __forceinline__ void jacobi3_rotate(double (&C)[3][3], double (&V)[3][3]) {
    const double apq = C[A][B];
    if (fabs(apq) <= 1e-300) return;
    const double tau = (C[B][B] - C[A][A]) / (2.0 * apq);
    const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    const double cs = rsqrt(1.0 + tt * tt), sn = tt * cs;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double cqa = C[q][A], cqb = C[q][B];
        C[q][A] = cs * cqa - sn * cqb; C[q][B] = sn * cqa + cs * cqb;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double caq = C[A][q], cbq = C[B][q];
        C[A][q] = cs * caq - sn * cbq; C[B][q] = sn * caq + cs * cbq;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double vqa = V[q][A], vqb = V[q][B];
        V[q][A] = cs * vqa - sn * vqb; V[q][B] = sn * vqa + cs * vqb;
    }
}
__device__ __forceinline__ void jacobi3_smallest(double c00, double c01, double c02, double c11, double c12, double c22,
                                                 double* theta, double* y0, double* y1, double* y2) {
    double C[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}};
    double V[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
#pragma unroll
    for (int sweep = 0; sweep < 6; ++sweep) {
        jacobi3_rotate<0, 1>(C, V);
        jacobi3_rotate<0, 2>(C, V);
        jacobi3_rotate<1, 2>(C, V);
    }
    // selects, not branches: a data-dependent column index would put V in scratch memory
    const bool s1 = C[1][1] < C[0][0];
    double th = s1 ? C[1][1] : C[0][0];
    double a0 = s1 ? V[0][1] : V[0][0], a1 = s1 ? V[1][1] : V[1][0], a2 = s1 ? V[2][1] : V[2][0];
    const bool s2 = C[2][2] < th;
    th = s2 ? C[2][2] : th;
    a0 = s2 ? V[0][2] : a0; a1 = s2 ? V[1][2] : a1; a2 = s2 ? V[2][2] : a2;
    *theta = th; *y0 = a0; *y1 = a1; *y2 = a2;
}

// All scalars: nothing here may be indexed dynamically (that would put it in scratch memory).
__device__ __forceinline__ LobCoef lob_rayleigh_ritz(const double* s, int n, bool havep) {
    LobCoef o;
    const double dn = (double)n;
    const double rn = 1.0 / dn;
    o.mx = s[12] * rn; o.mw = s[13] * rn; o.mp = s[14] * rn; o.bad = 0;
    const double G00 = s[0] - dn * o.mx * o.mx, G01 = s[1] - dn * o.mx * o.mw, G02 = s[2] - dn * o.mx * o.mp;
    const double G11 = s[3] - dn * o.mw * o.mw, G12 = s[4] - dn * o.mw * o.mp, G22 = s[5] - dn * o.mp * o.mp;
    o.z0 = G00 > 0.0 ? rsqrt(G00) : 0.0; o.z1 = 0.0; o.z2 = 0.0; o.theta = G00 > 0.0 ? s[6] / G00 : 0.0;
    if (!(G00 > 0.0) || !(G11 > 0.0)) { o.bad = 1; return o; }
    bool k3 = havep && G22 > 0.0;
    const double d0 = rsqrt(G00), d1 = rsqrt(G11), d2 = k3 ? rsqrt(G22) : 0.0;
    // scaled Gram matrix (unit diagonal) = R^T R, R upper triangular with r00 = 1
    const double r01 = G01 * d0 * d1, r02 = G02 * d0 * d2, g12 = G12 * d1 * d2;
    const double q11 = 1.0 - r01 * r01;
    if (!(q11 > 1e-14)) { o.bad = 1; return o; }       // w parallel to x: nothing left to gain
    const double ir11 = rsqrt(q11);
    const double r12 = (g12 - r01 * r02) * ir11;
    const double q22 = 1.0 - r02 * r02 - r12 * r12;
    if (k3 && !(q22 > 1e-12)) k3 = false;              // p (nearly) dependent: restart the recurrence
    const double ir22 = k3 ? rsqrt(q22) : 0.0;
    // Ri = R^-1 (upper triangular); columns of S D Ri are orthonormal
    const double i01 = -r01 * ir11;
    const double i12 = k3 ? -r12 * ir11 * ir22 : 0.0;
    const double i02 = k3 ? -(r01 * i12 + r02 * ir22) : 0.0;
    // As = D A D
    const double A00 = s[6] * d0 * d0, A01 = s[7] * d0 * d1, A02 = s[8] * d0 * d2;
    const double A11 = s[9] * d1 * d1, A12 = s[10] * d1 * d2, A22 = s[11] * d2 * d2;
    // C = Ri^T As Ri, column by column: u_b = As Ri[:, b]
    //   Ri[:,0] = (1,0,0); Ri[:,1] = (i01, ir11, 0); Ri[:,2] = (i02, i12, ir22)
    const double u10 = A00 * i01 + A01 * ir11, u11 = A01 * i01 + A11 * ir11, u12 = A02 * i01 + A12 * ir11;
    const double u20 = A00 * i02 + A01 * i12 + A02 * ir22, u21 = A01 * i02 + A11 * i12 + A12 * ir22,
                 u22 = A02 * i02 + A12 * i12 + A22 * ir22;
    const double C00 = A00, C01 = u10, C02 = u20;
    const double C11 = i01 * u10 + ir11 * u11;
    const double C12 = i01 * u20 + ir11 * u21;
    const double C22 = i02 * u20 + i12 * u21 + ir22 * u22;
    (void)u12;
    // smallest eigenpair (theta, y) of C.  Two vectors: one Jacobi rotation is exact.  Three:
    // Rayleigh-quotient iteration from e0 (x is the previous Ritz vector, so C00 is close to the wanted
    // eigenvalue) with adjugate solves -- no divisions, well defined even at an exact eigenvalue --
    // then a definiteness check of C - (theta - delta) I by its leading minors.
    double y0 = 1.0, y1 = 0.0, y2 = 0.0, theta = C00;
    if (!k3) {
        if (fabs(C01) > 1e-300) {
            const double tau = (C11 - C00) / (2.0 * C01);
            const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            const double cs = rsqrt(1.0 + tt * tt), sn = tt * cs;
            const double e0 = C00 - tt * C01, e1 = C11 + tt * C01;
            if (e0 <= e1) { theta = e0; y0 = cs; y1 = -sn; }
            else { theta = e1; y0 = sn; y1 = cs; }
        } else if (C11 < C00) { theta = C11; y0 = 0.0; y1 = 1.0; }
    } else {
        const double cmax = fmax(fabs(C00), fmax(fabs(C11), fabs(C22)));
#pragma unroll 1
        for (int it = 0; it < 4; ++it) {
            const double b00 = C00 - theta, b11 = C11 - theta, b22 = C22 - theta;
            const double a00 = b11 * b22 - C12 * C12, a01 = C02 * C12 - C01 * b22, a02 = C01 * C12 - C02 * b11;
            const double a11 = b00 * b22 - C02 * C02, a12 = C01 * C02 - b00 * C12, a22 = b00 * b11 - C01 * C01;
            double z0 = a00 * y0 + a01 * y1 + a02 * y2;
            double z1 = a01 * y0 + a11 * y1 + a12 * y2;
            double z2 = a02 * y0 + a12 * y1 + a22 * y2;
            const double n2 = z0 * z0 + z1 * z1 + z2 * z2;
            if (!(n2 > 0.0) || !(n2 < 1e300)) break;
            const double inv = rsqrt(n2);
            z0 *= inv; z1 *= inv; z2 *= inv;
            const double t0 = C00 * z0 + C01 * z1 + C02 * z2;
            const double t1 = C01 * z0 + C11 * z1 + C12 * z2;
            const double t2 = C02 * z0 + C12 * z1 + C22 * z2;
            const double nt = z0 * t0 + z1 * t1 + z2 * t2;
            const bool done = fabs(nt - theta) <= 4e-16 * cmax;
            y0 = z0; y1 = z1; y2 = z2; theta = nt;
            if (done) break;
        }
        const double sft = theta - fmax(1e-9 * fabs(theta), 1e-13 * cmax);
        const double m00 = C00 - sft, m11 = C11 - sft, m22 = C22 - sft;
        const double min2 = m00 * m11 - C01 * C01;
        const double det = m00 * (m11 * m22 - C12 * C12) - C01 * (C01 * m22 - C12 * C02) + C02 * (C01 * C12 - m11 * C02);
        if (!(m00 > 0.0 && min2 > 0.0 && det > 0.0)) {   // not the smallest one
            jacobi3_smallest(C00, C01, C02, C11, C12, C22, &theta, &y0, &y1, &y2);
        }
    }
    // back to the coefficients of (x, w, p): z = D Ri y
    double zz0 = (y0 + i01 * y1 + i02 * y2) * d0;
    double zz1 = (ir11 * y1 + i12 * y2) * d1;
    double zz2 = (ir22 * y2) * d2;
    if (zz0 < 0.0) { zz0 = -zz0; zz1 = -zz1; zz2 = -zz2; }
    if (!(zz0 == zz0) || !(zz1 == zz1) || !(zz2 == zz2)) { o.bad = 1; return o; }
    o.z0 = zz0; o.z1 = zz1; o.z2 = k3 ? zz2 : 0.0; o.theta = theta;
    return o;
}

// ---- x, p, Lx, Lp, r of the next iteration -----------------------------------------------------
// JAC: diagonal (Jacobi) preconditioner -- w = r / diag(L) is formed right here (L.tdinv = 1 / diag in natural order,
// c = 1: the chunk-transposed layout is the identity), so an iteration is two launches: SpMV + inner products, update.
template <bool JAC = false>
__global__ __launch_bounds__(kBlock) void k_lob_update(LobView L, int jrel) {
    __shared__ double sc[8];
    __shared__ double sm[4];
    const int itn = L.st->it0 + jrel + 1;          // index of the iterate this launch produces
    // record of the iterate entering this launch: wave 1 of workgroup 0, which would otherwise idle
    // at the barrier while wave 0 runs the Rayleigh-Ritz step (st->theta is rewritten after the barrier)
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {
        double a = 0.0;
        const double* pr = L.partR + (size_t)((itn - 1) & 1) * kMaxGrid;
        for (int i = threadIdx.x - 64; i < L.P_a; i += 64) a += pr[i];
        a = wave_total(a);
        if (threadIdx.x == 64) {
            L.hrec[4 * (size_t)(itn - 1)] = L.st->theta;
            L.hrec[4 * (size_t)(itn - 1) + 1] = a;
            L.hrec[4 * (size_t)(itn - 1) + 2] = lob_tag(L.st->epoch, itn - 1);
        }
    }
    if (threadIdx.x < 64) {
        double s[kLobNS];
        {   // all partial loads in flight at once, masked afterwards (a loop with a run-time trip count compiles into one
            // dependent cold round trip per value -- see pipe_prologue_wave0 in kernels.h); same order of additions
#pragma unroll
            for (int q = 0; q < kLobNS; ++q) s[q] = 0.0;
            for (int base = 0; base < L.P_c; base += 256) {
                double v[kLobNS][4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int q = 0; q < kLobNS; ++q) v[q][c] = L.part[(size_t)q * kMaxGrid + base + threadIdx.x + 64 * c];   // < kMaxGrid
#pragma unroll
                for (int q = 0; q < kLobNS; ++q)
#pragma unroll
                    for (int c = 0; c < 4; ++c) s[q] += (base + (int)threadIdx.x + 64 * c < L.P_c) ? v[q][c] : 0.0;
            }
            wave_total_n<kLobNS>(s);
        }
        if (threadIdx.x == 0) {
            const LobCoef co = lob_rayleigh_ritz(s, L.n, L.st->havep0 != 0 || jrel > 0);
            sc[0] = co.z0; sc[1] = co.z1; sc[2] = co.z2; sc[3] = co.theta; sc[4] = co.mx; sc[5] = co.mw; sc[6] = co.mp;
            sc[7] = co.bad ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        L.st->theta = sc[3];
        if (sc[7] != 0.0) L.st->bad = 1;
    }
    const double z0 = sc[0], z1 = sc[1], z2 = sc[2], th = sc[3], mx = sc[4], mw = sc[5], mp = sc[6];
    double l1 = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < L.n; r += gridDim.x * kBlock) {
        const int k = tri_perm(r, L.c, L.stride);
        const double w = L.wT[k] - mw;
        const double pn = z1 * w + z2 * (L.p[r] - mp);
        const double lpn = z1 * L.Lw[r] + z2 * L.Lp[r];
        const double xn = z0 * (L.x[r] - mx) + pn;
        const double lxn = z0 * L.Lx[r] + lpn;
        L.p[r] = pn; L.Lp[r] = lpn; L.x[r] = xn; L.Lx[r] = lxn;
        const double res = lxn - th * xn;
        if (JAC) L.wT[k] = res * L.tdinv[k]; else L.rT[k] = res;
        l1 += fabs(res);
    }
    l1 = block_sum(l1, sm);
    if (threadIdx.x == 0) L.partR[(size_t)(itn & 1) * kMaxGrid + blockIdx.x] = l1;
}

// ---- column-panel product, TWO launches per iteration (round 4, late) ---------------------------------------------------------
// k_pan_mul<RPT, RAW> leaves the partial products per (row, panel) and w^T L w per cell; this kernel is k_pan_find + k_lob_update<JAC>
// in one: prologue (15 sums -> Rayleigh-Ritz), then per row  Lw = sum of the partials in panel order,  the update of x / p / Lx / Lp,
// the residual, w' = r / diag -- and the sums of the NEXT iteration from the values just stored: by the symmetry of L
//     x'^T L w' = w'^T (L x'),   p'^T L w' = w'^T (L p'),
// so only w'^T L w' has to wait for the next product (it comes with it).  The sums are double-buffered by the parity of the iterate
// (a fast workgroup writes the next sums while a slow one still reads the current ones).  Same Rayleigh-Ritz, same update arithmetic
// as k_lob_update<true>; the inner products differ from k_pan_find's by roundings (other association), not by definition.
//   sums: 0 xx 1 xw 2 xp 3 ww 4 wp 5 pp | 6 xLx 7 xLw 8 xLp 9 wLw 10 wLp 11 pLp | 12 Sx 13 Sw 14 Sp
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_lob_update_pan(LobView L, const double* __restrict__ ypart, int NP, const double* __restrict__ partW, int P_w, int jrel, int par) {
    __shared__ double sc[8];
    __shared__ double red[BLOCK / 64][kLobNS + 1];
    // par = parity of the iterate this launch produces (itn = it0 + jrel + 1; the host knows it0's parity when it enqueues the chunk:
    // the partial sums can be requested without waiting for the state word)
    const int P_u = (int)gridDim.x;
    const double* __restrict__ pin = L.part + (size_t)(par ^ 1) * (kLobNS * kMaxGrid);
    double* __restrict__ pout = L.part + (size_t)par * (kLobNS * kMaxGrid);
    const int n = L.n;
    // The first row of every thread is requested BEFORE the prologue's barrier (the grid is sized for one or two rows per thread): the
    // partial products and the six vectors travel while wave 0 reduces the sums and runs the Rayleigh-Ritz step.
    constexpr int PB = 16;
    const int r0 = blockIdx.x * BLOCK + threadIdx.x, rc = min(r0, n - 1);
    double y0[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) y0[q] = ypart[(size_t)min(q, NP - 1) * n + rc];
    double f_w = L.wT[rc], f_p = L.p[rc], f_lp = L.Lp[rc], f_x = L.x[rc], f_lx = L.Lx[rc], f_d = L.tdinv[rc];
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {      // record of the iterate entering this launch (cf. k_lob_update)
        const int itn = L.st->it0 + jrel + 1;
        double a = 0.0;
        const double* pr = L.partR + (size_t)(par ^ 1) * kMaxGrid;
        for (int i = threadIdx.x - 64; i < P_u; i += 64) a += pr[i];
        a = wave_total(a);
        if (threadIdx.x == 64) {
            L.hrec[4 * (size_t)(itn - 1)] = L.st->theta;
            L.hrec[4 * (size_t)(itn - 1) + 1] = a;
            L.hrec[4 * (size_t)(itn - 1) + 2] = lob_tag(L.st->epoch, itn - 1);
        }
    }
    if (threadIdx.x < 64) {
        double s[kLobNS];
#pragma unroll
        for (int q = 0; q < kLobNS; ++q) s[q] = 0.0;
        for (int base = 0; base < max(P_u, P_w); base += 256) {      // (both grids hold at most 256 workgroups by plan: one round)
            double v[kLobNS][4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < kLobNS; ++q) {
                    const int i = min(base + (int)threadIdx.x + 64 * c, kMaxGrid - 1);
                    v[q][c] = q == 9 ? partW[i] : pin[(size_t)q * kMaxGrid + i];
                }
#pragma unroll
            for (int q = 0; q < kLobNS; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) s[q] += (base + (int)threadIdx.x + 64 * c < (q == 9 ? P_w : P_u)) ? v[q][c] : 0.0;
        }
        wave_total_n<kLobNS>(s);
        if (threadIdx.x == 0) {
            const LobCoef co = lob_rayleigh_ritz(s, L.n, L.st->havep0 != 0 || jrel > 0);
            sc[0] = co.z0; sc[1] = co.z1; sc[2] = co.z2; sc[3] = co.theta; sc[4] = co.mx; sc[5] = co.mw; sc[6] = co.mp;
            sc[7] = co.bad ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        L.st->theta = sc[3];
        if (sc[7] != 0.0) L.st->bad = 1;
    }
    const double z0 = sc[0], z1 = sc[1], z2 = sc[2], th = sc[3], mx = sc[4], mw = sc[5], mp = sc[6];
    double a[kLobNS + 1];
#pragma unroll
    for (int q = 0; q <= kLobNS; ++q) a[q] = 0.0;
    for (int r = r0; r < n; r += gridDim.x * BLOCK) {
        double lw = 0.0;
        if (r != r0) {
#pragma unroll
            for (int q = 0; q < PB; ++q) y0[q] = ypart[(size_t)min(q, NP - 1) * n + r];
            f_w = L.wT[r]; f_p = L.p[r]; f_lp = L.Lp[r]; f_x = L.x[r]; f_lx = L.Lx[r]; f_d = L.tdinv[r];
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) lw += (q < NP) ? y0[q] : 0.0;      // added in panel order (k_pan_fin's loop)
        for (int p0 = PB; p0 < NP; p0 += PB) {
            double y[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) y[q] = ypart[(size_t)min(p0 + q, NP - 1) * n + r];
#pragma unroll
            for (int q = 0; q < PB; ++q) lw += (p0 + q < NP) ? y[q] : 0.0;
        }
        const double w = f_w - mw;
        const double pn = z1 * w + z2 * (f_p - mp);
        const double lpn = z1 * lw + z2 * f_lp;
        const double xn = z0 * (f_x - mx) + pn;
        const double lxn = z0 * f_lx + lpn;
        L.p[r] = pn; L.Lp[r] = lpn; L.x[r] = xn; L.Lx[r] = lxn;
        const double res = lxn - th * xn;
        const double wn = res * f_d;
        L.wT[r] = wn;
        a[0] += xn * xn; a[1] += xn * wn; a[2] += xn * pn; a[3] += wn * wn; a[4] += wn * pn; a[5] += pn * pn;
        a[6] += xn * lxn; a[7] += wn * lxn; a[8] += xn * lpn; a[10] += wn * lpn; a[11] += pn * lpn;
        a[12] += xn; a[13] += wn; a[14] += pn;
        a[kLobNS] += fabs(res);
    }
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q <= kLobNS; ++q) {
        const double tq = wave_total(a[q]);
        if ((threadIdx.x & 63) == 0) red[wv][q] = tq;
    }
    __syncthreads();
    if (threadIdx.x <= kLobNS) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < BLOCK / 64; ++k) t += red[k][threadIdx.x];
        if (threadIdx.x < kLobNS) pout[(size_t)threadIdx.x * kMaxGrid + blockIdx.x] = t;
        else L.partR[(size_t)par * kMaxGrid + blockIdx.x] = t;
    }
}

#ifdef MACHIP_EXPERIMENTS
// 1 / diag(L) in natural order (Jacobi preconditioner; any CSR: the row is searched for its diagonal entry)
__global__ __launch_bounds__(kBlock) void k_jac_dinv(CsrView A, double* __restrict__ dinv, int* bad) {
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < A.n; r += gridDim.x * kBlock) {
        double d = 0.0;
        for (int p = A.rowptr[r]; p < A.rowptr[r + 1]; ++p) if (A.col[p] == r) d += A.val[p];
        if (!(d > 0.0)) { *bad = 1; d = 1.0; }
        dinv[r] = 1.0 / d;
    }
}
#endif

// First residual of a (re)started recurrence: x = yvec (unit, mean free), Lx = w2 = L yvec.
template <bool JAC = false, bool SUMS = false>     // SUMS: also the inner products k_lob_update_pan expects from its predecessor (p = 0)
__global__ __launch_bounds__(kBlock) void k_lob_start(LobView L, const double* __restrict__ xin, const double* __restrict__ lxin,
                                                      const double* __restrict__ rq, int it0, unsigned int epoch) {
    __shared__ double sm[4];
    const double th = *rq;
    double l1 = 0.0;
    double a[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // xx xw ww xLx xLw Sx Sw
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < L.n; r += gridDim.x * kBlock) {
        const double x = xin[r], lx = lxin[r];
        L.x[r] = x; L.Lx[r] = lx; L.p[r] = 0.0; L.Lp[r] = 0.0;
        const double res = lx - th * x;
        if (JAC) L.wT[r] = res * L.tdinv[r]; else L.rT[tri_perm(r, L.c, L.stride)] = res;
        if (JAC && SUMS) {
            const double wn = res * L.tdinv[r];
            a[0] += x * x; a[1] += x * wn; a[2] += wn * wn; a[3] += x * lx; a[4] += wn * lx; a[5] += x; a[6] += wn;
        }
        l1 += fabs(res);
    }
    if (JAC && SUMS) {
        double* __restrict__ pout = L.part + (size_t)(it0 & 1) * (kLobNS * kMaxGrid);
        constexpr int slot[7] = {0, 1, 3, 6, 7, 12, 13};
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const double t = block_sum(a[q], sm);
            __syncthreads();
            if (threadIdx.x == 0) pout[(size_t)slot[q] * kMaxGrid + blockIdx.x] = t;
        }
        if (threadIdx.x == 0) {
            constexpr int zero[7] = {2, 4, 5, 8, 10, 11, 14};
#pragma unroll
            for (int q = 0; q < 7; ++q) pout[(size_t)zero[q] * kMaxGrid + blockIdx.x] = 0.0;
        }
    }
    l1 = block_sum(l1, sm);
    if (threadIdx.x == 0) {
        L.partR[(size_t)(it0 & 1) * kMaxGrid + blockIdx.x] = l1;
        if (blockIdx.x == 0) { L.st->theta = th; L.st->it0 = it0; L.st->havep0 = 0; L.st->bad = 0; L.st->epoch = epoch; }
    }
}

// End of a chunk: last record, advance the chunk base, raise the host flag.
__global__ void k_lob_tail(LobView L, int steps) {
    const int it = L.st->it0 + steps;
    lob_publish(L, it);
    if (threadIdx.x == 0) {
        L.st->it0 = it;
        L.st->havep0 = 1;
        __threadfence_system();
        *L.hflag = ((unsigned long long)L.st->epoch << 32) | (unsigned long long)(unsigned int)it | (L.st->bad ? 0x80000000ull : 0ull);
    }
}

}  // namespace machip
