// mac_amd/csrc/persist.h -- single-workgroup, LDS-resident Lanczos for small graphs.
//
// On the pose graphs of BASELINE.json configs[2] and [4] (intel: n = 1 728, sphere2500: n = 2 500;
// also kitti_05) the whole matrix is a few tens of KB and a Lanczos step of the multi-workgroup
// path is nothing but launch latency (~4.5 us for ~0.2 us of work).  Here one workgroup (8 waves)
// keeps the gather operand and the off-band entries in LDS (148 of the CU's 160 KB), every thread
// owns up to 6 rows in registers, and a chunk of Lanczos steps runs inside ONE launch: the two
// global reductions of a step are workgroup reductions (DPP wave totals + 16 LDS words + a barrier).
// Classic three-term form (beta_j from the vector itself), so it also serves restarts.  Same
// records / flag protocol towards the host as k_pipe_tail, same basis V in HBM for the Ritz vector.
#pragma once
#include "kernels.h"

namespace machip {

#ifdef PERSIST_CLOCKS   // tools/ubench_persist.hip: shader-clock stamps of thread 0
#define PCLK(cond, i) do { if (t == 0 && (cond)) L.clk[i] = clock64(); } while (0)
#else
#define PCLK(cond, i) do { } while (0)
#endif

constexpr int kPersistThreads = 512;      // 8 waves, two per SIMD.  The kernel is FP64-issue / latency bound: a second
                                           // wave per SIMD hides the dependent-op latency, more waves only repeat the
                                           // reduction epilogues (intel / sphere2500 FW it/s: 256 threads 329 / 702,
                                           // 512 threads 430 / 990, 1 024 threads 397 / 746)
constexpr int kPersistPool = 148 * 1024;   // bytes of LDS for val, col, rowptr and the gather operand
constexpr int kPersistMaxSteps = 256;      // steps per launch (records staged in LDS)
constexpr int kPersistMaxRows = 6;        // rows per thread held in registers (instantiated for 4, 8, 12): n <= 3072;
                                           // beyond that the register file spills and one CU's FP64 issue rate loses
                                           // to the multi-workgroup path (n = 4 661: 4.0 us/step either way)

template <typename T>
struct PersistViewT {
    int n;
    LanState* st;
    double* u;      // un-normalised next Lanczos vector (state between chunks)
    double* vprev;  // v_{J-1}
    T* V;           // basis, column-major (fp32 in the mixed-precision mode)
    double* tri;    // (alpha_j, beta_j, ||v_j||_1) triples
    double* htri;   // pinned mirror
    unsigned long long* hflag;
#ifdef PERSIST_CLOCKS
    long long* clk;   // tools/ubench_persist.hip: phase stamps of thread 0
#endif
};
using PersistView = PersistViewT<double>;

// nc_max: upper bound of the entries outside the tridiagonal band (the band itself lives in registers)
inline bool persist_fits(int n, long nc_max) {
    return n <= kPersistThreads * kPersistMaxRows && nc_max >= 0 &&
           (size_t)nc_max * 20 + (size_t)n * 8 + ((size_t)n + 2) * 4 + 64 <= (size_t)kPersistPool;   // col 4 + val 8 + product 8 per overflow entry
}

// the filtered variant (k_lan_persist<.., CHEB = true>) keeps a second copy of the operand (double buffering)
inline bool persist_fits_cheb(int n, long nc_max) {
    return n <= kPersistThreads * kPersistMaxRows && nc_max >= 0 &&
           (size_t)nc_max * 20 + (size_t)n * 16 + ((size_t)n + 2) * 4 + 64 <= (size_t)kPersistPool;
}

// Chebyshev filter of the filtered variant: the Lanczos operator is C = -T_d(M), M = c1 L - c0 I mapping [a, b] onto
// [-1, 1] (solver.h chooses a > lambda_2 from a rigorous upper bound, b >= lambda_max, d even).
struct PersistCheb { int deg = 0; double c0 = 0.0, c1 = 0.0; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// queue (s_waitcnt vmcnt(0)), i.e. it would wait ~1 us per step for the basis column v_j just stored
// to HBM, which nothing in this kernel ever reads back.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The eight wave totals of a value, added as a balanced tree (three dependent additions instead of eight; round 3:
// tools/ubench_persist.hip had 420 of a step's 3 200 cycles in the serial sums of the first reduction).
static_assert(kPersistThreads / 64 == 8, "persist_tree8 adds exactly eight wave totals");
__device__ __forceinline__ double persist_tree8(const double* r) {
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// Sum two values over the workgroup; `red` = 3 x kPersistThreads/64 doubles, two such buffers used alternately.
__device__ __forceinline__ void persist_sum2(double& a, double& b, double* red) {
    { double v[2] = {a, b}; wave_total_n<2>(v); a = v[0]; b = v[1]; }
    const int w = threadIdx.x >> 6;
    constexpr int W = kPersistThreads / 64;
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[W + w] = b; }
    lds_barrier();
    a = persist_tree8(red); b = persist_tree8(red + W);
}

// One value (the alpha reduction of a step).
__device__ __forceinline__ void persist_sum1(double& a, double* red) {
    a = wave_total(a);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = a;
    lds_barrier();
    a = persist_tree8(red);
}
// Three values (chunk end; the first two exactly as persist_sum2 adds them).
__device__ __forceinline__ void persist_sum3(double& a, double& b, double& c, double* red) {
    { double v[3] = {a, b, c}; wave_total_n<3>(v); a = v[0]; b = v[1]; c = v[2]; }
    const int w = threadIdx.x >> 6;
    constexpr int W = kPersistThreads / 64;
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[W + w] = b; red[2 * W + w] = c; }
    lds_barrier();
    a = persist_tree8(red); b = persist_tree8(red + W); c = persist_tree8(red + 2 * W);
}

// Per-matrix packed form of what a launch of k_lan_persist keeps in registers and LDS, built ONCE per solve by
// k_persist_pack.  (Rounds 1-2 walked the CSR inside every launch: dependent rowptr -> col -> val loads per row, a prefix
// scan and a second walk -- 11.5 us at intel, 20 us at sphere2500 per launch (tools/ubench_persist.hip), 12-16 % of a
// 64-step chunk.  A launch now issues one round of coalesced loads.)
constexpr int kPersistPad = kPersistThreads * kPersistMaxRows;    // plane stride of the packed arrays
struct PersistPack {
    double* band;   // 5 planes x kPersistPad: diagonal, sub-, super-diagonal, first and second off-band value of every row
    int* cc;        // 2 planes x kPersistPad: columns of those two off-band entries
    int* crow;      // kPersistPad + 1: offsets of the rows' further off-band entries (crow[n] = their number)
    int* ccol;      // ... their columns and values, row by row in CSR order
    double* cval;
};
constexpr size_t kPersistPackEntries = (size_t)kPersistPool / 20 + 8;    // capacity of ccol / cval (persist_fits bounds nc)

__global__ __launch_bounds__(1024) void k_persist_pack(CsrView A, PersistPack P) {
    __shared__ int s_scan[1024];
    const int t = threadIdx.x, n = A.n;
    for (int r = t; r < kPersistPad; r += 1024) {
        double dgd = 0.0, lod = 0.0, upd = 0.0, v0 = 0.0, v1 = 0.0;
        int c0 = 0, c1 = 0, c = 0;
        if (r < n) {
            for (int p = A.rowptr[r]; p < A.rowptr[r + 1]; ++p) {
                const int col = A.col[p];
                const double x = A.val[p];
                if (col == r) dgd += x;
                else if (col == r - 1) lod += x;
                else if (col == r + 1) upd += x;
                else {
                    if (c == 0) { c0 = col; v0 = x; }
                    else if (c == 1) { c1 = col; v1 = x; }
                    ++c;
                }
            }
        }
        P.band[r] = dgd; P.band[kPersistPad + r] = lod; P.band[2 * kPersistPad + r] = upd;
        P.band[3 * kPersistPad + r] = v0; P.band[4 * kPersistPad + r] = v1;
        P.cc[r] = c0; P.cc[kPersistPad + r] = c1;
        P.crow[r + 1] = max(0, c - 2);
    }
    if (t == 0) P.crow[0] = 0;
    __syncthreads();
    // inclusive prefix sum of crow[1..n]: contiguous segment per thread, then a scan of the segment totals
    {
        const int seg = (n + 1023) / 1024;
        const int b0 = 1 + t * seg, e0 = min(n + 1, b0 + seg);
        int tot = 0;
        for (int i = b0; i < e0; ++i) tot += P.crow[i];
        s_scan[t] = tot;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int add = t >= o ? s_scan[t - o] : 0;
            __syncthreads();
            s_scan[t] += add;
            __syncthreads();
        }
        int run = s_scan[t] - tot;
        for (int i = b0; i < e0; ++i) { run += P.crow[i]; P.crow[i] = run; }
        __syncthreads();
    }
    for (int r = t; r < n; r += 1024) {
        int q = P.crow[r], c = 0;
        for (int p = A.rowptr[r]; p < A.rowptr[r + 1]; ++p) {
            const int col = A.col[p];
            if (col < r - 1 || col > r + 1) {
                if (c >= 2) { P.ccol[q] = col; P.cval[q] = A.val[p]; ++q; }
                ++c;
            }
        }
    }
}

template <typename T>
__global__ void k_persist_begin(PersistViewT<T> L, int epoch, const double* __restrict__ src = nullptr) {
    // (src: the start vector of the sequence, copied into the state u here instead of by a copy kernel of its own)
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < L.n; r += gridDim.x * blockDim.x) { L.vprev[r] = 0.0; if (src) L.u[r] = src[r]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { L.st->jA = 0; L.st->jN = 0; L.st->epoch = epoch; }
}

// Matrix layout inside the workgroup (pose graph = odometry chain + loop closures): the tridiagonal
// band of every owned row (diag, sub, super) sits in REGISTERS and its operands v[r-1], v[r], v[r+1]
// are consecutive LDS words for consecutive lanes -- conflict free.  Only the entries outside the band
// (the closures) are split: the first two of every row also sit in registers, further ones -- rare hub
// rows -- are kept as a CSR in LDS next to the gather operand.  (A plain
// thread-per-row CSR in LDS was tried first: rows of ~4 entries put the 64 lanes of a load on 4 banks
// and the step cost 4 us, LDS-bound.)
// T = float (mixed-precision mode): the row data, the operand in LDS, the vectors in registers and the basis column
// are fp32 and the matrix-vector product runs in fp32; the two reductions of a step accumulate in fp64.
//
// CHEB = true (fp64 only): a step applies C = -T_d(c1 L - c0 I) instead of L -- d matrix-vector products that need no
// reduction at all (one LDS barrier each, the operand double-buffered) between the two reductions of a step.  The plain
// step spends 2 160 of its 4 090 cycles in those two reductions (tools/ubench_persist.hip); on stiff pose graphs the
// filtered recurrence needs about as many PRODUCTS as the plain one needs steps (Lanczos on a degree-d Chebyshev
// polynomial of L with d << sqrt(lambda_max / a) converges d times faster per step), so the reductions, the basis
// columns and the host's tridiagonal shrink d-fold.
template <int RPT, typename T = double, bool CHEB = false>
__global__ __launch_bounds__(kPersistThreads) void k_lan_persist(PersistPack P, PersistViewT<T> L, int steps, PersistCheb ch = PersistCheb()) {
    __shared__ __align__(16) unsigned char pool[kPersistPool];
    __shared__ double red1[3 * kPersistThreads / 64], red2[2 * kPersistThreads / 64];
    __shared__ double srec[3 * (kPersistMaxSteps + 1)];   // (alpha, beta, l1) of this chunk
    const int t = threadIdx.x, n = L.n;
    PCLK(true, 0);
#ifdef PERSIST_CLOCKS
    if (t == 0) L.clk[10] = wall_clock64();
#endif
    T* svec = reinterpret_cast<T*>(pool);
    // ---- band and the first two off-band entries -> registers, the further ones -> LDS (all from the packed form) ----
    T dg[RPT], lo[RPT], up[RPT], c0v[RPT], c1v[RPT];
    int c0c[RPT], c1c[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = t + k * kPersistThreads;
        dg[k] = (T)P.band[r]; lo[k] = (T)P.band[kPersistPad + r]; up[k] = (T)P.band[2 * kPersistPad + r];
        c0v[k] = (T)P.band[3 * kPersistPad + r]; c1v[k] = (T)P.band[4 * kPersistPad + r];
        c0c[k] = P.cc[r]; c1c[k] = P.cc[kPersistPad + r];
    }
    const int nc = P.crow[n];
    int* ccol = reinterpret_cast<int*>(pool + (size_t)n * 8);
    T* cval = reinterpret_cast<T*>(pool + (((size_t)n * 8 + (size_t)nc * 4 + 7) & ~(size_t)7));
    T* cprod = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(cval) + (size_t)nc * 8);   // products of the overflow entries (per step)
    T* svec2 = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(cprod) + (size_t)nc * 8);  // CHEB: second operand buffer (persist_fits_cheb)
    // the flat list's entries e = t, t + 512 of this thread stay in registers (intel: at most 189 entries in all); longer
    // lists are read from LDS in every step
    constexpr int kFlat = 2;
    const bool flat_regs = nc <= kFlat * kPersistThreads;
    int fcol[kFlat]; T fval[kFlat];
#pragma unroll
    for (int q = 0; q < kFlat; ++q) {
        const int e = t + q * kPersistThreads;
        fcol[q] = e < nc ? P.ccol[e] : 0;
        fval[q] = e < nc ? (T)P.cval[e] : (T)0;
    }
    if (!flat_regs) for (int e = t; e < nc; e += kPersistThreads) { ccol[e] = P.ccol[e]; cval[e] = (T)P.cval[e]; }
    int ob[RPT], ol[RPT];      // the row's segment of that list: offset, length (registers: no LDS look-up per step)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = t + k * kPersistThreads;
        ob[k] = P.crow[r];
        ol[k] = r < n ? P.crow[r + 1] - ob[k] : 0;
    }
    bool any_over = false;
#pragma unroll
    for (int k = 0; k < RPT; ++k) any_over = any_over || ol[k] > 0;
    const int J0 = L.st->jA;
    T u[RPT], vp[RPT], v[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = t + k * kPersistThreads;
        u[k] = r < n ? (T)L.u[r] : (T)0;
        vp[k] = r < n ? (T)L.vprev[r] : (T)0;
    }
    const double dn = (double)n, rdn = 1.0 / dn;
    __syncthreads();
    PCLK(true, 1);
    for (int s = 0; s <= steps; ++s) {
        const int j = J0 + s;
        PCLK(s == 9, 8);
        // ---- beta_j = ||u - mean||, v_j = (u - mean) / beta_j  (nx:209-213 project()) ----
        // (round 2, tools/ubench_persist.hip: a wave-wide fp64 DPP total costs ~400 shader cycles per VALUE on the step's
        // critical path, the barrier around it ~250: merging the two reductions of a step into one four-value reduction
        // made the step slower (2.34 against 1.61 us), dropping a value makes it faster -- ||v_j||_1 is only used by the
        // host for the LAST vector of a chunk, so it moved into the chunk's closing reduction)
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) { const double uk = u[k]; s1 += uk; s2 += uk * uk; }
        double l1_last = 0.0;
        if (s == steps) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) l1_last += fabs((double)vp[k]);      // ||v_{J-1}||_1
            persist_sum3(s1, s2, l1_last, red1);
        } else {
            persist_sum2(s1, s2, red1);
        }
        PCLK(s == 8, 2);
        const double mu = s1 * rdn;
        const double nrm2 = s2 - dn * mu * mu;
        const double rs = nrm2 > 1e-290 ? rsqrt(nrm2) : 0.0;
        const double beta = nrm2 * rs;
        if (s == steps) {              // chunk end: beta_J (the host's residual estimate) and ||v_{J-1}||_1
            if (t == 0) { srec[3 * s] = 0.0; srec[3 * s + 1] = beta; if (s > 0) srec[3 * (s - 1) + 2] = l1_last; }
            break;
        }
        const double inv = rs;
        double al = 0.0;
        T* vj = L.V + (size_t)j * (size_t)n;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int r = t + k * kPersistThreads;
            v[k] = (T)(((double)u[k] - mu) * inv);
            if (r < n) { svec[r] = v[k]; vj[r] = v[k]; } else v[k] = 0;
        }
        PCLK(s == 8, 3);
        lds_barrier();
        PCLK(s == 8, 4);
        // ---- w = L v_j: band and two closures per row from registers (independent LDS gathers, no
        // loops), the rare rows with more closures add theirs from the LDS CSR; alpha_j = v_j . w ----
        // (x: the operand's own entries in registers, sv: the same vector in LDS, published before the last barrier)
        auto spmv = [&](const T (&x)[RPT], const T* sv, T (&w)[RPT]) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int r = t + k * kPersistThreads;
                const int rm = max(r - 1, 0), rp = min(r + 1, n - 1);     // lo/up are 0 where the neighbour does not exist
                T a = dg[k] * x[k];
                if (r < n) a += lo[k] * sv[rm] + up[k] * sv[rp] + c0v[k] * sv[c0c[k]] + c1v[k] * sv[c1c[k]];
                w[k] = a;
            }
            if (nc > 0) {   // workgroup-uniform
                // Round 2: the entries beyond a row's two register slots used to be walked by the row's thread -- three
                // dependent LDS reads per entry (column -> operand, value), a hub row of 10 entries holding up its wave and,
                // at the next barrier, the workgroup (tools/ubench_persist.hip, 600 closures on 1 728 nodes: 2 360 of a step's
                // 5 400 cycles).  Now all threads form the products of the flat entry list (gathers in parallel, perfectly
                // balanced), and after one more LDS barrier the row's thread only adds its segment, in the same order as before
                // (round 3: segment offset and length come from registers, not from a row-pointer array in LDS).
                if (flat_regs) {
#pragma unroll
                    for (int q = 0; q < kFlat; ++q) { const int e = t + q * kPersistThreads; if (e < nc) cprod[e] = fval[q] * sv[fcol[q]]; }
                } else {
                    for (int e = t; e < nc; e += kPersistThreads) cprod[e] = cval[e] * sv[ccol[e]];
                }
                lds_barrier();
                if (any_over) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k)
                        for (int p = ob[k], e = ob[k] + ol[k]; p < e; ++p) w[k] += cprod[p];
                }
            }
        };
        spmv(v, svec, u);
        if (CHEB) {
            // u = L v  ->  t1 = M v;  t_{i+1} = 2 M t_i - t_{i-1};  C v = -t_d   (T_d is even: T_d(M) = T_d(-M))
            const T c0 = (T)ch.c0, c1 = (T)ch.c1;
            T t0[RPT], t1[RPT], w[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) { t0[k] = v[k]; t1[k] = c1 * u[k] - c0 * v[k]; }
            for (int i = 2; i <= ch.deg; ++i) {
                T* sv = (i & 1) ? svec : svec2;           // (the buffer two products back: every thread is done reading it)
#pragma unroll
                for (int k = 0; k < RPT; ++k) { const int r = t + k * kPersistThreads; if (r < n) sv[r] = t1[k]; }
                lds_barrier();
                spmv(t1, sv, w);
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const T t2 = (T)2 * (c1 * w[k] - c0 * t1[k]) - t0[k];
                    t0[k] = t1[k]; t1[k] = (t + k * kPersistThreads < n) ? t2 : (T)0;
                }
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) u[k] = -t1[k];
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) al += (double)v[k] * (double)u[k];
        PCLK(s == 8, 5);
        persist_sum1(al, red2);
        PCLK(s == 8, 6);
        // ---- u_{j+1} = w - alpha_j v_j - beta_j v_{j-1} ----
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            u[k] = (u[k] - (T)al * v[k]) - (T)beta * vp[k];
            vp[k] = v[k];
        }
        if (t == 0) { srec[3 * s] = al; srec[3 * s + 1] = beta; srec[3 * s + 2] = 0.0; }
        PCLK(s == 8, 7);
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = t + k * kPersistThreads;
        if (r < n) { L.u[r] = u[k]; L.vprev[r] = vp[k]; }
    }
    // ---- hand the chunk's records to the host (as k_pipe_tail does): triples J0..J-1 and beta_J ----
    __syncthreads();
    const int J = J0 + steps;
    const int cnt = 3 * steps + 2;
    for (int i = t; i < cnt; i += kPersistThreads) {
        if (i == 3 * steps) continue;      // the alpha_J slot belongs to the next chunk (the host poisons and awaits it)
        const double x = srec[i];
        L.tri[3 * (size_t)J0 + i] = x;
        L.htri[3 * (size_t)J0 + i] = x;
    }
    __threadfence_system();
    __syncthreads();
    PCLK(true, 9);
#ifdef PERSIST_CLOCKS
    if (t == 0) L.clk[11] = wall_clock64();
#endif
    if (t == 0) {
        L.st->jA = J; L.st->jN = J;
        const unsigned long long epoch = (unsigned long long)(unsigned int)L.st->epoch;
        __hip_atomic_store(L.hflag, (epoch << 32) | (unsigned long long)(unsigned int)J, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace machip
