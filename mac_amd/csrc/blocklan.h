// blocklan.h -- BLOCK Lanczos step (block size 4), one launch per block step (round 5).
//
// Why: on stiff pose graphs (city10000: lambda_2 = 0.05832, lambda_3 = 0.05872 at the reference's last iterate) the scalar recurrence
// needs 650-780 dependent steps per eigen-solve, each a launch of ~4 us however small the matrix is: the chip waits on a chain of
// kernel boundaries (tools/ubench_cluster.hip: a dependent step INSIDE one launch on the workgroups of a single XCD costs 4.4 us as
// well).  A block of 4 vectors sees lambda_3..lambda_5 inside the block: 220-330 block steps instead (tools/experiments/
// city_block_pipelined_emulation.py), and a block step is still one launch -- the gathers move 4 x the bytes of a launch that was
// bound by its latency chain, not by bytes.
//
// Recurrence (no re-orthogonalisation, like the scalar form):   L V_j = V_{j-1} B_j^T + V_j A_j + V_{j+1} B_{j+1},
// T_j := L V_j - V_{j-1} B_j^T,  A_j = sym(V_j^T T_j),  U = T_j - V_j A_j (minus its column means: everything stays orthogonal to 1),
// U = V_{j+1} B_{j+1} with B_{j+1} = R upper triangular from the Cholesky factorisation of U^T U.  As in kernels.h (pipe_coefs), U is
// never formed before its normalisation is known: launch j+1 receives the records (T_j, V_j) per row plus 48 partial sums per
// workgroup (T^T T, V^T T, V^T V, column sums, column 1-norms), derives A_j, the means, U^T U = T^T T - A X - X^T A + A M A - n mu mu^T,
// R and R^-1 in its prologue (wave 0 of every workgroup, identical arithmetic), and applies them on the fly:
//   V_{j+1}[r] = (T_j[r] - V_j[r] A - mu) R^-1,   (L V_{j+1})[r] = ((L T_j)[r] - (L V_j)[r] A) R^-1,   T_{j+1}[r] = (L V_{j+1})[r] - V_j[r] R^T.
// Host side (solver.h solve_block, band.h): the block tridiagonal (half-bandwidth 4), its smallest eigenpair, the residual estimate
// |B_J s_last|, the same explicit check with the reference's stop rule as every other mode (nx:232-246 via mac/utils/fiedler.py:38-75).
#pragma once
#include "kernels.h"

namespace machip {
#define BLK_CLK(cond, i) do { if (L.clk && (cond)) L.clk[blockIdx.x * 16 + (i)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)

constexpr int kBW = 4;                    // block width
constexpr int kBQ = 48;                   // partial sums per workgroup: TT (10) | X = V^T T (16) | M = V^T V (10) | sum t (4) | sum v (4) | |v|_1 (4)
constexpr int kBRec = 36;                 // host record per block step: A_j (16) | B_j (16, upper triangular) | |V_j|_1 per column (4)
constexpr int kBlkThreads = 512;          // largest workgroup of the step kernel (the other shape: 256)
constexpr int kBlkMaxGrid = 256;          // workgroups of a step (= partial sums per quantity)

struct alignas(64) BRec { double t[kBW]; double v[kBW]; };

struct BlkView {
    int n;
    LanState* st;
    BRec* Z0;
    BRec* Z1;
    double* V;                 // basis, column-major n x (4 * steps): column 4 j + c = column c of V_j
    double* rec;               // records in device memory, kBRec doubles per block step (written by workgroup 0 of every step)
    double* hrec;              // host-pinned mirror (device-mapped pointer): the tail kernel copies a chunk's records there
    unsigned long long* hflag; // (epoch << 32) | J once records < J (and B_J) have been written
    double* part;              // 2 x kMaxGrid x kBQ (workgroup-major), ping-ponged by step parity like the scalar form
    int P;                     // workgroups of the step kernel
    long long* clk;            // (probe) phase stamps
    double inv_n;              // 1 / n
};

__host__ __device__ constexpr int blk_ui(int i, int j) { return i * 4 - i * (i - 1) / 2 + (j - i); }     // upper-triangle index, i <= j
constexpr int kQtt = 0, kQx = 10, kQm = 26, kQst = 36, kQsv = 40, kQl1 = 44;
// coefficient area in LDS (doubles): A (16, full symmetric) | mu (4) | R (16, upper, row-major) | Ri (16, upper) | j
constexpr int kCA = 0, kCmu = 16, kCR = 20, kCRi = 36, kCj = 52, kCoefs = 56;

// The barriers of the step kernel order LDS traffic only: a __syncthreads() would also wait for the global stores in flight (basis
// rows, records, workgroup 0's host records) -- measured 2 us per step.
__device__ __forceinline__ void blk_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 1 / sqrt(p) for a normal positive p: the hardware estimate (v_rsq_f64) + two Newton steps (the library routine's range handling is
// dead weight here; it sits four times on the critical path of every step)
__device__ __forceinline__ double blk_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
}

// Wave 0 of a workgroup: sum the P partials of each of the 48 quantities -- the partials are stored workgroup-major (48 consecutive
// doubles per workgroup), so lane q simply loads quantity q of every workgroup (one coalesced 384-byte request per workgroup, up to
// 64 in flight) and adds them in a fixed order; no cross-lane reduction at all -- then derive the step's coefficients and publish
// them in `sc`.  Workgroup 0 also writes the host record.  adv >= 0: tail kernel (advances the chunk base).
// (First version: quantity-major partials, lane = workgroup, 48 x 64 transposition through LDS: 1.3 us of the step.)
__device__ __forceinline__ void blk_prologue(const BlkView& L, int jrel, int adv, double* sc, double* red, int* j_out) {
    const int lane = threadIdx.x;       // caller guarantees < 64
    const int jA = L.st->jA;
    const double* __restrict__ pin = L.part + (size_t)(jrel & 1) * (kBQ * kMaxGrid) + lane;      // (lanes 48..63 add up garbage nobody reads)
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < L.P; base += 64) {
        double v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = pin[(size_t)(base + i) * kBQ];       // unconditional (kMaxGrid slots: in bounds), masked below
#pragma unroll
        for (int i = 0; i < 64; ++i) s4[i & 3] += base + i < L.P ? v[i] : 0.0;
    }
    BLK_CLK(lane == 0, 1);
    red[kBQ * 65 + lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const double* tot = red + kBQ * 65;
    BLK_CLK(lane == 0, 2);
    // ---- 4 x 4 algebra: lane e = 4 i + j (mod 16) owns entry (i, j) of every product (the four 16-lane quarters of the wave compute
    // the same values and store them to the same addresses); exchange through LDS, wave-synchronous ----
    const double dn = (double)L.n;
    const int ei = (lane >> 2) & 3, ej = lane & 3, e = lane & 15;
    double* tA = sc + kCA;                          // A goes straight to its place in the coefficient area
    double* tAX = red + kBQ * 65 + 64;              // scratch: A X (16) | A M (16) | G (16)
    double* tAM = tAX + 16;
    double* tG = tAM + 16;
    tA[e] = 0.5 * (tot[kQx + ei * 4 + ej] + tot[kQx + ej * 4 + ei]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        double ax = 0.0, am = 0.0, m = tot[kQst + ej];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double aik = tA[ei * 4 + k];
            ax += aik * tot[kQx + k * 4 + ej];
            am += aik * tot[kQm + (k <= ej ? blk_ui(k, ej) : blk_ui(ej, k))];
            m -= tot[kQsv + k] * tA[k * 4 + ej];
        }
        tAX[e] = ax; tAM[e] = am;
        sc[kCmu + ej] = m * L.inv_n;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    {
        double g = tot[kQtt + (ei <= ej ? blk_ui(ei, ej) : blk_ui(ej, ei))] - tAX[ei * 4 + ej] - tAX[ej * 4 + ei] - dn * sc[kCmu + ei] * sc[kCmu + ej];
#pragma unroll
        for (int k = 0; k < 4; ++k) g += tAM[ei * 4 + k] * tA[k * 4 + ej];
        tG[e] = g;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    BLK_CLK(lane == 0, 5);
    double G[10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) G[blk_ui(i, j)] = tG[i * 4 + j];
    // Cholesky G = R^T R (R upper).  U^T U is a difference of O(|T|^2) terms: a pivot below 1e-10 of that size is rounding noise,
    // i.e. the block has (numerically) lost rank -> reported as a breakdown (R = 0), the host falls back to the scalar recurrence.
    double R[16], Ri[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { R[i] = 0.0; Ri[i] = 0.0; }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double p = G[blk_ui(k, k)];
#pragma unroll
        for (int q = 0; q < k; ++q) p -= R[q * 4 + k] * R[q * 4 + k];
        const double scale = tot[kQtt + blk_ui(k, k)];
        ok = ok && p > 1e-10 * scale && p > 1e-290;
        const double rs = ok ? blk_rsqrt(p) : 0.0;
        R[k * 4 + k] = p * rs;
        Ri[k * 4 + k] = rs;          // (diagonal of R^-1)
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
            double s = G[blk_ui(k, j)];
#pragma unroll
            for (int q = 0; q < k; ++q) s -= R[q * 4 + k] * R[q * 4 + j];
            R[k * 4 + j] = s * rs;
        }
    }
    BLK_CLK(lane == 0, 6);
    // R^-1 (upper): Ri(i, j) = -(sum_{k=i}^{j-1} Ri(i, k) R(k, j)) / R(j, j)
#pragma unroll
    for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = j - 1; i >= 0; --i) {
            double s = 0.0;
#pragma unroll
            for (int k = i; k < j; ++k) s += Ri[i * 4 + k] * R[k * 4 + j];
            Ri[i * 4 + j] = -s * Ri[j * 4 + j];
        }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { R[i] = 0.0; Ri[i] = 0.0; }
    }
    BLK_CLK(lane == 0, 3);
    const int j = jA + jrel;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = i; k < 4; ++k) { sc[kCR + i * 4 + k] = R[i * 4 + k]; sc[kCRi + i * 4 + k] = Ri[i * 4 + k]; }     // (the lower triangles are never read)
        sc[kCj] = (double)j;
        if (blockIdx.x == 0) {
            // (device memory: a step kernel that stored to host memory could not retire before PCIe had acknowledged the stores)
            double* __restrict__ h = L.rec;
            if (j > 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) h[(size_t)(j - 1) * kBRec + i] = tA[i];
#pragma unroll
                for (int c = 0; c < 4; ++c) h[(size_t)(j - 1) * kBRec + 32 + c] = tot[kQl1 + c];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) h[(size_t)j * kBRec + 16 + i] = R[i];
            if (adv >= 0) L.st->jA = j;
        }
    }
    BLK_CLK(lane == 0, 7);
    *j_out = j;
}

// One block step.  Wave 0 does nothing but the prologue (its reduction chain stays off the row work); in the other 7 waves G lanes
// share a row (G = 2 or 4 for pose graphs: 3-5 entries per row), so a workgroup owns (THREADS - 64) / G rows (512 threads: 224 rows
// at G = 2, city10000 = 45 workgroups -- one pass of 64 partial loads in the next prologue).
template <int G, int THREADS>
__global__ __launch_bounds__(THREADS) void k_blk_vec(CsrView A, BlkView L, int jrel) {
    constexpr int RPB = (THREADS - 64) / G;          // rows per workgroup
    constexpr int NW = THREADS / 64;
    constexpr int NM = RPB / 4;                      // MFMAs (4 rows each) of the workgroup's Gram matrix
    static_assert(RPB % 4 == 0, "the Gram product consumes 4 rows per MFMA");
    __shared__ double sc[kCoefs];
    __shared__ double red[kBQ * 65 + 112];           // prologue scratch (only the tail of it is used now)
    __shared__ __attribute__((aligned(16))) double Fl[RPB][16];      // per row: t (4) | v (4) | 1 | |v| (4) | 0 0 0
    __shared__ double Dl[NW][256];                   // per wave: its share of F^T F
    const int tid = threadIdx.x;
    const int wt = tid - 64;                       // worker thread index (< 0: the prologue wave)
    const int slot = wt >= 0 ? wt / G : 0, sub = wt >= 0 ? wt % G : 0;
    const int r = wt >= 0 ? (int)blockIdx.x * RPB + slot : A.n;
    const BRec* __restrict__ Zc = (jrel & 1) ? L.Z1 : L.Z0;
    BRec* __restrict__ Zn = (jrel & 1) ? L.Z0 : L.Z1;
    // row work that does not need the coefficients: raw sums (L T)[r], (L V)[r]
    double st[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0};
    BRec z;
    BLK_CLK(tid == 0, 0); BLK_CLK(tid == 64, 8);
    if (tid < 64) {
        int jd;
        blk_prologue(L, jrel, -1, sc, red, &jd);
    } else if (r < A.n) {
        const int lo = A.rowptr[r], hi = A.rowptr[r + 1];
        if (sub * (G >= 4 ? 1 : 4 / G) < 4) z = Zc[r];
        for (int e = lo + sub; e < hi; e += G) {
            const double a = A.val[e];
            const BRec zc = Zc[A.col[e]];
#pragma unroll
            for (int c = 0; c < 4; ++c) { st[c] = __builtin_fma(a, zc.t[c], st[c]); sv[c] = __builtin_fma(a, zc.v[c], sv[c]); }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { st[c] = group_sum<G>(st[c]); sv[c] = group_sum<G>(sv[c]); }
    BLK_CLK(tid == 64, 9); BLK_CLK(tid == 0, 4);
    blk_lds_barrier();
    BLK_CLK(tid == 64, 10);
    const int j = (int)sc[kCj];
    // finish: the lanes of a row's group share the four columns (CPL each); every one of them forms the row of U and of
    // L T - (L V) A (32 multiply-adds) and then its own columns of V_{j+1}, L V_{j+1}, T_{j+1}
    constexpr int CPL = G >= 4 ? 1 : 4 / G;
    if (wt >= 0 && sub * CPL < 4) {
#pragma clang fp contract(off)
        const int c0 = sub * CPL;
        double* f = Fl[slot];
        if (r < A.n) {
            double u[4], wl[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                double su = z.t[d], sw = st[d];
#pragma unroll
                for (int k = 0; k < 4; ++k) { su = __builtin_fma(-z.v[k], sc[kCA + k * 4 + d], su); sw = __builtin_fma(-sv[k], sc[kCA + k * 4 + d], sw); }
                u[d] = su - sc[kCmu + d]; wl[d] = sw;
            }
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) {
                const int c = c0 + cc;
                double vn = 0.0, w = 0.0;
#pragma unroll
                for (int d = 0; d < 4; ++d) if (d <= c) { vn = __builtin_fma(u[d], sc[kCRi + d * 4 + c], vn); w = __builtin_fma(wl[d], sc[kCRi + d * 4 + c], w); }
                double tn = w;
#pragma unroll
                for (int d = 0; d < 4; ++d) if (d >= c) tn = __builtin_fma(-z.v[d], sc[kCR + c * 4 + d], tn);
                L.V[(size_t)(4 * j + c) * A.n + r] = vn;
                Zn[r].t[c] = tn; Zn[r].v[c] = vn;
                f[c] = tn; f[4 + c] = vn; f[9 + c] = fabs(vn);
            }
        } else {
#pragma unroll
            for (int cc = 0; cc < CPL; ++cc) { f[c0 + cc] = 0.0; f[4 + c0 + cc] = 0.0; f[9 + c0 + cc] = 0.0; }
        }
        if (sub == 0) { f[8] = r < A.n ? 1.0 : 0.0; f[13] = 0.0; f[14] = 0.0; f[15] = 0.0; }
    }
    BLK_CLK(tid == 64, 11);
    blk_lds_barrier();
    // ---- the 48 sums of the step = entries of F^T F (F: RPB rows x 16 features) on the f64 matrix cores: lane l supplies
    // F[4 m + (l >> 4)][l & 15] as BOTH operands of v_mfma_f64_16x16x4_f64 (A[i][k] = B[k][i] here); wave w takes every NW-th
    // group of 4 rows; D[row = (l >> 4) + 4 reg][col = l & 15].  (First version: every finishing lane parked its 48 products in
    // LDS, 48 x 8 lanes added them up: 1-2 us per step in LDS traffic.) ----
    {
        typedef double blk_d4 __attribute__((ext_vector_type(4)));
        const int wv = tid >> 6, l = tid & 63;
        blk_d4 acc = {0.0, 0.0, 0.0, 0.0};
        for (int m = wv; m < NM; m += NW) {
            const double f = Fl[4 * m + (l >> 4)][l & 15];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(f, f, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) Dl[wv][((l >> 4) + 4 * q) * 16 + (l & 15)] = acc[q];
    }
    blk_lds_barrier();
    if (tid < kBQ) {
        // quantity -> entry of the Gram matrix: TT(i, k) = D[i][k], X(i, k) = v_i t_k = D[4 + i][k], M(i, k) = D[4 + i][4 + k],
        // sum t_c = D[8][c], sum v_c = D[8][4 + c], |v_c|_1 = D[8][9 + c]
        int row, col;
        if (tid < kQx) {            // upper triangle of TT
            int i = 0, rem = tid;
            while (rem >= 4 - i) { rem -= 4 - i; ++i; }
            row = i; col = i + rem;
        } else if (tid < kQm) { row = 4 + (tid - kQx) / 4; col = (tid - kQx) % 4; }
        else if (tid < kQst) {
            int i = 0, rem = tid - kQm;
            while (rem >= 4 - i) { rem -= 4 - i; ++i; }
            row = 4 + i; col = 4 + i + rem;
        } else if (tid < kQsv) { row = 8; col = tid - kQst; }
        else if (tid < kQl1) { row = 8; col = 4 + tid - kQsv; }
        else { row = 8; col = 9 + tid - kQl1; }
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += Dl[w][row * 16 + col];
        L.part[(size_t)((jrel + 1) & 1) * (kBQ * kMaxGrid) + (size_t)blockIdx.x * kBQ + tid] = s;
    }
    BLK_CLK(tid == 64, 12);
}

// Start of a sequence: Z0 = (T = U0, V = 0); partials such that step 0 orthonormalises U0 (A = 0, G = U0^T U0 - n mu mu^T).
// U0: column-major n x 4.  Launched with the step kernel's grid (every partial slot is written).
__global__ __launch_bounds__(kBlock) void k_blk_init(BlkView L, const double* __restrict__ U0, int epoch) {
    __shared__ double sm[4];
    double tt[10], st[4];
#pragma unroll
    for (int i = 0; i < 10; ++i) tt[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) st[i] = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < L.n; r += gridDim.x * kBlock) {
        BRec o;
#pragma unroll
        for (int c = 0; c < 4; ++c) { o.t[c] = U0[(size_t)c * L.n + r]; o.v[c] = 0.0; st[c] += o.t[c]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = i; k < 4; ++k) tt[blk_ui(i, k)] += o.t[i] * o.t[k];
        L.Z0[r] = o;
    }
    for (int q = 0; q < kBQ; ++q) {
        double v = 0.0;
        if (q < 10) v = tt[q];
        else if (q >= kQst && q < kQst + 4) v = st[q - kQst];
        if (q < 10 || (q >= kQst && q < kQst + 4)) v = block_sum(v, sm);
        if (threadIdx.x == 0) L.part[(size_t)blockIdx.x * kBQ + q] = v;
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { L.st->jA = 0; L.st->jN = 0; L.st->epoch = epoch; }
}

// End of a chunk of `adv` block steps: the record of step J = jA + adv (A_{J-1}, |V_{J-1}|_1, B_J), the new chunk base, the flag.
__global__ __launch_bounds__(64) void k_blk_tail(BlkView L, int adv) {
    __shared__ double sc[kCoefs];
    __shared__ double red[kBQ * 65 + 112];
    int j = 0;
    blk_prologue(L, adv, adv, sc, red, &j);
    // records [j - adv - 1, j] -> host (what this chunk delivers for the first time: A / l1 of steps j-adv-1 .. j-1, B of j-adv .. j; the
    // slots this very wave has just written are re-read through the same wave's in-order memory stream)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int lo = max(0, j - adv - 1);
    const int cnt = (j - lo + 1) * kBRec;
    for (int i = threadIdx.x; i < cnt; i += 64) {
        const int rj = lo + i / kBRec, f = i % kBRec;
        const bool is_b = f >= 16 && f < 32;
        if (is_b ? (rj > lo || lo == 0) : rj < j) L.hrec[(size_t)lo * kBRec + i] = L.rec[(size_t)lo * kBRec + i];
    }
    if (threadIdx.x == 0) {
        const unsigned long long epoch = (unsigned long long)(unsigned int)L.st->epoch;
        __hip_atomic_store(L.hflag, (epoch << 32) | (unsigned long long)(unsigned int)j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Start block: column c of U0 <- src (n values)
__global__ __launch_bounds__(kBlock) void k_blk_setcol(double* __restrict__ U0, int n, int c, const double* __restrict__ src) {
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) U0[(size_t)c * n + r] = src[r];
}

}  // namespace machip
