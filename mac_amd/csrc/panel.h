// panel.h -- column-panel (LDS-blocked) Lanczos step for matrices whose gather operand does not fit a cache a
// workgroup can reach cheaply (BASELINE.json configs[3]: n = 1e5, 16-byte records = 1.6 MB).
//
// Why (profiles/r2_c4_counters.md): the one-kernel step k_pipe_vec gathers a 16-byte record per matrix entry from a
// 1.6 MB operand; every gather is one L1->L2 request for a whole 128-byte line (2.7 M requests per step) and the CU's
// outstanding-miss capacity times the L2 latency bounds the step at ~22 us -- 0.22 of the HBM roofline.  Here the gather
// target is LDS instead:
//   * L(x) is kept a second time in PANEL FORM: NB row blocks x NP column panels; workgroup (b, p) owns the entries of
//     row block b whose column lies in panel p (~nnz / 256 of them), stored per 64-row tile in a jagged, lane-major
//     order (entry i of lane l at  tile base + #{(l', i') : i' < i or (i' = i and l' < l), len(l') > i'}), so that a
//     wave streams them with perfectly coalesced loads and every lane accumulates ITS OWN row: no shuffles, no
//     staging, no atomics, a fixed summation order.
//   * k_pan_mul (workgroup (b, p), 1024 threads): bulk-loads the panel's records {t_{j-1}, v_{j-1}} with coalesced
//     16-byte loads while wave 0 finishes step j-1's reductions, turns them into v_j with the coefficients (8 bytes per
//     column in LDS: a panel of up to 16 384 columns), then streams its tiles: y_p[r] = sum_{c in panel p} L[r,c] v_j[c]
//     with the gathers served by LDS.  One 8-byte partial per (row, panel) goes to HBM.
//   * k_pan_fin (row-parallel): w = sum_p y_p[r] in panel order, Paige's t_j = w - beta_j v_{j-1}, the next record
//     {t_j, v_j}, the basis column, and the six measured inner products of kernels.h (PipeRow): everything the
//     one-kernel step does after its SpMV.  The recurrence, the records, the tridiagonal hand-off and the tail kernel
//     are those of kernels.h; only the product L v_j is computed differently.
// Two launches per step instead of one; per step 21 x 1.6 MB of coalesced panel loads + 2 x 9.6 MB of partials replace
// 2.7 M scattered line requests.  Reference behaviour preserved: nx:209-213 (mean projection), stop rule nx:232/246
// (evaluated by solver.h exactly as for the other step kernels).
#pragma once
#include "kernels.h"

namespace machip {

struct PanView {
    int n;
    int NP;      // column panels
    int C;       // columns per panel (the last one may be shorter)
    int NB;      // row blocks
    int TPB;     // 64-row tiles per row block
    int* tptr;              // [NB*NP*TPB + 1] first entry of tile (b, p, t); tiles ordered ((b*NP + p)*TPB + t)
    unsigned short* tlen;   // [NB*NP*TPB*64]  entries of row (b*TPB + t)*64 + lane inside panel p
    double* bval;           // panel-form values
    unsigned short* bcol;   // column minus the panel's first column
    double* ypart;          // [NP][n] per-panel partial products
    double* coef;           // 8 doubles: (alpha, beta, mu, inv, j) of the running step, published by k_pan_mul for k_pan_fin
    int* tcount;            // [NB*NP*TPB] entries per tile (assembly scratch)
#ifdef PAN_CLOCKS
    long long* clk;         // tools/ubench6.hip: 16 wall-clock stamps (100 MHz) per workgroup
#endif
};

#ifdef PAN_CLOCKS
#define PAN_CLK(cond, i) do { if (cond) A.clk[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define PAN_CLK(cond, i) do { } while (0)
#endif

// v_j[c] from the record {t_{j-1}, v_{j-1}}: evaluated by k_pan_mul (gather operand) and k_pan_fin (stored vector)
// with the SAME three roundings, so the inner products k_pan_fin measures are those of the vector that was multiplied.
__device__ __forceinline__ double pan_vj(double alpha, double mu, double inv, double zt, double zv) {
#pragma clang fp contract(off)
    const double a = __builtin_fma(-alpha, zv, zt);
    const double b = a - mu;
    return b * inv;
}

// ------------------------------------------------------------------------------------------
// Panel form of an assembled CSR (diagonal first, other columns ascending): three launches, integers only.
// ------------------------------------------------------------------------------------------
// Pass 1: one wave per 64-row group, lane = row: entries per (row, panel) and per tile.
__global__ __launch_bounds__(kBlock) void k_pan_count(CsrView A, PanView P) {
    const int lane = threadIdx.x & 63;
    const int gt = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);   // 64-row group
    if (gt >= P.NB * P.TPB) return;
    const int b = gt / P.TPB, t = gt - b * P.TPB;
    const int r = gt * 64 + lane;
    const bool valid = r < A.n;
    int e = 0, end = 0, pd = -1;
    if (valid) { e = A.rowptr[r] + 1; end = A.rowptr[r + 1]; pd = r / P.C; }   // (the diagonal sits first and is counted with its panel)
    for (int p = 0; p < P.NP; ++p) {
        const int hi = (p + 1) * P.C;
        int c = (p == pd) ? 1 : 0;
        while (e < end && A.col[e] < hi) { ++e; ++c; }
        const int vt = (b * P.NP + p) * P.TPB + t;
        P.tlen[(size_t)vt * 64 + lane] = (unsigned short)c;
        const int tot = wave_sum_i(c);
        if (lane == 0) P.tcount[vt] = tot;
    }
}

// Pass 2: exclusive scan of the tile counts (one workgroup; a few ten thousand values).
__global__ __launch_bounds__(1024) void k_pan_scan(PanView P) {
    __shared__ int s_part[1024];
    const int NT = P.NB * P.NP * P.TPB;
    const int per = (NT + 1023) / 1024;
    const int tid = threadIdx.x;
    const int lo = tid * per, hi = min(NT, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += P.tcount[i];
    s_part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += add;
        __syncthreads();
    }
    int run = s_part[tid] - s;
    for (int i = lo; i < hi; ++i) { P.tptr[i] = run; run += P.tcount[i]; }
    if (tid == 1023) P.tptr[NT] = s_part[1023];
}

// Pass 3: same walk; entry i of a lane goes behind the entries i' < i of its tile and the lower lanes' entries i.
__global__ __launch_bounds__(kBlock) void k_pan_fill(CsrView A, PanView P) {
    const int lane = threadIdx.x & 63;
    const int gt = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (gt >= P.NB * P.TPB) return;
    const int b = gt / P.TPB, t = gt - b * P.TPB;
    const int r = gt * 64 + lane;
    const bool valid = r < A.n;
    int cur = 0, pd = -1, diag = 0;
    if (valid) { diag = A.rowptr[r]; cur = diag + 1; pd = r / P.C; }
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int p = 0; p < P.NP; ++p) {
        const int vt = (b * P.NP + p) * P.TPB + t;
        const int len = P.tlen[(size_t)vt * 64 + lane];
        const int c0 = p * P.C;
        int off = P.tptr[vt];
        const int shift = (p == pd) ? 1 : 0;     // this lane's entry 0 is the diagonal
        for (int i = 0;; ++i) {
            const bool act = i < len;
            const unsigned long long m = __ballot(act);
            if (!m) break;
            if (act) {
                const int src = (shift && i == 0) ? diag : cur + i - shift;
                const int dst = off + __popcll(m & below);
                P.bval[dst] = A.val[src];
                P.bcol[dst] = (unsigned short)(A.col[src] - c0);
            }
            off += __popcll(m);
        }
        cur += len - shift;
    }
}

// ------------------------------------------------------------------------------------------
// Step kernel 1: y_p = L[block b, panel p] v_j
// ------------------------------------------------------------------------------------------
constexpr int kPanThreads = 1024;
constexpr int kPanWaves = kPanThreads / 64;
constexpr int kPanTG = 6;     // tiles a wave walks together

template <int RPT>   // records per thread: the panel holds at most RPT * 1024 columns
__global__ __launch_bounds__(kPanThreads) void k_pan_mul(PanView A, PipeView L, int jrel) {
    __shared__ double sv[RPT * kPanThreads];
    __shared__ double scoef[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.x / A.NP, p = blockIdx.x - b * A.NP;
    const int c0 = p * A.C;
    const int Cp = min(A.C, A.n - c0);           // >= 1 by construction of the plan
    const Z2* __restrict__ Zc = ((jrel & 1) ? L.Z1 : L.Z0) + c0;
    PAN_CLK(tid == 0, 0); PAN_CLK(tid == 64, 1);
    // the panel's records, all requested at once (clamped index: unconditional loads stay batched, cf. the prologue)
    Z2 z[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) z[i] = Zc[min(tid + kPanThreads * i, Cp - 1)];
    // this wave's tiles: t = wv + 16 q.  The entry stream of a tile is a chain of dependent round trips (lengths ->
    // entries -> LDS), so kPanTG tiles are walked TOGETHER: their loads are issued back to back (first build: one tile
    // after the other, ~12 serial round trips per wave, 24 us per step whatever the matrix held).
    const int vt0 = (b * A.NP + p) * A.TPB;
    int len[kPanTG], off[kPanTG];
#pragma unroll
    for (int q = 0; q < kPanTG; ++q) {
        const int t = wv + kPanWaves * q;
        const bool ok = t < A.TPB;
        const int vt = vt0 + (ok ? t : 0);
        const int l = A.tlen[(size_t)vt * 64 + lane], o = A.tptr[vt];
        len[q] = ok ? l : 0; off[q] = o;
    }
#ifdef PAN_CLOCKS
    if (tid >= 64 && tid < 128) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PAN_CLK(tid == 64, 2); }   // records + tile heads arrived (wave 1)
#endif
    if (tid < 64) {
        int jd;
        const PipeCoef c = pipe_prologue_wave0(L, jrel, -1, scoef, &jd);
        if (blockIdx.x == 0 && lane == 0) {
            A.coef[0] = c.alpha; A.coef[1] = c.beta; A.coef[2] = c.mu; A.coef[3] = c.inv; A.coef[4] = (double)jd;
        }
        PAN_CLK(tid == 0, 3);
    }
    __syncthreads();
    PAN_CLK(tid == 64, 4);
    {
        const double alpha = scoef[0], mu = scoef[2], inv = scoef[3];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int c = tid + kPanThreads * i;
            if (c < Cp) sv[c] = pan_vj(alpha, mu, inv, z[i].t, z[i].v);
        }
    }
    __syncthreads();
    PAN_CLK(tid == 64, 5);
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int tq = 0; wv + kPanWaves * tq < A.TPB; tq += kPanTG) {
        if (tq) {     // more than kPanTG tiles per wave: next group
#pragma unroll
            for (int q = 0; q < kPanTG; ++q) {
                const int t = wv + kPanWaves * (tq + q);
                const bool ok = t < A.TPB;
                const int vt = vt0 + (ok ? t : 0);
                const int l = A.tlen[(size_t)vt * 64 + lane], o = A.tptr[vt];
                len[q] = ok ? l : 0; off[q] = o;
            }
        }
        double acc[kPanTG];
#pragma unroll
        for (int q = 0; q < kPanTG; ++q) acc[q] = 0.0;
        for (int i0 = 0;; i0 += 2) {
            unsigned long long m0[kPanTG], m1[kPanTG];
            unsigned long long any = 0;
#pragma unroll
            for (int q = 0; q < kPanTG; ++q) { m0[q] = __ballot(len[q] > i0); m1[q] = __ballot(len[q] > i0 + 1); any |= m0[q]; }
            if (!any) break;
            double v0[kPanTG], v1[kPanTG];
            int k0[kPanTG], k1[kPanTG];
#pragma unroll
            for (int q = 0; q < kPanTG; ++q) {     // inactive lanes read the tile's entry `off` (in bounds) and are not added
                const int o1 = off[q] + __popcll(m0[q]);
                const int e0 = (len[q] > i0) ? off[q] + __popcll(m0[q] & below) : off[q];
                const int e1 = (len[q] > i0 + 1) ? o1 + __popcll(m1[q] & below) : off[q];
                v0[q] = A.bval[e0]; v1[q] = A.bval[e1];
                k0[q] = A.bcol[e0]; k1[q] = A.bcol[e1];
                off[q] = o1 + __popcll(m1[q]);
            }
#pragma unroll
            for (int q = 0; q < kPanTG; ++q) {
                const bool a0 = len[q] > i0, a1 = len[q] > i0 + 1;
                const double x0 = sv[a0 ? k0[q] : 0], x1 = sv[a1 ? k1[q] : 0];
                if (a0) acc[q] += v0[q] * x0;
                if (a1) acc[q] += v1[q] * x1;
            }
            PAN_CLK(tid == 64 && i0 == 0 && tq == 0, 6);
        }
        PAN_CLK(tid == 64 && tq == 0, 7);
#pragma unroll
        for (int q = 0; q < kPanTG; ++q) {
            const int t = wv + kPanWaves * (tq + q);
            const int row = (b * A.TPB + t) * 64 + lane;
            if (t < A.TPB && row < A.n) A.ypart[(size_t)p * A.n + row] = acc[q];
        }
    }
    PAN_CLK(tid == 64, 8); PAN_CLK(tid == 1023, 9);
}

// ------------------------------------------------------------------------------------------
// Step kernel 2: w = sum_p y_p, record / basis column / inner products of the step (row-parallel)
// ------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pan_fin(PanView A, PipeView L, int jrel) {
    __shared__ double smw[kNP * BLOCK];
    const double alpha = A.coef[0], beta = A.coef[1], mu = A.coef[2], inv = A.coef[3];
    const int j = (int)A.coef[4];
    const Z2* __restrict__ Zc = (jrel & 1) ? L.Z1 : L.Z0;
    Z2* __restrict__ Zn = (jrel & 1) ? L.Z0 : L.Z1;
    double* __restrict__ vj = L.V + (size_t)j * (size_t)L.n;
    PipeRow pr;
    pr.clear();
    const int n = A.n, NP = A.NP;
    for (int r = blockIdx.x * BLOCK + threadIdx.x; r < n; r += gridDim.x * BLOCK) {
        const Z2 z = Zc[r];
        double w = 0.0;
        for (int p0 = 0; p0 < NP; p0 += 16) {     // sixteen panels in flight; added in panel order
            double y[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) y[q] = A.ypart[(size_t)min(p0 + q, NP - 1) * n + r];
#pragma unroll
            for (int q = 0; q < 16; ++q) w += (p0 + q < NP) ? y[q] : 0.0;
        }
        Z2 o;
        o.v = pan_vj(alpha, mu, inv, z.t, z.v);
        o.t = w - beta * z.v;                       // Paige's intermediate for the next step
        vj[r] = o.v;
        Zn[r] = o;
        const double t = o.t, v = o.v;
        pr.acc[0] += t * t; pr.acc[1] += t * v; pr.acc[2] += v * v;
        pr.acc[3] += t; pr.acc[4] += v; pr.acc[5] += fabs(v);
    }
    pr.template store<BLOCK>(L, jrel, smw);
}

}  // namespace machip
