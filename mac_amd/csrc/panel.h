// panel.h -- column-panel (LDS-blocked) Lanczos step for matrices whose gather operand does not fit a cache a
// workgroup can reach cheaply (BASELINE.json configs[3]: n = 1e5, 16-byte records = 1.6 MB).
//
// Why (profiles/r2_c4_counters.md): the one-kernel step k_pipe_vec gathers a 16-byte record per matrix entry from a
// 1.6 MB operand; every gather is one L1->L2 request for a whole 128-byte line (2.7 M requests per step) and the CU's
// outstanding-miss capacity times the L2 latency bounds the step at ~22 us -- 0.22 of the HBM roofline.  Here the gather
// target is LDS instead:
//   * L(x) is kept a second time in PANEL FORM: NB row blocks x NP column panels; workgroup (b, p) owns the entries of
//     row block b whose column lies in panel p (~nnz / 256 of them).  Inside (b, p) the block's rows are SORTED by their
//     number of entries in the panel (longest first) and cut into tiles of 64 consecutive sorted rows; a tile is stored
//     as a dense  (longest row of the tile) x 64  array, entry i of lane l at  tile base + 64 i + l,  zero-padded
//     (sliced ELLPACK with a block-wide sorting scope: rows of one tile differ by at most one entry almost everywhere,
//     a few per cent of padding).  Sorted tile ts goes to worker wave ts mod 15, and a wave's tiles are adjacent in
//     memory: a wave streams ONE contiguous range in 64-entry chunks with static addresses, every lane accumulates its
//     own row (no shuffles, no staging of products, no atomics, a fixed summation order), every lane of every
//     instruction does useful work, and all of a wave's loads are in flight at once.
//   * k_pan_mul (workgroup (b, p), 1024 threads): wave 0 finishes step j-1's reductions; waves 1..15 bulk-load the
//     panel's records {t_{j-1}, v_{j-1}} with coalesced 16-byte loads and request their chunk range, then turn the
//     records into v_j with the coefficients (8 bytes per column in LDS) and accumulate
//     y_p[r] = sum_{c in panel p} L[r,c] v_j[c] with the gathers served by LDS; the sums are un-sorted through an LDS
//     image of the row block and written as one 8-byte partial per (row, panel), coalesced.
//   * k_pan_fin (row-parallel): w = sum_p y_p[r] in panel order, Paige's t_j = w - beta_j v_{j-1}, the next record
//     {t_j, v_j}, the basis column, and the six measured inner products of kernels.h (PipeRow): everything the
//     one-kernel step does after its SpMV.  The recurrence, the records, the tridiagonal hand-off and the tail kernel
//     are those of kernels.h; only the product L v_j is computed differently.
// Two launches per step instead of one; per step NB x 1.6 MB of coalesced panel loads + 2 x NP x 0.8 MB of partials
// replace 2.7 M scattered line requests.  Reference behaviour preserved: nx:209-213 (mean projection), stop rule
// nx:232/246 (evaluated by solver.h exactly as for the other step kernels).
//
// How the walk got here (tools/ubench6.hip, phase clocks on MI355X, n = 1e5): (1) rows in natural order, jagged tiles,
// a ballot + rank per entry: 30 % of the lanes busy, VALU-issue bound, 9-14 us; (2) rows sorted per 320-row window,
// loads per (tile, iteration): a chain of ~10 HBM round trips per wave -- once the number of outstanding loads depends
// on wave-uniform branches the compiler can only wait for all of them -- 5-9 us; (3) dense chunk loads up front,
// products parked in LDS, jagged sums out of LDS: the rank/mask bookkeeping alone kept the VALU busy for 3-6 us.
// Padded tiles with a block-wide sort need none of it: one LDS gather and one FMA per 64 entries.
#pragma once
#include "kernels.h"

namespace machip {

constexpr int kPanThreads = 1024;
constexpr int kPanWaves = kPanThreads / 64;
constexpr int kPanWork = kPanWaves - 1;            // worker waves: wave 0 of k_pan_mul only runs the reduction prologue
constexpr int kPanWorkThreads = 64 * kPanWork;
constexpr int kPanTW = 8;          // most tiles per worker wave
constexpr int kPanRows = 64 * kPanWork * kPanTW;   // most rows per row block (7 680)
constexpr int kPanCH = 20;         // 64-entry chunks a wave holds in registers per round
constexpr int kPanMaxLen = 127;    // longest row the build kernels' histograms describe (plan_panel checks)
constexpr int kPanSlack = 64 * kPanCH + 64;        // entries the value / column arrays carry past the last tile

struct PanView {
    int n;
    int NP;      // column panels
    int C;       // columns per panel (the last one may be shorter)
    int NB;      // row blocks
    int NTB;     // 64-row tiles per row block (rows per block R = 64 NTB <= kPanRows)
    int TWW;     // tiles per worker wave = ceil(NTB / 15); a (block, panel) has 15 TWW physical tiles
    int CELLS;   // row blocks a workgroup of k_pan_mul_multi walks (1: the single-cell kernel k_pan_mul)
    int* tptr;              // [cells (15 TWW + 1)] first entry of physical tile (w TWW + q) of cell b NP + p (sorted tile w + 15 q); a cell's last entry = end of its data
    int* cbase;             // [cells + 1] first entry of every cell's static range of bval / bcol (k_pan_cellcap)
    unsigned short* thead;  // [tiles*64] row (relative to the block) held by each slot
    double* bval;           // panel-form values, zero-padded tiles
    unsigned short* bcol;   // column minus the panel's first column
    double* ypart;          // [NP][n] per-panel partial products
    double* coef;           // 8 doubles: (alpha, beta, mu, inv, j) of the running step, published by k_pan_mul for k_pan_fin
    unsigned int* tick;     // [NB] one-launch step (k_pan_step): arrivals of a row block's NP workgroups, monotonic over a sequence
    unsigned int* claim;    // [NB * NP] ... and which step's share (row block b, rows of slice p) has been taken
    int spin_ticks;         // ... how long (100 MHz ticks) a workgroup waits for its row block before leaving its share to the last arriver
    int* ovf;               // set by k_pan_rows when a row has more than kPanMaxLen entries inside ONE panel (the build would clamp it)
    int rev;                // 1: k_pan_mul8 walks its chunks backwards on odd steps (option panel_rev; panel_u.h)
    int band;               // 1: the tridiagonal band (diagonal, columns r - 1 and r + 1) is kept OUT of the panel form -- bd holds it, k_pan_fin adds it
    double* bd;             // [3][n] band values: diagonal, column r - 1, column r + 1 (0 where absent)
    int* bpk;               // [n] (CSR index of the row's first off-diagonal band entry) << 3 | how many there are (0..7): the hole the build skips
    int* ps;                // [n][NP+1] first off-diagonal entry of row r at or behind panel p (assembly scratch)
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
// Panel form of an assembled CSR (diagonal first, other columns ascending): four launches, integers only.
// ------------------------------------------------------------------------------------------
// Pass 0: where every row's off-diagonal entries of panel p start, ps[p n + r] (p = 0 .. NP; one walk per row).
// (First build of the panel form walked the rows inside the count / fill kernels, one workgroup per row block: 21
// workgroups for the whole matrix, 0.9-1.3 ms per Frank-Wolfe iteration.  With this table count and fill run one
// workgroup per (row block, panel).)
__global__ __launch_bounds__(kBlock) void k_pan_rows(CsrView A, PanView P) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= A.n) return;
    int e = A.rowptr[r] + 1;               // (the diagonal sits first and is counted with its own panel)
    const int end = A.rowptr[r + 1];
    bool over = false;
    int hb = end, hn = 0;                  // band mode: the (at most two, adjacent) entries of columns r - 1, r + 1
    double vl = 0.0, vu = 0.0;
    for (int p = 0; p < P.NP; ++p) {
        P.ps[(size_t)r * (P.NP + 1) + p] = e;
        const int hi = (p + 1) * P.C, e0 = e;
        int inband = 0;
        while (e < end && A.col[e] < hi) {
            const int c = A.col[e];
            if (P.band && (c == r - 1 || c == r + 1)) {
                if (!hn) hb = e;
                ++hn; ++inband;
                if (c == r - 1) vl += A.val[e]; else vu += A.val[e];      // (+=: a candidate may duplicate a chain pair)
            }
            ++e;
        }
        over |= e - e0 - inband + ((!P.band && r / P.C == p) ? 1 : 0) > kPanMaxLen;
    }
    P.ps[(size_t)r * (P.NP + 1) + P.NP] = end;
    if (P.band) {
        P.bpk[r] = (hb << 3) | min(hn, 7);
        over |= hn > 7;
        P.bd[r] = A.val[A.rowptr[r]]; P.bd[(size_t)A.n + r] = vl; P.bd[2 * (size_t)A.n + r] = vu;
    }
    if (over) *P.ovf = 1;                  // (a hub row concentrated in one panel: the solver falls back to the gather step)
}

constexpr int kPanRT = (kPanRows + kPanThreads - 1) / kPanThreads;   // rows per thread (8)

// entries of row r inside panel p (diagonal included when r lies in the panel); *start = first off-diagonal one
// Band mode: neither the diagonal nor the entries of columns r - 1 / r + 1 count; *hole / *hlen = the CSR range of those inside
// [start, end of the panel's run) that the copy has to skip (hlen = 0: none).
__device__ __forceinline__ int pan_row_count(const PanView& P, int n, int r, int p, int* start, int* hole = nullptr, int* hlen = nullptr) {
    const int st = P.ps[(size_t)r * (P.NP + 1) + p], en = P.ps[(size_t)r * (P.NP + 1) + p + 1];
    *start = st;
    if (P.band) {
        const int pk = P.bpk[r], hb = pk >> 3, hn = pk & 7;
        const int h0 = max(st, hb), h1 = min(en, hb + hn), hl = max(0, h1 - h0);
        if (hole) { *hole = h0; *hlen = hl; }
        return min(en - st - hl, kPanMaxLen);
    }
    if (hole) { *hole = 0; *hlen = 0; }
    return min(en - st + (r / P.C == p ? 1 : 0), kPanMaxLen);
}

// Build kernel: workgroup id -> (row block b, panel p).  Workgroups are dealt round-robin to the 8 XCDs, and the NP cells of one
// row block read the same CSR rows -- a row's run inside a panel is ~2.5 entries, so every 128-byte line is wanted by several
// panels: with b = id / NP each line was fetched into eight different L2s (338 MB fetched per launch for a 36 MB matrix, round 3).
// XCD x therefore takes the cells [x per, (x + 1) per) in row-block-major order, per = ceil(cells / 8): whole row blocks almost
// everywhere, and no XCD more than its share (round 5: the build keeps a 150 KB LDS image -- one workgroup per CU -- and the
// round-3 mapping, row block b on XCD b mod 8, gave five XCDs 36 cells for 32 CUs: two waves of workgroups, twice the time).
__host__ __device__ inline int pan_build_grid(int NB, int NP) { return ((NB * NP + 7) / 8) * 8; }
__device__ __forceinline__ bool pan_build_bp(int id, int NB, int NP, int* b, int* p) {
    const int per = (NB * NP + 7) >> 3;
    const int cell = (id & 7) * per + (id >> 3);
    *b = cell / NP;
    *p = cell - *b * NP;
    return cell < NB * NP && (id >> 3) < per;
}

// Static capacity of the cells (round 5): the entries a (row block, panel) cell can ever hold are the slots of the UNION PATTERN
// inside it, so every cell owns a fixed range of the value / column arrays -- [cbase[cell], cbase[cell + 1]) -- and the build
// needs no scan over the tiles of the whole matrix (k_pan_scan, one workgroup, 20 us per Frank-Wolfe iteration, is gone; count
// and fill are one launch).  Counted once per handle and panel shape; cellcnt[cell] = pattern slots of the cell.
__global__ __launch_bounds__(kBlock) void k_pan_cellcap(PatternView Pt, int R, int C, int NP, int* __restrict__ cellcnt) {
    const int r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= Pt.n) return;
    const int b = r / R;
    int run = 0, cur = -1;
    for (int e = Pt.prow[r]; e < Pt.prow[r + 1]; ++e) {
        const int p = Pt.pcol[e] / C;
        if (p != cur) { if (run) atomicAdd(&cellcnt[b * NP + cur], run); cur = p; run = 0; }
        ++run;
    }
    if (run) atomicAdd(&cellcnt[b * NP + cur], run);
}

// The panel form of one cell in ONE launch (round 5; was k_pan_count + k_pan_scan + k_pan_fill): sort the block's rows by
// their number of entries in the panel (counting sort in LDS; ties in arrival order -- a row's sum does not depend on the slot
// it lands in, PROVIDED the multiply-accumulate of k_pan_mul rounds the same way at every chunk position: see the contraction
// note there), cut them into 64-row tiles, and copy every row's entries into the slot it was dealt, zero-padded to the tile's
// height.  The copy goes THROUGH LDS: a thread reads ITS rows' short runs (batched: the first four entries of all eight rows
// are in flight together) and drops them at their final offsets of an LDS image of the cell (zeroed first: the padding), and the
// image is then streamed out with perfectly coalesced stores -- written straight from the row-owning threads, a tile row's 512
// bytes came from 64 different waves, one 8-byte request each (117 us per launch at configs[3]; the first build of this kernel).
// An image holds kPanStage entries; a larger cell takes several rounds (ranges of the cell's entries, whatever tile they belong to).
// Tile table: tptr[cell (NTP + 1) + physical tile], the cell's last entry = end of its data (cells are not adjacent in memory).
constexpr int kPanStage = 15360;       // entries of the LDS image (120 KB of values + 30 KB of columns: one workgroup per CU)
// RT = rows per thread = ceil(rows of a row block / 1 024): the per-row state lives in registers (configs[3]: 5, not the 8 the largest
// row block needs -- with 8 the kernel spilled 156 bytes per thread).
template <int RT>
__global__ __launch_bounds__(kPanThreads) void k_pan_build(CsrView A, PanView P) {
    __shared__ double sval[kPanStage];
    __shared__ __attribute__((aligned(16))) unsigned short scol[kPanStage];
    __shared__ int hist[kPanMaxLen + 1], start[kPanMaxLen + 1], fill[kPanMaxLen + 1];
    __shared__ int toff[kPanWork * kPanTW + 1];
    static_assert(kPanStage * 10 + 3 * (kPanMaxLen + 1) * 4 + (kPanWork * kPanTW + 1) * 4 + 256 <= 163840, "the LDS image of a cell must fit one CU");
    int b, p;
    if (!pan_build_bp((int)blockIdx.x, P.NB, P.NP, &b, &p)) return;
    const int tid = threadIdx.x;
    const int R = 64 * P.NTB, NTP = kPanWork * P.TWW;
    const int cell = b * P.NP + p;
    const int c0 = p * P.C;
    if (tid <= kPanMaxLen) { hist[tid] = 0; fill[tid] = 0; }
    __syncthreads();
    int c[RT], st[RT], hole[RT], hlen[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        const int rl = tid + kPanThreads * j, r = b * R + rl;
        st[j] = 0; hole[j] = 0; hlen[j] = 0;
        c[j] = (rl < R && r < A.n) ? pan_row_count(P, A.n, r, p, &st[j], &hole[j], &hlen[j]) : 0;
    }
#pragma unroll
    for (int j = 0; j < RT; ++j)
        if (tid + kPanThreads * j < R) atomicAdd(&hist[c[j]], 1);
    __syncthreads();
    // (the scans below are wave scans / binary searches: as loops of one thread over LDS they were ~15 us of this launch)
    static_assert(kPanMaxLen + 1 == 128, "two length bins per lane of one wave");
    if (tid < 64) {           // start[L] = rows longer than L (descending sort: the longest rows take the first slots)
        const int a = hist[kPanMaxLen - 2 * tid], b2 = hist[kPanMaxLen - 1 - 2 * tid];
        int x = a + b2;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, kWave); if (tid >= o) x += y; }
        start[kPanMaxLen - 2 * tid] = x - a - b2;
        start[kPanMaxLen - 1 - 2 * tid] = x - b2;
    }
    __syncthreads();
    if (tid < NTP) {          // physical tile tid = (w, q) holds sorted tile ts = w + 15 q: its longest row comes first
        const int w = tid / P.TWW, q = tid - w * P.TWW, ts = w + kPanWork * q;
        int tm = 0;
        if (ts < P.NTB) {     // length of the row in sorted slot 64 ts = the largest L with more than 64 ts rows of length >= L
            const int s0 = ts * 64;
            int lo = 0, hi = kPanMaxLen;         // invariant: rows(>= lo) > s0 (lo = 0: all R rows)
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (start[mid] + hist[mid] > s0) lo = mid; else hi = mid - 1; }
            tm = lo;
        }
        toff[tid + 1] = 64 * tm;
    }
    __syncthreads();
    if (tid < 64) {           // prefix of the tile sizes (two tiles per lane: NTP <= 120), then the staging rounds
        const int t0 = 2 * tid, t1 = 2 * tid + 1;
        const int a = t0 < NTP ? toff[t0 + 1] : 0, b2 = t1 < NTP ? toff[t1 + 1] : 0;
        int x = a + b2;
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, kWave); if (tid >= o) x += y; }
        const int base = P.cbase[cell];
        if (t0 < NTP) toff[t0 + 1] = base + x - b2;
        if (t1 < NTP) toff[t1 + 1] = base + x;
        if (tid == 0) toff[0] = base;
    }
    __syncthreads();
    if (tid <= NTP) P.tptr[(size_t)cell * (NTP + 1) + tid] = toff[tid];
    const size_t vtb = (size_t)cell * NTP;
    int dst0[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
        const int rl = tid + kPanThreads * j;
        dst0[j] = 0x3fffffff;
        if (rl < R) {
            const int slot = start[c[j]] + atomicAdd(&fill[c[j]], 1);
            const int ts = slot >> 6, w = ts % kPanWork, q = ts / kPanWork, ph = w * P.TWW + q;
            P.thead[(vtb + (size_t)ph) * 64 + (slot & 63)] = (unsigned short)rl;
            dst0[j] = toff[ph] + (slot & 63);
        }
    }
    const int cend = toff[NTP];
    for (int rb = toff[0]; rb < cend; rb += kPanStage) {
        const int ext = min(kPanStage, cend - rb);
        for (int e = tid; e < ext; e += kPanThreads) { sval[e] = 0.0; scol[e] = 0; }
        __syncthreads();
        // entry i of a row's run inside the panel -> its CSR index (non-band form: entry 0 of a diagonal cell's row is the
        // diagonal; band form: the run skips the hole of columns r - 1 / r + 1); it sits at dst0 + 64 i of the cell.  Lanes with
        // nothing to fetch load entry 0 of the matrix (one line for everybody): unconditional loads stay batched.
#pragma unroll
        for (int j0 = 0; j0 < RT; j0 += 4) {
            constexpr int JN = 4;       // (rows of a batch; the last batch of an RT that is no multiple of 4 masks the rest out below)
            // the entries [lo, hi) of each row that fall into this round, and the longest such range among the wave's rows: the
            // batches below run that often for everybody (a loop per row over ITS run length is one dependent round trip per
            // entry for the whole wave: 70 of the first build's 117 us)
            int lo[4], hi[4], cm = 0;
#pragma unroll
            for (int jj = 0; jj < JN; ++jj) {
                const int j = min(j0 + jj, RT - 1);
                const bool real = j0 + jj < RT;
                lo[jj] = rb > dst0[j] ? (rb - dst0[j] + 63) >> 6 : 0;
                const int t = rb + ext - dst0[j];
                hi[jj] = (real && t > 0) ? min(c[j], (t + 63) >> 6) : 0;
                cm = max(cm, hi[jj] - lo[jj]);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cm = max(cm, __shfl_xor(cm, o, kWave));
            for (int i0 = 0; i0 < cm; i0 += 4) {
                double v[4][4];
                int cc[4][4];
#pragma unroll
                for (int jj = 0; jj < JN; ++jj) {
                    const int j = min(j0 + jj, RT - 1);
                    const int r = b * R + tid + kPanThreads * j;
                    const bool mine = lo[jj] + i0 < hi[jj];
                    const int shift = (!P.band && c[j] > 0 && r / P.C == p) ? 1 : 0;
                    const int dg = (shift && mine && lo[jj] + i0 == 0) ? A.rowptr[min(r, A.n - 1)] : 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ii = min(lo[jj] + i0 + i, max(c[j] - 1, 0));
                        int src = (shift && ii == 0) ? dg : st[j] + ii - shift;
                        if (hlen[j] && src >= hole[j]) src += hlen[j];
                        if (!mine) src = 0;
                        v[jj][i] = A.val[src]; cc[jj][i] = A.col[src];
                    }
                }
#pragma unroll
                for (int jj = 0; jj < JN; ++jj) {
                    const int j = min(j0 + jj, RT - 1);
                    const int d0 = dst0[j] - rb;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = lo[jj] + i0 + i;
                        if (e < hi[jj]) { sval[d0 + 64 * e] = v[jj][i]; scol[d0 + 64 * e] = (unsigned short)(cc[jj][i] - c0); }
                    }
                }
            }
        }
        __syncthreads();
        for (int e = tid; e < ext; e += kPanThreads) P.bval[rb + e] = sval[e];
        {   // columns: four 2-byte entries per 8-byte store (cell ranges start on 64-entry boundaries)
            const unsigned long long* s4 = reinterpret_cast<const unsigned long long*>(scol);
            unsigned long long* d4 = reinterpret_cast<unsigned long long*>(P.bcol + rb);
            for (int e = tid; e < (ext >> 2); e += kPanThreads) d4[e] = s4[e];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Step kernel 1: y_p = L[block b, panel p] v_j
// ------------------------------------------------------------------------------------------
#define PAN_MUL_ARGS(P, L, jrel) (((jrel) & 1) ? (L).Z1 : (L).Z0), (L).part, (L).st, (P).tptr, (P).thead, (P).n, (P).C, (P).NP, (P).TWW, (P), (L), (jrel)
#define PAN_FIN_ARGS(P, L, jrel) (((jrel) & 1) ? (L).Z1 : (L).Z0), (P).coef, (P).ypart, (P).n, (P).NP, (P), (L), (jrel)
// RAW (round 4, diagonally preconditioned LOBPCG on large graphs: precond.h / solver.h): the operand is a plain vector w (passed
// through z_cur), read with 8-byte loads and copied into LDS as it is -- no records, no coefficients, no reduction prologue.
template <int RPT, bool RAW = false>   // records per worker thread: the panel holds at most RPT * 960 columns
__global__ __launch_bounds__(kPanThreads) void k_pan_mul(const Z2* __restrict__ z_cur, double* l_part, LanState* l_st,
                                                          const int* __restrict__ a_tptr, const unsigned short* __restrict__ a_thead,
                                                          int a_n, int a_C, int a_NP, int a_TWW, PanView A_, PipeView L_, int jrel) {
    // (leading scalars = what the first loads need, preloaded into SGPRs with the wave: PAN_MUL_ARGS, cf. PIPE_ARGS in kernels.h)
    PanView A = A_;
    A.tptr = const_cast<int*>(a_tptr); A.thead = const_cast<unsigned short*>(a_thead); A.n = a_n; A.C = a_C; A.NP = a_NP; A.TWW = a_TWW;
    PipeView L = L_;
    L.part = l_part; L.st = l_st;
    __shared__ double sv[RPT * kPanWorkThreads];
    __shared__ double yblk[kPanRows];
    __shared__ double scoef[8];
    static_assert((RPT * kPanWorkThreads + kPanRows + 8) * 8 <= 163840, "panel + row-block image exceed the LDS");
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / A.NP, p = blockIdx.x - b * A.NP;
    const int c0 = p * A.C;
    const int Cp = min(A.C, A.n - c0);           // >= 1 by construction of the plan
    const int R = 64 * A.NTB;
    PAN_CLK(tid == 0, 0); PAN_CLK(tid == 64, 1);
    // Wave 0 does nothing but finish step j-1's reductions (the coefficients everyone waits for); waves 1..15 own the
    // panel's records and their tiles.  (One role per wave also keeps the two register-hungry phases -- 48 partial
    // loads in flight there, records + a wave's chunks here -- out of each other's allocation.)
    if (wv == 0) {
        if (!RAW) {
            int jd;
            const PipeCoef c = pipe_prologue_wave0(L, jrel, -1, scoef, &jd);
            if (blockIdx.x == 0 && lane == 0) {
                A.coef[0] = c.alpha; A.coef[1] = c.beta; A.coef[2] = c.mu; A.coef[3] = c.inv; A.coef[4] = (double)jd;
            }
        }
        PAN_CLK(tid == 0, 3);
        __syncthreads();
        __syncthreads();
        __syncthreads();
        if (RAW) { __syncthreads(); __syncthreads(); }      // (the workers' reduction of w^T L w)
        return;
    }
    const int wt = tid - 64, ww = wv - 1;        // worker thread / worker wave
#ifdef PAN_HEADSTART
    __builtin_amdgcn_s_sleep(PAN_HEADSTART);     // (experiment: let wave 0's partial loads into the memory pipeline first)
#endif
    const Z2* __restrict__ Zc = z_cur + c0;
    // the panel's records, all requested at once (clamped index: unconditional loads stay batched, cf. the prologue)
    Z2 z[RPT];
    if (RAW) {
        const double* __restrict__ wc = reinterpret_cast<const double*>(z_cur) + c0;
#pragma unroll
        for (int i = 0; i < RPT; ++i) { z[i].v = wc[min(wt + kPanWorkThreads * i, Cp - 1)]; z[i].t = 0.0; }
    } else {
#pragma unroll
        for (int i = 0; i < RPT; ++i) z[i] = Zc[min(wt + kPanWorkThreads * i, Cp - 1)];
    }
    // this wave's tiles: slot -> row, and the chunk (64 entries) at which each tile ends
    const int vt0 = ((b * A.NP + p) * kPanWork + ww) * A.TWW;
    int ro[kPanTW], cend[kPanTW];
    int E0;
    {
        int tp[kPanTW + 1];
#pragma unroll
        for (int q = 0; q <= kPanTW; ++q) tp[q] = __builtin_amdgcn_readfirstlane(A.tptr[vt0 + (b * A.NP + p) + min(q, A.TWW)]);      // (tile table: NTP + 1 entries per cell)
#pragma unroll
        for (int q = 0; q < kPanTW; ++q) ro[q] = A.thead[(size_t)(vt0 + min(q, A.TWW - 1)) * 64 + lane];
        E0 = tp[0];
#pragma unroll
        for (int q = 0; q < kPanTW; ++q) cend[q] = (tp[q + 1] - E0) >> 6;      // (tiles q >= TWW: same as the last real one)
    }
    const int nch = cend[kPanTW - 1];
    const double* __restrict__ bv = A.bval + E0 + lane;
    const unsigned short* __restrict__ bc = A.bcol + E0 + lane;
    double pv[kPanCH];
    int pk[kPanCH];
#pragma unroll
    for (int c = 0; c < kPanCH; ++c) { pv[c] = 0.0; pk[c] = 0; }
#pragma unroll
    for (int c = 0; c < kPanCH; ++c)
        if (c < nch) { pv[c] = bv[c * 64]; pk[c] = bc[c * 64]; }
#ifdef PAN_CLOCKS
    if (wv == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PAN_CLK(tid == 64, 2); }   // records, tile table, the wave's chunks arrived (wave 1)
#endif
    __syncthreads();
    PAN_CLK(tid == 64, 4);
    {
        const double alpha = RAW ? 0.0 : scoef[0], mu = RAW ? 0.0 : scoef[2], inv = RAW ? 1.0 : scoef[3];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int c = wt + kPanWorkThreads * i;
            if (c < Cp) sv[c] = RAW ? z[i].v : pan_vj(alpha, mu, inv, z[i].t, z[i].v);
        }
        for (int rl = wt; rl < R; rl += kPanWorkThreads) yblk[rl] = 0.0;      // rows of empty tiles
    }
    __syncthreads();
    PAN_CLK(tid == 64, 5);
    double acc = 0.0;
    for (int cb = 0; cb < nch; cb += kPanCH) {
        // No contraction in this block: a product and its addition to the row's sum stay two roundings for EVERY chunk.
        // (Left to the compiler, some of the 20 unrolled chunk positions were fused into fmas and others not; which positions
        // a row's entries occupy depends on the tile it was dealt to, ties among rows of equal length are dealt in arrival
        // order -- and lambda_2 differed in the last digit from run to run.  Found by tools/soak.sh / tools/det_probe.py.)
#pragma clang fp contract(off)
        if (cb) {     // more chunks than the registers hold: next round (one more round trip)
#pragma unroll
            for (int c = 0; c < kPanCH; ++c)
                if (cb + c < nch) { pv[c] = bv[(cb + c) * 64]; pk[c] = bc[(cb + c) * 64]; }
        }
        // gathers of the whole round back to back (padding entries multiply v_j[first column of the panel] by zero)
#pragma unroll
        for (int c = 0; c < kPanCH; ++c) pv[c] *= sv[min(pk[c], Cp - 1)];
        PAN_CLK(tid == 64 && cb == 0, 6);
        // chunks of this round that close a (non-empty) tile, as a bit mask: one scalar test per chunk instead of a
        // comparison with every tile end (the scalar unit was the bottleneck of this loop)
        unsigned endmask = 0;
#pragma unroll
        for (int q = 0; q < kPanTW; ++q) {
            const int prev = q ? cend[q - 1] : 0, last = cend[q] - 1 - cb;
            if (q < A.TWW && cend[q] > prev && last >= 0 && last < kPanCH) endmask |= 1u << last;
        }
#pragma unroll
        for (int c = 0; c < kPanCH; ++c) {
            if (cb + c < nch) {
                acc += pv[c];
                if (endmask & (1u << c)) {       // its lanes' rows are complete
#pragma unroll
                    for (int q = 0; q < kPanTW; ++q)
                        if (cb + c + 1 == cend[q] && (q == 0 ? cend[0] > 0 : cend[q] > cend[q - 1])) yblk[ro[q]] = acc;
                    acc = 0.0;
                }
            }
        }
    }
    PAN_CLK(tid == 64, 7);
    __syncthreads();
    // the row block's sums, un-sorted by the LDS image: coalesced stores
    double wdot = 0.0;
    for (int rl = wt; rl < R; rl += kPanWorkThreads) {
        const int row = b * R + rl;
        if (row < A.n) {
            A.ypart[(size_t)p * A.n + row] = yblk[rl];
            if (RAW) wdot = __builtin_fma(reinterpret_cast<const double*>(z_cur)[row], yblk[rl], wdot);
        }
    }
    if (RAW) {
        // w^T (L w) of this (row block, panel) cell -- the ONE inner product of the preconditioned iteration that needs the new
        // product (precond.h, k_lob_update_pan: every other sum comes from the previous update by the symmetry of L); slot
        // blockIdx of l_part, summed in slot order by that kernel's prologue.  Fixed order: worker waves 1..15.
        wdot = wave_total(wdot);
        if (lane == 0 && ww < 8) scoef[ww] = wdot;     // (scoef is idle in the RAW instantiation; 15 waves -> two rounds of 8 slots)
        __syncthreads();
        if (lane == 0 && ww >= 8) scoef[ww - 8] += wdot;
        __syncthreads();
        if (tid == 64) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += scoef[k];
            l_part[blockIdx.x] = a;
        }
    }
    PAN_CLK(tid == 64, 8); PAN_CLK(tid == 1023, 9);
}


// Several row blocks per workgroup (round 4): the workgroup keeps its panel in LDS and walks A.CELLS row blocks
// (b = blockIdx / NP + cell * gridDim / NP) -- more row blocks than one wave of workgroups has, without loading the operand again:
// lifts the n <= 145 000 limit of the single-cell kernel (plan_panel).  A cell is a serial chain (tile table -> chunk loads ->
// row sums -> stores); here the NEXT cell's tile table is requested before the current cell's sums and its chunks right behind
// them, in flight across the barrier and the stores -- the barriers order LDS traffic only (lds_barrier: s_waitcnt lgkmcnt(0) +
// s_barrier; a __syncthreads would wait for those loads).  Same per-row arithmetic and order as k_pan_mul.
__device__ __forceinline__ void pan_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int kPanCHM = 14;        // chunks per round in the multi-cell kernel (the prefetched tile table needs the registers)
template <int RPT>
__global__ __launch_bounds__(kPanThreads) void k_pan_mul_multi(const Z2* __restrict__ z_cur, double* l_part, LanState* l_st,
                                                                const int* __restrict__ a_tptr, const unsigned short* __restrict__ a_thead,
                                                                int a_n, int a_C, int a_NP, int a_TWW, PanView A_, PipeView L_, int jrel) {
    PanView A = A_;
    A.tptr = const_cast<int*>(a_tptr); A.thead = const_cast<unsigned short*>(a_thead); A.n = a_n; A.C = a_C; A.NP = a_NP; A.TWW = a_TWW;
    PipeView L = L_;
    L.part = l_part; L.st = l_st;
    __shared__ double sv[RPT * kPanWorkThreads];
    __shared__ double yblk[kPanRows];
    __shared__ double scoef[8];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bg = blockIdx.x / A.NP, p = blockIdx.x - bg * A.NP;
    const int ncell = A.CELLS, nbg = (int)gridDim.x / A.NP;
    const int c0 = p * A.C;
    const int Cp = min(A.C, A.n - c0);
    const int R = 64 * A.NTB;
    if (wv == 0) {
        int jd;
        const PipeCoef c = pipe_prologue_wave0(L, jrel, -1, scoef, &jd);
        if (blockIdx.x == 0 && lane == 0) {
            A.coef[0] = c.alpha; A.coef[1] = c.beta; A.coef[2] = c.mu; A.coef[3] = c.inv; A.coef[4] = (double)jd;
        }
        for (int cell = 0; cell < ncell; ++cell) { pan_lds_barrier(); pan_lds_barrier(); pan_lds_barrier(); }
        return;
    }
    const int wt = tid - 64, ww = wv - 1;
    const Z2* __restrict__ Zc = z_cur + c0;
    Z2 z[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) z[i] = Zc[min(wt + kPanWorkThreads * i, Cp - 1)];
    // tile table of a cell: tile ends (uniform) and slot -> row of this wave's tiles
    int tp[kPanTW + 1], ro[kPanTW];
    {
        const int vt0 = ((min(bg, A.NB - 1) * A.NP + p) * kPanWork + ww) * A.TWW;
#pragma unroll
        for (int q = 0; q <= kPanTW; ++q) tp[q] = __builtin_amdgcn_readfirstlane(A.tptr[vt0 + (min(bg, A.NB - 1) * A.NP + p) + min(q, A.TWW)]);
#pragma unroll
        for (int q = 0; q < kPanTW; ++q) ro[q] = A.thead[(size_t)(vt0 + min(q, A.TWW - 1)) * 64 + lane];
    }
    int cend[kPanTW];
    int E0 = tp[0];
#pragma unroll
    for (int q = 0; q < kPanTW; ++q) cend[q] = (tp[q + 1] - E0) >> 6;
    int nch = bg < A.NB ? cend[kPanTW - 1] : 0;
    double pv[kPanCHM];
    int pk[kPanCHM];
#pragma unroll
    for (int c = 0; c < kPanCHM; ++c) { pv[c] = 0.0; pk[c] = 0; }
#pragma unroll
    for (int c = 0; c < kPanCHM; ++c)
        if (c < nch) { pv[c] = A.bval[E0 + lane + c * 64]; pk[c] = A.bcol[E0 + lane + c * 64]; }
    pan_lds_barrier();           // the coefficients are there
    {                            // (before the loop: the records' registers are free from here on)
        const double alpha = scoef[0], mu = scoef[2], inv = scoef[3];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int c = wt + kPanWorkThreads * i;
            if (c < Cp) sv[c] = pan_vj(alpha, mu, inv, z[i].t, z[i].v);
        }
    }
    for (int cell = 0; cell < ncell; ++cell) {
        const int b = bg + cell * nbg;
        const bool live = b < A.NB;                  // (the last cells of some workgroups have no row block)
        const double* __restrict__ bv = A.bval + E0 + lane;
        const unsigned short* __restrict__ bc = A.bcol + E0 + lane;
        if (cell) pan_lds_barrier();                 // the previous cell's stores have read yblk
        for (int rl = wt; rl < R; rl += kPanWorkThreads) yblk[rl] = 0.0;
        pan_lds_barrier();
        // the next cell's tile table: in flight while this cell's rows are summed
        const int bn = bg + (cell + 1) * nbg;
        const bool more = cell + 1 < ncell;
        int tpn[kPanTW + 1], ron[kPanTW];
        {
            const int vtn = ((min(more ? bn : b, A.NB - 1) * A.NP + p) * kPanWork + ww) * A.TWW;
#pragma unroll
            for (int q = 0; q <= kPanTW; ++q) tpn[q] = A.tptr[vtn + (min(more ? bn : b, A.NB - 1) * A.NP + p) + min(q, A.TWW)];
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) ron[q] = A.thead[(size_t)(vtn + min(q, A.TWW - 1)) * 64 + lane];
        }
        double acc = 0.0;
        for (int cb = 0; cb < nch; cb += kPanCHM) {
#pragma clang fp contract(off)      // (two roundings at every chunk position: see k_pan_mul)
            if (cb) {
#pragma unroll
                for (int c = 0; c < kPanCHM; ++c)
                    if (cb + c < nch) { pv[c] = bv[(cb + c) * 64]; pk[c] = bc[(cb + c) * 64]; }
            }
#pragma unroll
            for (int c = 0; c < kPanCHM; ++c) pv[c] *= sv[min(pk[c], Cp - 1)];
            unsigned endmask = 0;
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) {
                const int prev = q ? cend[q - 1] : 0, last = cend[q] - 1 - cb;
                if (q < A.TWW && cend[q] > prev && last >= 0 && last < kPanCHM) endmask |= 1u << last;
            }
#pragma unroll
            for (int c = 0; c < kPanCHM; ++c) {
                if (cb + c < nch) {
                    acc += pv[c];
                    if (endmask & (1u << c)) {
#pragma unroll
                        for (int q = 0; q < kPanTW; ++q)
                            if (cb + c + 1 == cend[q] && (q == 0 ? cend[0] > 0 : cend[q] > cend[q - 1])) yblk[ro[q]] = acc;
                        acc = 0.0;
                    }
                }
            }
        }
        // the next cell's chunks: requested now, in flight across the barrier and this cell's stores
        if (more) {
#pragma unroll
            for (int q = 0; q <= kPanTW; ++q) tp[q] = __builtin_amdgcn_readfirstlane(tpn[q]);
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) ro[q] = ron[q];
            E0 = tp[0];
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) cend[q] = (tp[q + 1] - E0) >> 6;
            nch = bn < A.NB ? cend[kPanTW - 1] : 0;
#pragma unroll
            for (int c = 0; c < kPanCHM; ++c) { pv[c] = 0.0; pk[c] = 0; }
#pragma unroll
            for (int c = 0; c < kPanCHM; ++c)
                if (c < nch) { pv[c] = A.bval[E0 + lane + c * 64]; pk[c] = A.bcol[E0 + lane + c * 64]; }
        }
        pan_lds_barrier();
        for (int rl = wt; rl < R; rl += kPanWorkThreads) {
            const int row = b * R + rl;
            if (live && row < A.n) A.ypart[(size_t)p * A.n + row] = yblk[rl];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Step kernel 2: w = sum_p y_p, record / basis column / inner products of the step (row-parallel)
// ------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_pan_fin(const Z2* __restrict__ z_cur, const double* __restrict__ a_coef,
                                                    const double* __restrict__ a_ypart, int a_n, int a_NP, PanView A_, PipeView L, int jrel) {
    __shared__ double smw[kNP * BLOCK];
    PanView A = A_;
    A.coef = const_cast<double*>(a_coef); A.ypart = const_cast<double*>(a_ypart); A.n = a_n; A.NP = a_NP;     // (PAN_FIN_ARGS: preloaded)
    const double alpha = A.coef[0], beta = A.coef[1], mu = A.coef[2], inv = A.coef[3];
    const int j = (int)A.coef[4];
    const Z2* __restrict__ Zc = z_cur;
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
        if (A.band) {       // the tridiagonal band, kept out of the panel form (it sat in the diagonal cells only: 2.2x the entries of any other cell)
            const Z2 zl = Zc[max(r - 1, 0)], zu = Zc[min(r + 1, n - 1)];
            w = __builtin_fma(A.bd[r], o.v, w);
            w = __builtin_fma(A.bd[(size_t)n + r], pan_vj(alpha, mu, inv, zl.t, zl.v), w);
            w = __builtin_fma(A.bd[2 * (size_t)n + r], pan_vj(alpha, mu, inv, zu.t, zu.v), w);
        }
        o.t = __builtin_fma(-beta, z.v, w);         // Paige's intermediate for the next step (explicit fma: k_pan_step rounds alike)
        vj[r] = o.v;
        Zn[r] = o;
        const double t = o.t, v = o.v;
        pr.acc[0] = __builtin_fma(t, t, pr.acc[0]); pr.acc[1] = __builtin_fma(t, v, pr.acc[1]); pr.acc[2] = __builtin_fma(v, v, pr.acc[2]);
        pr.acc[3] += t; pr.acc[4] += v; pr.acc[5] += fabs(v);
    }
    pr.template store<BLOCK>(L, jrel, smw);
}


#ifdef MACHIP_EXPERIMENTS      // (measured slower than the two-launch step: profiles/r4_c4_one_launch_step.md; tools/ubench7.hip builds it)
// ------------------------------------------------------------------------------------------
// ONE launch per step (round 4): k_pan_mul's product + k_pan_fin's row work behind a per-row-block arrival ticket.
// ------------------------------------------------------------------------------------------
// The NP workgroups (b, 0..NP-1) of row block b publish their partial products y_p[rows of b] with WRITE-THROUGH stores
// (`sc1`: the bytes are in memory when the store has been acknowledged -- no release fence, which would write back the
// whole XCD's L2; MI355X_MICROARCH.md "publish-large": 3.0 against 8.2 us), drain them (s_waitcnt vmcnt(0)), and take ONE
// device-scope ticket of the row block (monotonic counter: step j of a sequence is complete at (j + 1) NP arrivals; the
// solver zeroes the counters when it starts a sequence).  The row block's 64 NTB rows are cut into NP slices; slice p is
// finished -- w = sum_p y_p[r] in panel order, Paige's t_j, the record, the basis column, the six measured inner products:
// the arithmetic of k_pan_fin, row for row -- by whoever CLAIMS it (compare-and-swap on a per-slice word, j -> j + 1):
//   * a workgroup that sees its row block complete within `spin_ticks` claims its own slice p (the normal case: all 252
//     workgroups are resident, the arrival skew inside a row block is 1-3 us);
//   * the LAST arriver (ticket == (j+1) NP - 1) takes its own slice and then every slice nobody has taken;
//   * everybody else leaves.  Nobody waits for a workgroup that has not been scheduled: no co-residency assumption, no
//     deadlock when other streams (evaluation lanes) share the chip; every slice is finished exactly once, by identical
//     arithmetic whoever does it, and its six sums go to the slice's own slot of the partial-sum array (P = NB NP) --
//     bit-reproducible.
// The partials are read back with `sc1` loads (served by L2 / memory, never by this CU's L1).
// 16-byte write-through store / L1-bypassing load of two consecutive partials (8-byte `sc1` accesses run at ~0.4-0.6 of the
// 16-byte rate; every row block and slice starts at an even row, the partial-product planes have an even stride)
typedef double pan_d2 __attribute__((ext_vector_type(2)));
struct PanPair { double a, b; };
__device__ __forceinline__ void pan_store_wt2(double* p, double a, double b) {
    pan_d2 v; v.x = a; v.y = b;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
// (the caller waits: s_waitcnt vmcnt(0) behind a batch of these -- the compiler does not count asm loads)
__device__ __forceinline__ pan_d2 pan_load_wt2(const double* p) {
    pan_d2 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
    return r;
}

// XCD-LOCAL variants (k_pan_step<RPT, true>: all NP workgroups of a row block run on ONE XCD and share its L2): plain 16-byte
// stores (the L1 writes through, `s_waitcnt vmcnt(0)` = acknowledged by the L2), loads that bypass this CU's L1 (`sc0`), tickets and
// claims as L2-local atomics (no `sc1`: they execute in the XCD's L2 and never travel to the fabric).
__device__ __forceinline__ pan_d2 pan_load_l2(const double* p) {
    pan_d2 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ unsigned pan_poll_l2(unsigned* p) {      // returning add of 0, executed by the L2 (an atomic LOAD of workgroup scope may hit the L1)
    unsigned r, z = 0u;
    asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory");
    return r;
}
__device__ __forceinline__ int pan_xcc_id() { return (int)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15; }      // HW_REG_XCC_ID[3:0]

#define PAN_STEP_ARGS(P, L, jrel) (((jrel) & 1) ? (L).Z1 : (L).Z0), (L).part, (L).st, (P).tptr, (P).thead, (P).n, (P).C, (P).NP, (P).TWW, (P), (L), (jrel)
template <int RPT, bool LOCAL = false>
__global__ __launch_bounds__(kPanThreads) void k_pan_step(const Z2* __restrict__ z_cur, double* l_part, LanState* l_st,
                                                           const int* __restrict__ a_tptr, const unsigned short* __restrict__ a_thead,
                                                           int a_n, int a_C, int a_NP, int a_TWW, PanView A_, PipeView L_, int jrel) {
    PanView A = A_;
    A.tptr = const_cast<int*>(a_tptr); A.thead = const_cast<unsigned short*>(a_thead); A.n = a_n; A.C = a_C; A.NP = a_NP; A.TWW = a_TWW;
    PipeView L = L_;
    L.part = l_part; L.st = l_st;
    __shared__ double sv[RPT * kPanWorkThreads];
    __shared__ double yblk[kPanRows];        // row-block image of the product phase; scratch of the slice epilogue afterwards
    __shared__ double scoef[8];
    __shared__ unsigned int sflag[4];
    static_assert((RPT * kPanWorkThreads + kPanRows + 8) * 8 + 16 <= 163840, "panel + row-block image exceed the LDS");
    static_assert(kNP * kPanWaves * 64 <= kPanRows, "slice epilogue scratch must fit the row-block image");
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, p;
    if (LOCAL) {        // workgroup w runs on XCD w mod 8: XCD x serves the row blocks x bpx .. x bpx + bpx - 1 (bpx NP <= 32 CUs)
        const int bpx = (A.NB + 7) >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = xcd * bpx + slot / A.NP; p = slot % A.NP;
        if (slot >= bpx * A.NP || b >= A.NB) return;
    } else { b = blockIdx.x / A.NP; p = blockIdx.x - b * A.NP; }
    const int c0 = p * A.C;
    const int Cp = min(A.C, A.n - c0);
    const int R = 64 * A.NTB;
    PAN_CLK(tid == 0, 0); PAN_CLK(tid == 64, 1);
#ifdef PAN_CLOCKS
    if (tid == 0) A.clk[blockIdx.x * 16 + 12] = pan_xcc_id();
#endif
    if (wv == 0) {      // wave 0: step j-1's reductions (the coefficients everyone waits for)
        int jd;
        (void)pipe_prologue_wave0(L, jrel, -1, scoef, &jd);
        PAN_CLK(tid == 0, 3);
        __syncthreads();
        __syncthreads();
        __syncthreads();
    } else {
        const int wt = tid - 64, ww = wv - 1;
        const Z2* __restrict__ Zc = z_cur + c0;
        Z2 z[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) z[i] = Zc[min(wt + kPanWorkThreads * i, Cp - 1)];
        const int vt0 = ((b * A.NP + p) * kPanWork + ww) * A.TWW;
        int ro[kPanTW], cend[kPanTW];
        int E0;
        {
            int tp[kPanTW + 1];
#pragma unroll
            for (int q = 0; q <= kPanTW; ++q) tp[q] = __builtin_amdgcn_readfirstlane(A.tptr[vt0 + (b * A.NP + p) + min(q, A.TWW)]);      // (tile table: NTP + 1 entries per cell)
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) ro[q] = A.thead[(size_t)(vt0 + min(q, A.TWW - 1)) * 64 + lane];
            E0 = tp[0];
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) cend[q] = (tp[q + 1] - E0) >> 6;
        }
        const int nch = cend[kPanTW - 1];
        const double* __restrict__ bv = A.bval + E0 + lane;
        const unsigned short* __restrict__ bc = A.bcol + E0 + lane;
        double pv[kPanCH];
        int pk[kPanCH];
#pragma unroll
        for (int c = 0; c < kPanCH; ++c) { pv[c] = 0.0; pk[c] = 0; }
#pragma unroll
        for (int c = 0; c < kPanCH; ++c)
            if (c < nch) { pv[c] = bv[c * 64]; pk[c] = bc[c * 64]; }
        __syncthreads();
        PAN_CLK(tid == 64, 4);
        {
            const double alpha = scoef[0], mu = scoef[2], inv = scoef[3];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int c = wt + kPanWorkThreads * i;
                if (c < Cp) sv[c] = pan_vj(alpha, mu, inv, z[i].t, z[i].v);
            }
            for (int rl = wt; rl < R; rl += kPanWorkThreads) yblk[rl] = 0.0;
        }
        __syncthreads();
        PAN_CLK(tid == 64, 5);
        double acc = 0.0;
        for (int cb = 0; cb < nch; cb += kPanCH) {
#pragma clang fp contract(off)      // (two roundings at every chunk position: see k_pan_mul)
            if (cb) {
#pragma unroll
                for (int c = 0; c < kPanCH; ++c)
                    if (cb + c < nch) { pv[c] = bv[(cb + c) * 64]; pk[c] = bc[(cb + c) * 64]; }
            }
#pragma unroll
            for (int c = 0; c < kPanCH; ++c) pv[c] *= sv[min(pk[c], Cp - 1)];
            PAN_CLK(tid == 64 && cb == 0, 6);
            unsigned endmask = 0;
#pragma unroll
            for (int q = 0; q < kPanTW; ++q) {
                const int prev = q ? cend[q - 1] : 0, last = cend[q] - 1 - cb;
                if (q < A.TWW && cend[q] > prev && last >= 0 && last < kPanCH) endmask |= 1u << last;
            }
#pragma unroll
            for (int c = 0; c < kPanCH; ++c) {
                if (cb + c < nch) {
                    acc += pv[c];
                    if (endmask & (1u << c)) {
#pragma unroll
                        for (int q = 0; q < kPanTW; ++q)
                            if (cb + c + 1 == cend[q] && (q == 0 ? cend[0] > 0 : cend[q] > cend[q - 1])) yblk[ro[q]] = acc;
                        acc = 0.0;
                    }
                }
            }
        }
        PAN_CLK(tid == 64, 7);
        __syncthreads();
        // the row block's sums of this panel, un-sorted by the LDS image: coalesced WRITE-THROUGH stores, drained
        {
            const size_t ys = (size_t)((A.n + 1) & ~1);            // plane stride (even: 16-byte aligned pairs)
            for (int rl = 2 * wt; rl < R; rl += 2 * kPanWorkThreads) {
                const int row = b * R + rl;
                if (row < A.n) {      // (R is even; a pair may run one past n: the plane is padded)
                    if (LOCAL) { pan_d2 v2; v2.x = yblk[rl]; v2.y = yblk[rl + 1]; *reinterpret_cast<pan_d2*>(A.ypart + (size_t)p * ys + row) = v2; }
                    else pan_store_wt2(A.ypart + (size_t)p * ys + row, yblk[rl], yblk[rl + 1]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PAN_CLK(tid == 64, 8);
    }
    __syncthreads();                      // every partial of this workgroup is in memory
    const int j = (int)scoef[4];
    const unsigned target = (unsigned)(j + 1) * (unsigned)A.NP;
    if (tid == 0) {
        const unsigned old = LOCAL ? __hip_atomic_fetch_add(A.tick + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                   : __hip_atomic_fetch_add(A.tick + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned state = old + 1 == target ? 2u : 0u;       // 2: last arriver of the row block
        if (!state && A.spin_ticks > 0) {
            const long long t0 = wall_clock64();
            do {
                if ((LOCAL ? pan_poll_l2(A.tick + b) : __hip_atomic_load(A.tick + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= target) { state = 1u; break; }
                __builtin_amdgcn_s_sleep(8);
            } while (wall_clock64() - t0 < (long long)A.spin_ticks);
        }
        sflag[0] = state;
    }
    __syncthreads();
    const unsigned state = sflag[0];
    PAN_CLK(tid == 64, 9);
    if (!state) return;                   // the row block is not complete and this workgroup has waited long enough
    // ---- slices of row block b: own slice first; the last arriver then sweeps the rest ----
    const int S = ((R + A.NP - 1) / A.NP + 1) & ~1;  // rows per slice (even)
    const double alpha = scoef[0], beta = scoef[1], mu = scoef[2], inv = scoef[3];
    const Z2* __restrict__ Zr = z_cur;
    Z2* __restrict__ Zn = (jrel & 1) ? L.Z0 : L.Z1;
    double* __restrict__ vj = L.V + (size_t)j * (size_t)L.n;
    const int n = A.n, NP = A.NP;
    const int nsweep = state == 2u ? NP : 1;
    for (int k = 0; k < nsweep; ++k) {
        const int sl = (p + k) % NP;
        if (tid == 0) {
            unsigned expect = (unsigned)j;
            sflag[1] = (LOCAL ? __hip_atomic_compare_exchange_strong(A.claim + (size_t)b * NP + sl, &expect, (unsigned)(j + 1), __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                              : __hip_atomic_compare_exchange_strong(A.claim + (size_t)b * NP + sl, &expect, (unsigned)(j + 1), __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ? 1u : 0u;
        }
        __syncthreads();
        const bool mine = sflag[1] != 0u;
        __syncthreads();
        if (!mine) continue;
        PipeRow pr;
        pr.clear();
        const size_t ys = (size_t)((n + 1) & ~1);
        const int r_lo = b * R + sl * S, r_hi = min(min(r_lo + S, (b + 1) * R), n);     // (S even: pairs of rows)
        for (int r = r_lo + 2 * tid; r < r_hi; r += 2 * kPanThreads) {
            const bool two = r + 1 < r_hi;
            const Z2 z0 = Zr[r], z1 = Zr[two ? r + 1 : r];
            double w0 = 0.0, w1 = 0.0;
            for (int p0 = 0; p0 < NP; p0 += 12) {     // twelve panels in flight; added in panel order
                pan_d2 y[12];
#pragma unroll
                for (int q = 0; q < 12; ++q) y[q] = LOCAL ? pan_load_l2(A.ypart + (size_t)min(p0 + q, NP - 1) * ys + r) : pan_load_wt2(A.ypart + (size_t)min(p0 + q, NP - 1) * ys + r);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]),
                             "+v"(y[8]), "+v"(y[9]), "+v"(y[10]), "+v"(y[11]) :: "memory");      // (ties the values to the wait)
#pragma unroll
                for (int q = 0; q < 12; ++q) { w0 += (p0 + q < NP) ? y[q].x : 0.0; w1 += (p0 + q < NP) ? y[q].y : 0.0; }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h && !two) break;
                const Z2 z = h ? z1 : z0;
                double w = h ? w1 : w0;
                Z2 o;
                o.v = pan_vj(alpha, mu, inv, z.t, z.v);
                if (A.band) {       // (the band terms exactly as k_pan_fin adds them)
                    const int rr = r + h;
                    const Z2 zl = Zr[max(rr - 1, 0)], zu = Zr[min(rr + 1, n - 1)];
                    w = __builtin_fma(A.bd[rr], o.v, w);
                    w = __builtin_fma(A.bd[(size_t)n + rr], pan_vj(alpha, mu, inv, zl.t, zl.v), w);
                    w = __builtin_fma(A.bd[2 * (size_t)n + rr], pan_vj(alpha, mu, inv, zu.t, zu.v), w);
                }
                o.t = __builtin_fma(-beta, z.v, w);      // Paige's intermediate for the next step
                vj[r + h] = o.v;
                Zn[r + h] = o;
                const double t = o.t, v = o.v;
                pr.acc[0] = __builtin_fma(t, t, pr.acc[0]); pr.acc[1] = __builtin_fma(t, v, pr.acc[1]); pr.acc[2] = __builtin_fma(v, v, pr.acc[2]);
                pr.acc[3] += t; pr.acc[4] += v; pr.acc[5] += fabs(v);
            }
        }
        // six sums of the slice -> its slot (b NP + sl) of the partial-sum array (PipeRow::store with the slot made explicit)
        {
            constexpr int NW = kPanWaves;
#pragma unroll
            for (int q = 0; q < kNP; ++q) yblk[(q * NW + wv) * 64 + lane] = pr.acc[q];
            __syncthreads();
            for (int q = wv; q < kNP; q += NW) {
                double sacc = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) sacc += yblk[(q * NW + w) * 64 + lane];
                sacc = wave_total(sacc);
                if (lane == 0) L.part[(size_t)((jrel + 1) & 1) * (kNP * kMaxGrid) + q * kMaxGrid + (b * NP + sl)] = sacc;
            }
            __syncthreads();
        }
    }
    PAN_CLK(tid == 64, 10); PAN_CLK(tid == 1023, 11);
}

#endif   // MACHIP_EXPERIMENTS

}  // namespace machip
