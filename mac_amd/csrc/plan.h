// plan.h -- launch-shape POLICY of the eigen-solver and the dispatch onto the kernel instantiations (mechanism lives in
// kernels.h / panel.h / persist.h, the driver in solver.h).
//
//   plan_spmv   shape of the stand-alone SpMV kernels (explicit residual check, preconditioned mode)
//   plan_pipe   shape of the one-kernel Lanczos step (lanes per row, load chains, workgroup size, deferred barrier)
//   plan_panel  whether the column-panel form runs, and its panels / row blocks (panel.h)
//   launch_*    the switch from a plan to a template instantiation; launch_pipe_shard: one rank's share of a
//               row-partitioned step (ShardGroup below)
//
// Every number here was measured on MI355X (tools/ubench*.hip, tools/sweep_pipe.py, tools/sweep_panel.py); the comments say
// which.  Shapes can be overridden through the handle's option table (options.h, machip_set_option; OPT(name, default) below):
// a plan is made once per eigen-solve from the handle's table, so the sweep tools and the variant tests switch shapes between
// solves of one process.  Nothing in here looks at timings or at the environment, so a plan -- and with it every rounding of a
// trajectory -- is a function of (n, nnz, longest row, options) only.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "kernels.h"
#include "options.h"
#include "panel.h"
#include "panel_u.h"
#include "persist.h"

namespace machip {

template <class T>
inline int dev_alloc(T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc((void**)p, count * sizeof(T)));
    return MACHIP_OK;
}

enum SpmvVariant { kAuto = 0, kStream = 1, kVec = 2, kPanel = 3, kEll = 4 };   // kEll: k_pipe_vec on the padded fixed-width copy (4 lanes per row)

struct SpmvPlan {
    int variant = kVec;   // kStream or kVec
    int width = 8;        // TPR for stream, G for vec
    int grid = 1;
    int block = kBlock;   // threads per workgroup of the fused step kernel (256, 512 or 1024)
    int unroll = 1;       // independent (val, col, gather) chains per lane
    int defer = 1;        // row tiles whose finish() waits behind the barrier (k_pipe_vec DEFER): 3 where a workgroup owns several
};

// Every workgroup of a step kernel re-reduces the previous step's per-workgroup partials, so
// that traffic grows with grid^2: cap the grid (grid-stride loops cover the rest of the rows).
inline int grid_cap(const Options& opt) { return std::max(1, std::min(kMaxGrid, OPT(maxgrid, 256))); }

inline SpmvPlan plan_spmv(const Options& opt, int n, long nnz, int forced_variant, int cap = 0) {
    if (cap <= 0) cap = grid_cap(opt);
    SpmvPlan pl;
    const double mean = n > 0 ? (double)nnz / (double)n : 1.0;
    int variant = forced_variant;
    if (variant == kAuto) {
        const int e = OPT(spmv, 0);       // 0 automatic, 1 LDS row tiles ("stream"), 2 sub-wave groups ("vec")
        if (e == 1) variant = kStream;
        else if (e == 2) variant = kVec;
        else variant = mean < 24.0 ? kStream : kVec;
    }
    pl.variant = variant;
    if (variant == kStream) {
        int tpr = 16;
        while (tpr > 1 && ((long)n * tpr / kBlock > 4L * cap || tpr > std::max(2.0, mean))) tpr >>= 1;
        tpr = OPT(tpr, tpr);
        pl.width = tpr;
        const int R = kBlock / tpr;
        pl.grid = (int)std::min<long>(cap, ((long)n + R - 1) / R);
    } else {
        int g = 4;
        while (g < 64 && g < mean * 0.75) g <<= 1;
        if (n <= 32768) g = std::min(g, 16);   // cache-resident operand: narrower groups, more rows in flight (17 -> 7 us at config 2)
        g = OPT(g, g);
        pl.width = g;
        const int gpb = kBlock / g;
        pl.grid = (int)std::min<long>(cap, ((long)n + gpb - 1) / gpb);
    }
    if (pl.grid < 1) pl.grid = 1;
    return pl;
}

// Launch shape of the fused Lanczos-step kernel (tools/ubench.hip ablations, MI355X):
// sub-wave groups of 4 lanes per row up to ~40 nnz/row, 16 beyond; at most grid_cap(opt) workgroups
// (each re-reads every workgroup's partials), so large n gets 1024-thread workgroups instead of
// more of them.
inline SpmvPlan plan_pipe(const Options& opt, int n, long nnz, int maxlen) {
    SpmvPlan pl;
    const double mean = n > 0 ? (double)nnz / (double)n : 1.0;
    if (OPT(spmv, 0) == 1) {
        pl = plan_spmv(opt, n, nnz, kStream);
        pl.block = kBlock;
        return pl;
    }
    pl.variant = kVec;
    // tools/ubench.hip, MI355X: 4 lanes per row up to ~16 nnz/row, 8 up to ~32 (and for every
    // n <= 32k, where the gather operand is cache resident), 16 beyond; two independent load chains
    // per lane from 8 lanes up.  Hub rows (Frank-Wolfe vertices concentrate the selected edges on
    // few nodes) would serialise a narrow group: widen when the longest row is far above the mean.
    // Round 2, in-solve sweep (tools/sweep_pipe.py, step time from the events around the Krylov chunks):
    //   n <= 32k (config 2): 8 lanes, 4 load chains, 512-thread workgroups (179 of them at n = 10k) are best or tied
    //     on every iterate, hub rows included: 7.2 us against 7.6 us step-weighted, 5.6 against 6.9 us on the first;
    //   n = 100k (config 4): 4 lanes up to 16 nnz/row -- also when a Frank-Wolfe vertex has produced hub rows, where
    //     the 16-lane groups chosen in round 1 sat mostly idle (14.0 against 17.4 us) -- two load chains once hubs
    //     exist; 8 lanes up to 24 nnz/row, 16 beyond.
    int g, unr, blk;
    if (n <= 32768) {
        g = 8; unr = 4; blk = 512;
        if (mean < 8.0) { g = 4; unr = 2; blk = 256; }   // pose-graph rows (city10000: 5 nnz/row): 5.2 against 5.4 us
    } else {
        g = mean < 16.0 ? 4 : (mean < 24.0 ? 8 : 16);
        if (g == 8 && maxlen > 8 * mean && maxlen > 128) g = 16;
        // (round 4: FOUR chains while the mean stays below 12 -- a Frank-Wolfe vertex gives 4 % of the rows 40-65 entries, a 4-lane group
        // walks them in 16 rounds while its wave idles along, profiles/r4_hub_rows_pmc.txt: configs[3] iterates 1-2 14.1 -> 13.1 / 13.5 us)
        unr = g == 4 ? (maxlen > 48 ? (mean < 12.0 ? 4 : 2) : 1) : 2;
        // at most grid_cap(opt) workgroups (each re-reads every workgroup's partials): grow the workgroup instead of
        // the grid (wave 0 of every workgroup only runs the prologue: BLOCK - 64 threads own rows)
        blk = 256;
        while (blk < 1024 && ((long)n + ((blk - 64) / g) - 1) / ((blk - 64) / g) > grid_cap(opt)) blk <<= 1;
    }
    pl.width = OPT(g, g);
    pl.unroll = OPT(unroll, unr);
    pl.block = OPT(block, blk);
    const int gpb = (pl.block - 64) / pl.width;
    const long tiles = ((long)n + gpb - 1) / gpb;
    pl.grid = (int)std::max<long>(1, std::min<long>(grid_cap(opt), tiles));
    // a workgroup with several row tiles keeps the first three un-finished behind the barrier (the prologue's latency
    // is then hidden); with one tile per workgroup that only costs registers
    pl.defer = OPT(defer, tiles > 2L * pl.grid ? 3 : 1);
    return pl;
}

// Column-panel step (panel.h): shape of the panel form for a matrix of n rows.  NP panels of C <= 16 384 columns (8 bytes
// per column in LDS), NB row blocks of NTB 64-row tiles (TWW per worker wave), NB * NP <= 256 workgroups (one per CU: the panel takes most of
// the CU's LDS).  Panel loads cost NB x 16 n bytes of coalesced L2 traffic per step, the partials 2 x NP x 8 n bytes: the
// defaults balance the two (MACHIP_PANEL_NP / MACHIP_PANEL_NB override; swept on MI355X, profiles/r3_c4_panel.md).
struct PanPlan {
    bool on = false;
    int NP = 1, C = 1, NB = 1, NTB = 1, TWW = 1, RPT = 1;
    int cells = 1;                   // row blocks per workgroup (> 1: k_pan_mul_multi, grid = NP * ceil(NB / cells))
    int grid2 = 1, block2 = 256;     // launch shape of k_pan_fin
    bool band = false;               // diagonal + columns r -/+ 1 kept out of the tiles and added by k_pan_fin (evens out the diagonal cells)
    bool verify = false;             // a row is longer than 127 entries: the build must confirm that no (row, panel) count exceeds 127
    bool fused = false;              // one launch per step (k_pan_step) instead of k_pan_mul + k_pan_fin (measured SLOWER: profiles/r4_c4_one_launch_step.md)
    bool u = false;                  // shifted recurrence with an 8-byte operand (panel_u.h: k_pan_mul8<LPT, TWT> + k_pan_finu); C is even then
    int LPT = 1, TWT = 8;            // ... 16-byte operand loads per thread, tiles per worker wave the instantiation holds (3 | 5 | 8)
};
constexpr int kPanUCmax = pan_u_cols(9, 3);      // widest panel any k_pan_mul8 instantiation holds (17 568 columns)
// nnz_cap: the most entries the handle's L(x) can ever hold (decides the band form once per handle); shape_only: the shape the
// automatic mode WOULD take for this n, whatever nnz is (the assembly writes the per-row tables before nnz is known, kernels.h PanSpec)
// allow_u: the shifted recurrence may be planned (its panels may be wider than the record form's 12 480 columns)
inline PanPlan plan_panel(const Options& opt, int n, long nnz, int maxlen, bool allowed, long nnz_cap = -1, bool shape_only = false, bool allow_u = true) {
    PanPlan pp;
    const bool want_u = allow_u && OPT(panel_u, 1) != 0;
    const int mode = OPT(panel, -1);     // -1 auto, 0 off, 1 forced (tests: small graphs with several panels)
    if (!allowed || mode == 0 || n < 128) return pp;
    const double mean = (double)nnz / (double)std::max(n, 1);
    // automatic: the operand must be too large for the gather path's caches to serve cheaply, and the matrix dense enough
    // for the panel step's fixed costs (two launches, NB x 16 n bytes of panel loads, 2 x NP x 8 n bytes of partials) to
    // pay: measured cross-over on MI355X at n = 1e5 (tools/sweep_panel.py, profiles/r3_c4_panel.md): ~17 entries per row
    // (gather step 18.7 us and rising 4 us per million entries, panel step 18.3 us and rising 0.9 us per million)
    // The build kernels' length histograms describe at most 127 entries of a row INSIDE ONE PANEL.  A row of up to 127 entries fits
    // whatever its columns are; a longer (hub) row fits when its entries spread over the panels -- k_pan_rows checks every (row, panel)
    // count and raises a flag, the solver then drops the panel form for this matrix (verify).  Rows beyond 64 x 127 entries: gather step.
    if (maxlen > 64 * kPanMaxLen) return pp;
    pp.verify = maxlen > kPanMaxLen;
    // (other sizes, tools/panel_size_probe.py: n = 66 000 .. 145 000 with a single wave of workgroups -- a tie at 23-26 entries
    // per row, 1.3-1.5x at 43-46; beyond n = 1e5 the build's 0.25 ms per solve moves the break-even to ~26)
    // (shifted recurrence, round 6, tools/sweep_panel_u.py at n = 1e5: 6 x 42 runs 12.8 us per step at 7 entries per row and 15.1 at 40 -- ahead of
    // the gather step from ~10.7 entries per row)
    const int min_mean10 = OPT(panel_min_mean10, want_u ? (n <= 105000 ? 110 : 200) : (n <= 105000 ? 170 : 260));
    if (mode < 0 && !(n >= OPT(panel_min_n, 65536) && (shape_only || mean >= 0.1 * min_mean10))) return pp;
    // Shape: NP panels x NB row blocks with NB * NP <= 256 workgroups -- ONE wave of workgroups, one per CU (a second wave
    // doubles the kernel: n = 131 072 with 16 x 18 = 288 workgroups ran 30.5 us per step against 24.9 for the gather step) --,
    // panels of at most 13 x 960 columns (LDS next to the row block's image), row blocks of at most 120 tiles (that image).
    // Preferred panel width ~8 448 columns (measured best at n = 1e5: 12 x 21); wider panels where the row blocks would
    // otherwise not fit.  No such shape beyond n ~ 145 000: the automatic mode then stays with the gather step.
    const int groups = (n + 63) / 64;
    const int cmax = want_u ? kPanUCmax : 13 * kPanWorkThreads, tmax = kPanWork * kPanTW;
    const int cpref = 8448;     // preferred panel width of the record form
    int np = 0, nb = 0;
    const int np_env = OPT(panel_np, 0), nb_env = OPT(panel_nb, 0);
    if (want_u && np_env <= 0 && nb_env <= 0) {
        // Shifted recurrence: the WIDEST panels an instantiation of k_pan_mul8 holds next to the row block's image -- every panel less
        // is n x 8 bytes of partial products neither written nor read back, and the operand costs 8 bytes per column only
        // (n = 1e5: 6 x 42, 13.5 us per step against 15.9 at 12 x 21 and 16.7 for the record form; profiles/r6_panel_u.md)
        for (int c = std::max(1, (n + kPanUCmax - 1) / kPanUCmax); c <= 64 && !np; ++c) {
            const int Cc = (((n + c - 1) / c) + 1) & ~1;
            const int b = std::max(1, std::min(grid_cap(opt) / c, groups));
            const int nt = (groups + b - 1) / b;
            if (nt > tmax) break;                    // (narrower panels leave fewer row blocks: no single-wave shape)
            const int tww = (nt + kPanWork - 1) / kPanWork, twt = tww <= 3 ? 3 : tww <= 5 ? 5 : 8;
            const int lpt = ((Cc + 1) / 2 + kPanWorkThreads - 1) / kPanWorkThreads;
            const int cap = twt == 3 ? pan_u_cols(lpt <= 9 ? lpt : 9, 3) : twt == 5 ? pan_u_cols(lpt <= 8 ? lpt : 8, 5) : pan_u_cols(lpt <= 8 ? lpt : 8, 8);
            if (lpt <= (twt == 3 ? 9 : 8) && Cc <= cap) { np = c; nb = b; }
        }
        if (!np) return plan_panel(opt, n, nnz, maxlen, allowed, nnz_cap, shape_only, false);      // (the record form's shape rules, multi-cell shapes included)
    } else if (np_env > 0 || nb_env > 0) {              // explicit shape (tests, sweeps): taken as given, clipped to what the kernels hold
        np = std::max(1, std::min(np_env > 0 ? np_env : (n + cpref - 1) / cpref, 64));
        if ((n + np - 1) / np > cmax) np = (n + cmax - 1) / cmax;
        nb = nb_env > 0 ? nb_env : std::max(1, grid_cap(opt) / np);
    } else {
        for (int c = std::max(1, (n + cpref - 1) / cpref); c >= 1; --c) {
            if ((n + c - 1) / c > cmax) break;
            const int b = std::max(1, grid_cap(opt) / c);
            if ((groups + b - 1) / b <= tmax) { np = c; nb = b; break; }
        }
        if (!np) {
            // no single-wave shape (n > ~145 000).  Round 4: ONE wave of workgroups all the same, each keeping its panel in LDS and
            // walking `cells` row blocks (k_pan_mul_multi): panels of ~8 448 columns, as many row blocks as the 7 680-row image needs
            const int maxcells = OPT(panel_maxcells, 12);
            for (int c = std::max(1, (n + 8447) / 8448); c >= 1 && c <= grid_cap(opt); --c) {
                if ((n + c - 1) / c > cmax) break;
                const int nbg = std::max(1, grid_cap(opt) / c);
                const int nb_min = (groups + tmax - 1) / tmax;
                const int cl = (nb_min + nbg - 1) / nbg;
                if (cl <= maxcells) { np = c; nb = nbg * cl; pp.cells = cl; break; }
            }
            if (!np) {
                if (mode < 0) return pp;
                np = (n + cmax - 1) / cmax; nb = std::max(1, grid_cap(opt) / np);   // forced: several waves of workgroups
            }
        }
    }
    if (OPT(panel_cells, 0) > 0) pp.cells = OPT(panel_cells, 0);      // (tests: the multi-cell kernel on small graphs)
    if (np > 64) return pp;
    int C = (n + np - 1) / np;
    if (want_u) C = (C + 1) & ~1;                      // (16-byte operand loads: every panel starts on an even column)
    np = (n + C - 1) / C;                              // panels that actually hold columns
    nb = std::max(1, std::min(nb, groups));
    int ntb = (groups + nb - 1) / nb;                            // tiles per row block
    ntb = std::min(ntb, tmax);                                   // (the row block's LDS image holds 7 680 rows)
    nb = (groups + ntb - 1) / ntb;
    pp.cells = std::max(1, std::min(pp.cells, nb));
    // several cells per workgroup (k_pan_mul_multi: the next cell's tile table and chunks are in flight behind the current cell's
    // sums): measured against the gather step (tools/panel_size_probe.py, profiles/r4_panel_sizes.txt) -- per step a tie at ~29
    // entries per row for n = 150 000, ahead from there (n = 200 000: 41.1 vs 43.7 us at 29 / row, 43.8 vs 50.9 at 36; n = 400 000:
    // 119 vs 132 at 34, 129 vs 200 at 49); with the panel build (0.3-0.5 ms per solve) the whole solve wins from ~33 entries per row
    if (mode < 0 && !shape_only && pp.cells > 1 && mean < 0.1 * OPT(panel_multi_min_mean10, 330)) return PanPlan();
    pp.on = true; pp.NP = np; pp.C = C; pp.NB = nb; pp.NTB = ntb; pp.TWW = (ntb + kPanWork - 1) / kPanWork;
    pp.RPT = (C + kPanWorkThreads - 1) / kPanWorkThreads;
    if (want_u) {
        pp.LPT = ((C + 1) / 2 + kPanWorkThreads - 1) / kPanWorkThreads;      // (16-byte loads per worker thread)
        pp.TWT = pp.TWW <= 3 ? 3 : pp.TWW <= 5 ? 5 : 8;
        pp.u = pp.cells == 1 && pp.LPT <= 9 && (pp.LPT < 9 || pp.TWT == 3) &&
               C <= (pp.TWT == 3 ? pan_u_cols(pp.LPT, 3) : pp.TWT == 5 ? pan_u_cols(pp.LPT, 5) : pan_u_cols(pp.LPT, 8));
        if (!pp.u) return plan_panel(opt, n, nnz, maxlen, allowed, nnz_cap, shape_only, false);      // the record form's shape rules
    }
    // MACHIP_PANEL_FUSED=1: the one-launch form (k_pan_step; tickets for 256 row blocks, one partial-sum slot per slice).  Off by
    // default: 26.9 against 19.1 us per step at configs[3] -- the in-launch hand-off costs more than the launch it saves.
#ifdef MACHIP_EXPERIMENTS
    pp.fused = OPT(panel_fused, 0) != 0 && nb <= 256 && nb * np <= 256 && pp.cells == 1 && !pp.u;
#endif
    pp.band = OPT(panel_band, 1) != 0 && (nnz_cap >= 0 ? nnz_cap : nnz) < (1l << 28);     // (CSR positions are packed with 3 count bits; the LOBPCG kernels finish rows without the band terms: solver.h passes band = false there)
    pp.block2 = OPT(panel_b2, 512);
    if (pp.block2 != 256 && pp.block2 != 512 && pp.block2 != 1024) pp.block2 = 256;
    pp.grid2 = (int)std::max<long>(1, std::min<long>(OPT(panel_g2, grid_cap(opt)), ((long)n + pp.block2 - 1) / pp.block2));
    return pp;
}

template <class Op>
inline void launch_spmv(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const double* x, const Op& op) {
    if (pl.variant == kStream) {
        switch (pl.width) {
            case 1: k_spmv_stream<1, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 2: k_spmv_stream<2, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 4: k_spmv_stream<4, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 8: k_spmv_stream<8, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            default: k_spmv_stream<16, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
        }
    } else {
        switch (pl.width) {
            case 2: k_spmv_vec<2, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 4: k_spmv_vec<4, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 8: k_spmv_vec<8, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 16: k_spmv_vec<16, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            case 32: k_spmv_vec<32, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
            default: k_spmv_vec<64, Op><<<pl.grid, kBlock, 0, s>>>(A, x, op); break;
        }
    }
}

// fp32 storage (mixed-precision mode): the shapes plan_pipe actually chooses; anything else maps to the nearest
template <int BLOCK>
inline void launch_pipe_b(const SpmvPlan& pl, hipStream_t s, const CsrViewT<float>& A, const PipeViewT<float>& L, int jrel) {
    const int key = pl.width * 10 + pl.unroll;
    switch (key) {
        case 41: k_pipe_vec<BLOCK, 4, 1, true, float, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 42: case 44: k_pipe_vec<BLOCK, 4, 2, true, float, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 81: case 82: k_pipe_vec<BLOCK, 8, 2, true, float, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 84: k_pipe_vec<BLOCK, 8, 4, true, float, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        default: k_pipe_vec<BLOCK, 16, 2, true, float, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
    }
}

template <int BLOCK>
inline void launch_pipe_b(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const PipeView& L, int jrel) {
    const int key = pl.width * 10 + pl.unroll;
    if (pl.defer >= 3) {   // several row tiles per workgroup: the shapes plan_pipe picks there (others: DEFER = 1 below)
        switch (key) {
            case 41: k_pipe_vec<BLOCK, 4, 1, true, double, 3><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); return;
            case 42: k_pipe_vec<BLOCK, 4, 2, true, double, 3><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); return;
            case 82: k_pipe_vec<BLOCK, 8, 2, true, double, 3><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); return;
            case 84: k_pipe_vec<BLOCK, 8, 4, true, double, 3><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); return;
            case 162: k_pipe_vec<BLOCK, 16, 2, true, double, 3><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); return;
            default: break;
        }
    }
    switch (key) {
        case 41: k_pipe_vec<BLOCK, 4, 1, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 42: k_pipe_vec<BLOCK, 4, 2, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 44: k_pipe_vec<BLOCK, 4, 4, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 84: k_pipe_vec<BLOCK, 8, 4, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 164: k_pipe_vec<BLOCK, 16, 4, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 81: k_pipe_vec<BLOCK, 8, 1, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 82: k_pipe_vec<BLOCK, 8, 2, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 161: k_pipe_vec<BLOCK, 16, 1, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 162: k_pipe_vec<BLOCK, 16, 2, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 321: k_pipe_vec<BLOCK, 32, 1, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 322: k_pipe_vec<BLOCK, 32, 2, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        case 641: k_pipe_vec<BLOCK, 64, 1, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
        default: k_pipe_vec<BLOCK, 64, 2, true, double, 1><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel)); break;
    }
}

// One rank's share of a row-partitioned step: workgroups [PS.first, PS.first + grid) of the pl.grid-workgroup launch.
// The SAME instantiation the unsharded switch picks (DEFER included: the deferred-barrier build is a different piece of
// generated code, and its last bits differ from the plain one's at config 4).
template <int BLOCK>
inline void launch_pipe_shard_b(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const PipeView& L, int jrel, const PeerSet& PS, int grid) {
    const int key = pl.width * 10 + pl.unroll;
    if (pl.defer >= 3) {
        switch (key) {
            case 41: k_pipe_vec<BLOCK, 4, 1, true, double, 3, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); return;
            case 42: k_pipe_vec<BLOCK, 4, 2, true, double, 3, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); return;
            case 82: k_pipe_vec<BLOCK, 8, 2, true, double, 3, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); return;
            case 84: k_pipe_vec<BLOCK, 8, 4, true, double, 3, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); return;
            case 162: k_pipe_vec<BLOCK, 16, 2, true, double, 3, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); return;
            default: break;
        }
    }
    switch (key) {
        case 41: k_pipe_vec<BLOCK, 4, 1, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 42: k_pipe_vec<BLOCK, 4, 2, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 44: k_pipe_vec<BLOCK, 4, 4, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 81: k_pipe_vec<BLOCK, 8, 1, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 82: k_pipe_vec<BLOCK, 8, 2, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 84: k_pipe_vec<BLOCK, 8, 4, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 161: k_pipe_vec<BLOCK, 16, 1, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 162: k_pipe_vec<BLOCK, 16, 2, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 164: k_pipe_vec<BLOCK, 16, 4, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 321: k_pipe_vec<BLOCK, 32, 1, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 322: k_pipe_vec<BLOCK, 32, 2, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        case 641: k_pipe_vec<BLOCK, 64, 1, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
        default: k_pipe_vec<BLOCK, 64, 2, true, double, 1, true><<<grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel), PS); break;
    }
}
// rows per workgroup tile of the instantiation the switches above pick (unlisted shapes run as G = 64)
inline int pipe_gpb(const SpmvPlan& pl) {
    int g = 64;
    switch (pl.width * 10 + pl.unroll) {
        case 41: case 42: case 44: g = 4; break;
        case 81: case 82: case 84: g = 8; break;
        case 161: case 162: case 164: g = 16; break;
        case 321: case 322: g = 32; break;
        default: g = 64; break;
    }
    return (pl.block - 64) / g;
}

// Row-partitioned eigen-solve of an in-process communicator (machip_comm_init_local): the LEADER's
// solver (rank 0) drives every rank's stream -- per Lanczos step one launch per rank (that rank's share of the step's
// workgroups, on that rank's copy of matrix and operand, writing next records and partial sums into every copy),
// ordered by events: step s of rank r waits for step s - 1 of every rank.  Everything else of the solve (host analysis
// of the tridiagonal, explicit residual check, restarts) runs on the leader alone; the converged vector is copied to
// the peers.  The basis V is sharded by rows: each rank keeps the rows its workgroups produced and forms its part of
// the Ritz vector.
struct ShardRank {
    int device = 0;
    hipStream_t stream = nullptr;
    Z2 *Z0 = nullptr, *Z1 = nullptr;
    double *part = nullptr, *V = nullptr, *ypart = nullptr, *sdev = nullptr, *yvec = nullptr;
    CsrView A{};                 // this rank's own assembled copy of L(x)
    hipEvent_t ev[2] = {nullptr, nullptr};
    int g0 = 0, g1 = 0;          // workgroups [g0, g1) of a step (set per solve from the plan)
};
struct ShardGroup {
    std::vector<ShardRank> rk;   // rk[0] = the leader
    hipEvent_t fork = nullptr;
    bool same_device = true;
};
inline void launch_pipe_shard(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const PipeView& L, int jrel, const PeerSet& PS, int grid) {
    if (pl.block == 1024) launch_pipe_shard_b<1024>(pl, s, A, L, jrel, PS, grid);
    else if (pl.block == 512) launch_pipe_shard_b<512>(pl, s, A, L, jrel, PS, grid);
    else launch_pipe_shard_b<256>(pl, s, A, L, jrel, PS, grid);
}

// Padded fixed-width form (kEll): 4 lanes per row, all of a row's W = 8 / 16 slots in flight at once, one row tile per
// workgroup (A.col / A.val are the padded arrays, A.rowptr is not read).
template <int BLOCK>
inline void launch_pipe_ell_b(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const PipeView& L, int jrel) {
    if (pl.unroll == 2) k_pipe_vec<BLOCK, 4, 2, true, double, 1, false, 8><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel));
    else k_pipe_vec<BLOCK, 4, 4, true, double, 1, false, 16><<<pl.grid, BLOCK, 0, s>>>(PIPE_ARGS(A, L, jrel));
}

inline void launch_pipe(const SpmvPlan& pl, hipStream_t s, const CsrView& A, const PipeView& L, int jrel) {
    if (pl.variant == kEll) {
        if (pl.block == 1024) launch_pipe_ell_b<1024>(pl, s, A, L, jrel);
        else if (pl.block == 512) launch_pipe_ell_b<512>(pl, s, A, L, jrel);
        else launch_pipe_ell_b<256>(pl, s, A, L, jrel);
    } else if (pl.variant == kStream) {
        switch (pl.width) {
            case 1: k_pipe_stream<1><<<pl.grid, kBlock, 0, s>>>(A, L, jrel); break;
            case 2: k_pipe_stream<2><<<pl.grid, kBlock, 0, s>>>(A, L, jrel); break;
            case 4: k_pipe_stream<4><<<pl.grid, kBlock, 0, s>>>(A, L, jrel); break;
            case 8: k_pipe_stream<8><<<pl.grid, kBlock, 0, s>>>(A, L, jrel); break;
            default: k_pipe_stream<16><<<pl.grid, kBlock, 0, s>>>(A, L, jrel); break;
        }
    } else if (pl.block == 1024) launch_pipe_b<1024>(pl, s, A, L, jrel);
    else if (pl.block == 512) launch_pipe_b<512>(pl, s, A, L, jrel);
    else launch_pipe_b<256>(pl, s, A, L, jrel);
}
inline void launch_pipe(const SpmvPlan& pl, hipStream_t s, const CsrViewT<float>& A, const PipeViewT<float>& L, int jrel) {
    if (pl.block == 1024) launch_pipe_b<1024>(pl, s, A, L, jrel);      // (the LDS row-tile variant exists in fp64 only)
    else if (pl.block == 512) launch_pipe_b<512>(pl, s, A, L, jrel);
    else launch_pipe_b<256>(pl, s, A, L, jrel);
}

}  // namespace machip
