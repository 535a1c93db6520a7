// panel_u.h -- column-panel Lanczos step with an 8-BYTE operand ("shifted records", round 6).
//
// Why.  The panel step of panel.h is bound by the bytes ONE CU pulls through its vector-memory path (profiles/r3_c4_panel.md:
// ~300 KB per workgroup arrive at ~71 GB/s whatever waits for them).  133 KB of that are the panel's 16-byte records {t, v}: the
// operand v_j = (t - alpha v - mu) / beta_j cannot be stored by the previous step's row kernel because alpha and beta are the
// reductions that kernel only starts.  But the PRODUCT does not need v_j:
//     L v_j = (L t_{j-1} - alpha_{j-1} L v_{j-1} - mu L 1) / beta_j = (L t_{j-1} - alpha_{j-1} w_{j-1}) / beta_j,     w_{j-1} = L v_{j-1}  (kept, 8 B per row)
// so the matrix kernel can multiply a vector that IS known when the previous launch ends -- 8 bytes per column, no coefficients, no
// reduction prologue, no conversion pass in LDS.  Taken literally that recurrence is unstable: the difference d_j = w_j - L v_j obeys
// d_j = -(alpha_{j-1} / beta_j) d_{j-1} + rounding, and alpha / beta ~ 6 on the Erdos-Renyi Laplacians (emulation: 1e16 after 20 steps).
// With the operand SHIFTED by the previous step's alpha,
//     u_{j-1} = t_{j-1} - sigma_{j-1} v_{j-1},   sigma_{j-1} = alpha_{j-2},
//     L v_j = (L u_{j-1} - (alpha_{j-1} - sigma_{j-1}) w_{j-1}) / beta_j,
// the factor becomes |alpha_{j-1} - alpha_{j-2}| / beta_j, which is << 1 once the recurrence has left its first steps (alpha_j settles
// at the centre of the spectrum): tools/experiments/shifted_operand_emulation.py -- same step counts, same lambda_2, drift 1e-16 on every
// configs[1] / configs[3] iterate.  The host watches the accumulated factor (solver.h, `amp`) and restarts a sequence in record form if it
// ever grows (never seen; forced by option panel_u_amp in the tests).
//
// The recurrence in the shifted variables (alpha' = alpha - sigma):  the six sums are MEASURED on (u_j, v_j) exactly as kernels.h
// measures them on (t_j, v_j) -- u.v = alpha', ||u - alpha' v - mu||^2 = beta^2: pipe_coefs applies unchanged -- and
//     v_j = (u_{j-1} - alpha' v_{j-1} - mu) / beta_j                      (pan_vj, the same three roundings)
//     w_j = (sum_p y_p + band . u_{j-1} - alpha' w_{j-1}) / beta_j        y_p = L[., panel p] u_{j-1}  (k_pan_mul8)
//     u_j = w_j - beta_j v_{j-1} - sigma_j v_j,   sigma_j = sigma_{j-1} + alpha'          (= the TRUE alpha_{j-1}: what the host's tridiagonal gets)
// Step 0 (u_{-1} = start vector, v_{-1} = w_{-1} = 0, sigma_0 = 0) normalises the start vector like kernels.h' first step.
// Per row and step k_pan_finu reads NP partials + u, v_{j-1} (the basis column), w + the band, writes u, v_j (basis column), w: 24 bytes
// each way like the record form; k_pan_mul8 loads 8 C bytes of operand instead of 16 C -- which is what makes wide panels (NP = 6 .. 9:
// half the partial-product round trip) affordable.  Reference behaviour preserved: nx:209-213 (mean projection), stop rule nx:232/246.
#pragma once
#include "panel.h"
#ifndef PAN_U_AHEAD
#define PAN_U_AHEAD 2
#endif

namespace machip {

typedef double pan_f2 __attribute__((ext_vector_type(2)));

struct PanU {             // vectors of the shifted recurrence
    double* U0;           // [n + 2] operand u_{j-1}, ping-ponged by step parity (16-byte loads of a panel may touch one double past n)
    double* U1;
    double* W;            // [n] w_{j-1} = L v_{j-1}, updated in place (row-local)
    double* sig;          // [2] sigma by step parity (PipeView::sig)
};

// LDS budget of k_pan_mul8<LPT, TW>: operand panel + row-block image (64 * 15 * TW rows)
__host__ __device__ constexpr int pan_u_rows(int TW) { return 64 * kPanWork * TW; }
__host__ __device__ constexpr int pan_u_cols(int LPT, int TW) {
    return ((2 * kPanWorkThreads * LPT) < ((163840 - 8 * pan_u_rows(TW) - 256) / 8) ? (2 * kPanWorkThreads * LPT) : ((163840 - 8 * pan_u_rows(TW) - 256) / 8)) & ~1;
}

#define PAN_MUL8_ARGS(P, U, L, jrel) (((jrel) & 1) ? (U).U1 : (U).U0), (P).tptr, (P).thead, (P).bval, (P).bcol, (P).n, (P).C, ((P).NP | ((P).TWW << 16)), (((P).NTB << 16) | ((P).rev << 15)), (P), (L), (jrel)
#define PAN_MUL8_ARGS_AT(P, U, L, jrel, b_first) (((jrel) & 1) ? (U).U1 : (U).U0), (P).tptr, (P).thead, (P).bval, (P).bcol, (P).n, (P).C, ((P).NP | ((P).TWW << 16)), ((b_first) | ((P).NTB << 16) | ((P).rev << 15)), (P), (L), (jrel)

// y_p = L[block b, panel p] u  for a plain operand vector: every thread of the workgroup loads its share of the panel with 16-byte
// loads (LPT per thread; C even, so every panel starts on a 16-byte boundary), the 15 worker waves their tiles exactly as k_pan_mul.
// The product needs no coefficients: nothing here waits for anything but its own loads.  The step's reduction prologue still runs in
// THIS launch -- wave 0 of workgroup 0 alone, beside 251 workgroups of loads -- and leaves (alpha', beta, mu, 1 / beta, j, sigma_j) in
// A.coef for the row kernel: inside k_pan_finu the same chain (24 loads per lane, six wave totals, a dozen dependent fp64 operations)
// sat in front of every row's arithmetic (first build of this file: 17.8 against 16.8 us per step at 12 x 21).
// TV: type the tile VALUES are read as -- double, or float for the mixed mode (machip_set_precision(1): `bv32` is the panel form's value
// array rounded to fp32, 6 bytes per entry instead of 10; operand, products and sums stay fp64).
template <int LPT, int TW, typename TV = double>
__global__ __launch_bounds__(kPanThreads) void k_pan_mul8(const double* __restrict__ u_cur, const int* __restrict__ a_tptr,
                                                           const unsigned short* __restrict__ a_thead, const double* __restrict__ a_bval,
                                                           const unsigned short* __restrict__ a_bcol, int a_n, int a_C, int a_npt, int a_bf,
                                                           PanView A_, PipeView L, int jrel, const TV* __restrict__ bv32 = nullptr) {
    // (the first 14 dwords of the arguments arrive in SGPRs with the wave: everything the operand, tile-table and chunk loads need -- a field of
    // the by-value views costs a scalar load of the argument block and a wait in front of the first vector load)
    __shared__ double scoef[8];
    PanView A = A_;
    A.tptr = const_cast<int*>(a_tptr); A.thead = const_cast<unsigned short*>(a_thead); A.n = a_n; A.C = a_C; A.NP = a_npt & 0xffff; A.TWW = a_npt >> 16; A.NTB = a_bf >> 16;
    const int b_first = a_bf & 0x7fff;
    constexpr int SVN = pan_u_cols(LPT, TW), ROWS = pan_u_rows(TW);
    __shared__ __attribute__((aligned(16))) double sv[SVN];
    __shared__ double yblk[ROWS];
    static_assert((SVN + ROWS) * 8 <= 163840 - 128, "panel + row-block image exceed the LDS");
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (b_first: row-partitioned step between processes, solver.h launch_chunk_ipc_pan -- this rank launches the row blocks its rows of the
    // row kernel need, all panels each; cells, sums and their order are those of the whole launch)
    const int bl = blockIdx.x / A.NP, p = blockIdx.x - bl * A.NP, b = bl + b_first;
    const int c0 = p * A.C;
    const int Cp = min(A.C, A.n - c0);
    const int npair = (Cp + 1) >> 1;
    const int R = 64 * A.NTB;
    PAN_CLK(tid == 0, 0); PAN_CLK(tid == 64, 1);
    if (wv == 0) {
        // Wave 0 owns no tiles.  In workgroup 0 it is the step's lead (counters, tridiagonal records, the host's copy, the row kernel's
        // coefficients) -- a role of its own, so that the prologue's 48 partial loads and the workers' chunks never share a register file.
        if (blockIdx.x == 0 && jrel >= 0) {      // (jrel < 0: a plain product y_p = L[., p] x outside the recurrence -- k_pan_rowop finishes it)
            int jd;
            const PipeCoef c = pipe_prologue_wave0(L, jrel, -1, scoef, &jd);
            if (lane == 0) { A.coef[0] = c.alpha; A.coef[1] = c.beta; A.coef[2] = c.mu; A.coef[3] = c.inv; A.coef[4] = (double)jd; A.coef[5] = c.atrue; }
        }
        for (int rl = lane; rl < R; rl += 64) yblk[rl] = 0.0;      // rows of empty tiles
        pan_lds_barrier();
        pan_lds_barrier();
        return;
    }
    const int wt = tid - 64, ww = wv - 1;        // worker thread / worker wave
    // The operand first, with the tile table; it goes into LDS BEFORE the chunk loads leave -- its registers are free for them -- and
    // from there on the only loads in flight are this wave's chunks, ONE straight line of unconditional instructions (a chunk the wave
    // does not have reads the cell's first entry: one line for the whole wave), so that the compiler can count them: chunk c is
    // multiplied when ITS loads have landed (s_waitcnt vmcnt(what was issued behind it)) while the rest of the matrix stream is still in
    // flight, and the barriers order LDS traffic only.  (With wave-uniform branches around the loads every wait was vmcnt(0): the
    // workgroup sat until its last byte had arrived, 5.6 us at configs[3]'s dense iterates, and only then started 4 us of LDS work:
    // tools/pan_clocks.py, profiles/r6_pan_clocks.txt.)
    const pan_f2* __restrict__ Uc = reinterpret_cast<const pan_f2*>(u_cur + c0);
    pan_f2 z[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) z[i] = Uc[min(wt + kPanWorkThreads * i, npair - 1)];
    // this wave's tiles: slot -> row, and the chunk (64 entries) at which each tile ends
    const int vt0 = ((b * A.NP + p) * kPanWork + ww) * A.TWW;
    int ro[TW], cend[TW];
    int E0;
    {
        int tp[TW + 1];
#pragma unroll
        for (int q = 0; q <= TW; ++q) tp[q] = __builtin_amdgcn_readfirstlane(A.tptr[vt0 + (b * A.NP + p) + min(q, A.TWW)]);      // (tile table: NTP + 1 entries per cell)
#pragma unroll
        for (int q = 0; q < TW; ++q) ro[q] = A.thead[(size_t)(vt0 + min(q, A.TWW - 1)) * 64 + lane];
        E0 = tp[0];
#pragma unroll
        for (int q = 0; q < TW; ++q) cend[q] = (tp[q + 1] - E0) >> 6;      // (tiles q >= TWW: same as the last real one)
    }
    const int nch = cend[TW - 1];
    // Odd steps walk the wave's chunks BACKWARDS (option panel_rev): what a step requested last is what the XCD's L2 still holds when the next
    // step starts, and a stream of 5 MB per XCD through a 4 MB L2 in the same order every time never hits -- in alternating order the most
    // recent part does.  cendP / roP: tile ends and slot rows in PROCESSING order; position k of the walk is chunk nch - 1 - k.  (A row's
    // sum then adds its entries in the opposite order on odd steps: equal to rounding, and the same in every run and on every rank.)
    const bool rev = ((a_bf >> 15) & 1) != 0 && jrel >= 0 && (jrel & 1) != 0;
    int cendP[TW], roP[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        const int q = TW - 1 - i;
        cendP[i] = rev ? nch - (q ? cend[q - 1] : 0) : cend[i];
        roP[i] = rev ? ro[q] : ro[i];
    }
    const unsigned voff = (unsigned)(E0 + lane);
    const TV* __restrict__ bv = sizeof(TV) == 8 ? reinterpret_cast<const TV*>(a_bval) : bv32;
    const unsigned short* __restrict__ bc = a_bcol;
    TV pv[kPanCH];
    int pk[kPanCH];
    constexpr int G = 4;        // chunks per group: their loads leave together, their gathers go out together once the loads are in
    constexpr int AHEAD = PAN_U_AHEAD;    // groups of loads in flight ahead of the group being multiplied; the first AHEAD groups leave BEFORE the panel goes into LDS
    static_assert(kPanCH % G == 0, "groups of chunks");
    auto issue = [&](int g0) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int c = g0 + i;
            const unsigned off = c < nch ? voff + (unsigned)((rev ? nch - 1 - c : c) * 64) : (unsigned)E0;      // (wave-uniform choice; no branch)
            pv[c] = bv[off]; pk[c] = bc[off];      // (non-temporal loads here: +10 % per step; non-temporal stores of the partials +6 %, of the row kernel +8 %: profiles/r6_panel_u.md)
        }
    };
#pragma unroll
    for (int g0 = 0; g0 < G * AHEAD && g0 < kPanCH; g0 += G) issue(g0);
    {
        pan_f2* __restrict__ sv2 = reinterpret_cast<pan_f2*>(sv);
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = wt + kPanWorkThreads * i;
            if (c < npair) sv2[c] = z[i];
        }
    }
#ifdef PAN_CLOCKS
    if (wv == 1) PAN_CLK(tid == 64, 2);          // operand in LDS, tile table known (wave 1)
#endif
    pan_lds_barrier();          // (the panel is in LDS; nothing but LDS traffic is awaited here -- and no chunk load has been issued yet: a wave cannot
                                //  pass a barrier behind loads the memory pipeline has not ACCEPTED, and 600 of them per CU are not accepted before ~85 % of
                                //  the matrix stream has landed: first build of this kernel, barrier passed at 5.9 us)
    PAN_CLK(tid == 64, 5);
    double acc = 0.0;
    {
        // No contraction in this block: a product and its addition to the row's sum stay two roundings for EVERY chunk (see k_pan_mul).
#pragma clang fp contract(off)
        unsigned endmask = 0;      // chunks of the first round that close a (non-empty) tile
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            const int prev = q ? cendP[q - 1] : 0, last = cendP[q] - 1;
            if (cendP[q] > prev && last >= 0 && last < kPanCH) endmask |= 1u << last;      // (tiles beyond TWW are empty: they end where their predecessor does)
        }
#pragma unroll
        for (int g0 = 0; g0 < kPanCH; g0 += G) {
            if (g0 + G * AHEAD < kPanCH) issue(g0 + G * AHEAD);
            double x[G];
#pragma unroll
            for (int i = 0; i < G; ++i) x[i] = sv[min(pk[g0 + i], Cp - 1)];
#pragma unroll
            for (int i = 0; i < G; ++i) x[i] *= (double)pv[g0 + i];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int c = g0 + i;
                if (c < nch) {
                    acc += x[i];
                    if (endmask & (1u << c)) {       // its lanes' rows are complete
#pragma unroll
                        for (int q = 0; q < TW; ++q)
                            if (c + 1 == cendP[q] && (q == 0 ? cendP[0] > 0 : cendP[q] > cendP[q - 1])) yblk[roP[q]] = acc;
                        acc = 0.0;
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // (the next group's gathers must not be hoisted above this one's: they would wait for later loads)
            PAN_CLK(tid == 64 && g0 == 0, 6);
        }
        for (int cb = kPanCH; cb < nch; cb += kPanCH) {     // more chunks than the registers hold: further rounds (dense rows; one more round trip each)
#pragma unroll
            for (int c = 0; c < kPanCH; ++c)
                if (cb + c < nch) { const unsigned o2 = voff + (unsigned)((rev ? nch - 1 - (cb + c) : cb + c) * 64); pv[c] = bv[o2]; pk[c] = bc[o2]; }
            double xr[kPanCH];
#pragma unroll
            for (int c = 0; c < kPanCH; ++c) xr[c] = (double)pv[c] * sv[min(pk[c], Cp - 1)];
            unsigned em = 0;
#pragma unroll
            for (int q = 0; q < TW; ++q) {
                const int prev = q ? cendP[q - 1] : 0, last = cendP[q] - 1 - cb;
                if (cendP[q] > prev && last >= 0 && last < kPanCH) em |= 1u << last;
            }
#pragma unroll
            for (int c = 0; c < kPanCH; ++c) {
                if (cb + c < nch) {
                    acc += xr[c];
                    if (em & (1u << c)) {
#pragma unroll
                        for (int q = 0; q < TW; ++q)
                            if (cb + c + 1 == cendP[q] && (q == 0 ? cendP[0] > 0 : cendP[q] > cendP[q - 1])) yblk[roP[q]] = acc;
                        acc = 0.0;
                    }
                }
            }
        }
    }
    PAN_CLK(tid == 64, 7);
    pan_lds_barrier();
    // the row block's sums, un-sorted by the LDS image: coalesced stores
    for (int rl = wt; rl < R; rl += kPanWorkThreads) {
        const int row = b * R + rl;
        if (row < A.n) A.ypart[(size_t)p * A.n + row] = yblk[rl];
    }
    PAN_CLK(tid == 64, 8); PAN_CLK(tid == 1023, 9);
}

#define PAN_FINU_ARGS(P, U, L, jrel, jhost) (((jrel) & 1) ? (U).U1 : (U).U0), (P).ypart, (U).W, (L).V, ((P).band ? (P).bd : nullptr), (P).coef, (P).n, \
    ((P).NP | (((jrel) & 0xff) << 7) | (((jhost) + 1) << 15)), (((jrel) & 1) ? (U).U0 : (U).U1), (L).part

// Row kernel of the shifted recurrence: the coefficients come from k_pan_mul8's workgroup 0 through the coefficient block (as k_pan_fin gets them).
// Argument order (round 6, late): everything the ROW LOADS need -- operand, partial products, w, the basis, the band, n, NP and both step indices (one packed dword) -- and the
// coefficient block's address sit in the first 14 dwords, which arrive in SGPRs with the wave (build.sh: kernarg preload), and the six coefficients are requested
// BEHIND the row loads as one more (uniform) vector load each.  Before, the kernel waited for the rest of its argument block, then for the
// coefficient block (scalar loads: one counter, so the wait for an address also waited for them), and only then asked for its rows: three
// round trips in a row where one is needed (ISA of round 6's first build; profiles/r6_panel_u.md section 6).
// jhost >= 0: the step index, known to the host on eager launches -- the address of the basis column of v_{j-1} then does not wait
// for the coefficient block.
// NPM: panels whose partials are requested in one batch (>= NP wherever a shape is dispatched on it; clamped loads beyond NP would
// only re-read the last plane).
// SH (compile time): this launch is one rank's share of a row-partitioned step; its PeerSet sits in device memory (PSd), its first workgroup and
// the launch's total are plain arguments.  (First builds of the partitioned form: the set as a by-value argument whose address was taken at run
// time went to scratch memory, 13.5 -> 18.2 us per step; as a by-value argument never touched it still cost the un-partitioned row kernel 0.46 us.)
template <int BLOCK, int NPM, bool SH = false>
__global__ __launch_bounds__(BLOCK) void k_pan_finu(const double* __restrict__ u_cur, const double* __restrict__ a_ypart, double* __restrict__ wvec,
                                                     double* __restrict__ a_V, const double* __restrict__ a_bd, const double* a_coef, int n, int a_pk,
                                                     double* __restrict__ u_nxt, double* l_part,
                                                     const PeerSet* __restrict__ PSd = nullptr, int sh_first = 0, int sh_total = 0
#ifdef PAN_CLOCKS
                                                     , long long* clk = nullptr
#endif
                                                     ) {
#ifdef PAN_CLOCKS
#define FINU_CLK(i) do { if (clk && threadIdx.x == 0) clk[blockIdx.x * 16 + (i)] = wall_clock64(); } while (0)
#else
#define FINU_CLK(i) do { } while (0)
#endif
    FINU_CLK(10);
    // SH: this rank's share [sh_first, sh_first + gridDim.x) of a sh_total-workgroup launch (row-partitioned step between processes):
    // same rows per workgroup, same partial-sum slots; the next operand's rows and the six sums go into EVERY rank's copy (the operand
    // buffers of such a sequence live in the record buffers Z0 / Z1, which the peers have mapped), v_j and w stay with the owner.
    __shared__ double smw[kNP * BLOCK];
    const int NP = a_pk & 0x7f, jrel = (a_pk >> 7) & 0xff, jhost = (a_pk >> 15) - 1;      // (one dword: the 14 preloaded ones are all taken)
    const int bid = SH ? sh_first + (int)blockIdx.x : (int)blockIdx.x;
    const int gtot = SH ? sh_total : (int)gridDim.x;
    const int par = jrel & 1;
    const int j = jhost >= 0 ? jhost : (int)a_coef[4];
    const double* __restrict__ vprev = a_V + (size_t)max(j - 1, 0) * (size_t)n;
    double* __restrict__ vj = a_V + (size_t)j * (size_t)n;
    const bool first = j == 0;
    PipeRow pr;
    pr.clear();
    int r = bid * BLOCK + threadIdx.x;
    // the first batch of row loads leaves before anything else is asked for
    double y[NPM], up = 0.0, ul = 0.0, uu = 0.0, vp = 0.0, wp = 0.0, b0 = 0.0, bl = 0.0, bu = 0.0;
    auto request = [&](int rr) {
#pragma unroll
        for (int q = 0; q < NPM; ++q) y[q] = __builtin_nontemporal_load(a_ypart + (size_t)min(q, NP - 1) * n + rr);      // (read once: must not displace the tiles the next step finds in L2)
        up = u_cur[rr]; ul = u_cur[max(rr - 1, 0)]; uu = u_cur[min(rr + 1, n - 1)];
        vp = vprev[rr]; wp = wvec[rr];
        if (a_bd) { b0 = a_bd[rr]; bl = a_bd[(size_t)n + rr]; bu = a_bd[2 * (size_t)n + rr]; }
    };
    if (r < n) request(r);
    __builtin_amdgcn_sched_barrier(0);
    // the coefficients: uniform vector loads BEHIND the rows' (a scalar load would share its counter with the argument block's)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));        // (0 in a vector register the compiler knows nothing about: keeps these loads on the vector-memory path)
    const pan_f2* __restrict__ cf = reinterpret_cast<const pan_f2*>(a_coef) + zero;
    const pan_f2 c01 = cf[0], c23 = cf[1], c45 = cf[2];
    const double alpha = c01.x, beta = c01.y, mu = c23.x, inv = c23.y, sigma = c45.y;
#ifdef PAN_CLOCKS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FINU_CLK(11);
#endif
    for (; r < n; ) {
        if (first) { vp = 0.0; wp = 0.0; }
        double q = 0.0;
#pragma unroll
        for (int k = 0; k < NPM; ++k) q += (k < NP) ? y[k] : 0.0;
        for (int p0 = NPM; p0 < NP; p0 += NPM) {     // (more panels than one batch: tests)
#pragma unroll
            for (int k = 0; k < NPM; ++k) y[k] = a_ypart[(size_t)min(p0 + k, NP - 1) * n + r];
#pragma unroll
            for (int k = 0; k < NPM; ++k) q += (p0 + k < NP) ? y[k] : 0.0;
        }
        {
#pragma clang fp contract(off)
            if (a_bd) {       // the tridiagonal band, kept out of the panel form
                q = __builtin_fma(b0, up, q);
                q = __builtin_fma(bl, ul, q);
                q = __builtin_fma(bu, uu, q);
            }
            const double v = pan_vj(alpha, mu, inv, up, vp);
            const double w = __builtin_fma(-alpha, wp, q) * inv;
            const double u = __builtin_fma(-sigma, v, __builtin_fma(-beta, vp, w));
            vj[r] = v; wvec[r] = w;      // (a non-temporal store of the basis column: +1 % per step; of the partial products in k_pan_mul8: a tie)
            if (SH) { for (int q2 = 0; q2 < PSd->n; ++q2) peer_store(reinterpret_cast<double*>(par ? PSd->Z0[q2] : PSd->Z1[q2]) + r, u); }      // (write-through: kernels.h peer_store)
            else u_nxt[r] = u;
            pr.acc[0] = __builtin_fma(u, u, pr.acc[0]); pr.acc[1] = __builtin_fma(u, v, pr.acc[1]); pr.acc[2] = __builtin_fma(v, v, pr.acc[2]);
            pr.acc[3] += u; pr.acc[4] += v; pr.acc[5] += fabs(v);
        }
        r += gtot * BLOCK;
        if (r < n) request(r);
    }
    FINU_CLK(12);
    struct { double* part; } Lp{l_part};          // (the six sums' slots are all the row kernel needs of the recurrence's view: 200 bytes of arguments less)
    pr.template store<BLOCK>(Lp, jrel, smw, SH ? PSd : nullptr);
    if (SH) peer_drain();
#ifdef PAN_CLOCKS
    FINU_CLK(13);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FINU_CLK(14);
#endif
}

// Row kernel of a PLAIN product in panel form (round 6): (L x)[r] = sum_p y_p[r] + band . x, handed to the same `Op` objects the CSR
// products take (kernels.h: begin / row / end) -- the landscape sweeps of the cold start (OpLand) and the product of the explicit residual
// check (OpLanczos) run on the panel form while a solve has it (k_pan_mul8 with jrel < 0 + this kernel: 9.4 + ~4 us against 27.5 / 35 us for
// the gathering CSR kernels at configs[3]).  Row sums are added in panel order, not CSR order: equal to rounding.
template <class Op>
__global__ __launch_bounds__(kBlock) void k_pan_rowop(PanView A, const double* __restrict__ x, Op op) {
    __shared__ double sm[4];
    op.begin(sm);
    const int n = A.n, NP = A.NP;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < n; r += gridDim.x * kBlock) {
        double q = 0.0;
        for (int p0 = 0; p0 < NP; p0 += 8) {
            double y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = A.ypart[(size_t)min(p0 + k, NP - 1) * n + r];
#pragma unroll
            for (int k = 0; k < 8; ++k) q += (p0 + k < NP) ? y[k] : 0.0;
        }
        if (A.band) {
            q = __builtin_fma(A.bd[r], x[r], q);
            q = __builtin_fma(A.bd[(size_t)n + r], x[max(r - 1, 0)], q);
            q = __builtin_fma(A.bd[2 * (size_t)n + r], x[min(r + 1, n - 1)], q);
        }
        op.row(r, q);
    }
    op.end(sm);
}

// Start a sequence of the shifted recurrence from u0: U0 = u0, sigma = 0; partials such that step 0 normalises u0 (k_pipe_init's rule).
__global__ __launch_bounds__(kBlock) void k_pipe_init_u(PipeView L, PanU U, const double* __restrict__ u0, int epoch) {
    __shared__ double sm[4];
    double s1 = 0.0, s2 = 0.0;
    for (int r = blockIdx.x * kBlock + threadIdx.x; r < L.n; r += gridDim.x * kBlock) {
        const double t = u0[r];
        U.U0[r] = t;
        s1 += t; s2 += t * t;
    }
    s1 = block_sum(s1, sm); s2 = block_sum(s2, sm);
    if (threadIdx.x == 0) {
        for (int q = 0; q < kNP; ++q) L.part[q * kMaxGrid + blockIdx.x] = 0.0;
        L.part[0 * kMaxGrid + blockIdx.x] = s2;
        L.part[3 * kMaxGrid + blockIdx.x] = s1;
        if (blockIdx.x == 0) {
            L.st->jA = 0; L.st->jN = 0; L.st->epoch = epoch;
            U.sig[0] = 0.0; U.sig[1] = 0.0;
            U.U0[L.n] = 0.0; U.U0[L.n + 1] = 0.0; U.U1[L.n] = 0.0; U.U1[L.n + 1] = 0.0;      // (the 16-byte loads' overhang)
        }
    }
}

}  // namespace machip
