// solver.h -- device-resident Lanczos solver for the Fiedler pair of a CSR Laplacian.
//
// Algorithm (replaces nx:151-256 TraceMIN + SuperLU, which the reference calls at
// mac/utils/fiedler.py:42): plain Lanczos on L restricted to 1-perp (the mean is projected out
// of every new vector, as nx:209-213 does), no re-orthogonalisation, the whole basis kept in HBM
// (288 GB makes that free) so the Ritz vector is one tall-skinny pass at the end.  One kernel
// launch per Lanczos step (kernels.h, "pipelined" form); chunks of steps are replayed from cached
// hipGraphs and run one chunk AHEAD of the host, which meanwhile analyses the previous chunk's
// O(J) scalars (tridiag.h).  Convergence is declared by the reference's own rule evaluated
// explicitly on the device:   || L v - lambda v ||_1 / || L ||_inf < tol     (nx:232, nx:246)
//
// This file is the DRIVER (allocation, chunk graphs, host analysis, mode selection, restarts); launch shapes and the
// dispatch onto kernel instantiations are in plan.h, kernels in kernels.h / panel.h / persist.h / precond.h / woodbury.h.
#pragma once
#include <functional>
#include <dlfcn.h>

#include <array>
#include <deque>
#include <limits>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "kernels.h"
#include "persist.h"
#include "precond.h"
#include "woodbury.h"
#include "panel.h"
#include "plan.h"
#include "tridiag.h"
#include "follow.h"
#ifdef MACHIP_EXPERIMENTS
#include "band.h"
#include "blocklan.h"
#endif

namespace machip {

// (Rounds 2-3 serialised stream captures against rocBLAS / rocSOLVER calls of other threads behind a process-wide lock: those
// libraries touch the legacy stream, which HIP refuses while another thread captures.  Round 4 removed the libraries -- the dense
// inverse is hand-written, woodbury.h -- and the lock with them: captures are thread-local, every other call of this library
// names its own non-blocking stream.)

// Communicator between PROCESSES whose eigen-solve is row-partitioned (machip_comm_init_ipc; kernels.h IpcView): this rank's
// share of every Lanczos step, the peers' buffers mapped through hipIpcOpenMemHandle, ordering by flag words in device memory.
struct IpcGroup {
    int nranks = 0, rank = 0;
    IpcView view;                        // flags / counters (kernel argument)
    void* Z0[kMaxPeers] = {};            // every rank's record buffers, partial sums, Ritz-vector staging, gradient (index = rank;
    void* Z1[kMaxPeers] = {};            // my own entries are my own pointers)
    double* part[kMaxPeers] = {};
    double* yraw[kMaxPeers] = {};
    double* g[kMaxPeers] = {};
    std::vector<void*> opened;           // what hipIpcCloseMemHandle must release
    unsigned long long* flags_mem = nullptr;   // my flag words + counters (device, exported to the peers)
    int* h_err = nullptr;                // mapped pinned
    int g0 = 0, g1 = 0;                  // my workgroups of the running sequence's step launch
    bool clean_exit = false;             // the owner said goodbye (machip_comm_close_ipc): destroying the handle raises no abort
    // (every error path of machip_ipc_export / machip_comm_init_ipc ends here: mappings closed, flag words and the error word freed)
    ~IpcGroup() {
        for (void* q : opened) (void)hipIpcCloseMemHandle(q);
        if (flags_mem) (void)hipFree(flags_mem);
        if (h_err) (void)hipHostFree(h_err);
    }
};

struct Solver {
    Options opt = default_options();    // the handle's option table (machip_set_option); a copy of the process defaults at creation
    int n = 0;
    hipStream_t stream = nullptr;
    size_t vcap = 0;            // Lanczos vectors that fit in V
    // Krylov state
    double *u = nullptr, *V = nullptr, *tri = nullptr, *part = nullptr;
    Z2 *Z0 = nullptr, *Z1 = nullptr;
    LanState* st = nullptr;
    // explicit-check / result state (OpLanczos "column 0" machinery)
    double *y_raw = nullptr, *w2 = nullptr, *yvec = nullptr, *ypart = nullptr, *sdev = nullptr;
    double *part_c = nullptr, *part_a2 = nullptr, *part_r = nullptr, *scratch3 = nullptr, *rq_dev = nullptr;
    LanState* st2 = nullptr;
    // classic (two-kernel) Lanczos state: the accurate fallback for tiny / nearly exhausted Krylov spaces
    double *wc = nullptr, *ctri = nullptr, *part_u = nullptr, *part_a = nullptr;
    LanState* stc = nullptr;
    float* valf = nullptr;      // fp32 copy of the matrix values (mixed-precision mode)
    size_t valf_cap = 0, csr_cap = 0;   // csr_cap: capacity of the caller's CSR buffers (0 = unknown: grow on demand)
    bool last_seq_f32 = false;  // the basis V of the last sequence holds floats (ritz_block must not read it as fp64)
    long last_steps_lowp = 0;   // fp32 steps of the last Lanczos solve
    double* start = nullptr;    // persistent cold-start vector
    bool have_start = false, have_prev = false;
    int ks_max = 16;
    // pinned host staging
    double* h_tri = nullptr;    // mirror of tri (pinned, device-mapped: the tail kernel writes it)
    double* d_htri = nullptr;   // device view of h_tri
    double* d_hpin = nullptr;   // device view of h_pin (kernels write the host's small results there themselves)
    unsigned long long* h_flag = nullptr;   // pinned completion flag polled by the host
    unsigned long long* d_hflag = nullptr;
    unsigned int epoch = 0;
    double* h_pin = nullptr;    // misc
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evs0 = nullptr, evs1 = nullptr;   // evs*: bracket the Krylov chunks only
    bool ev1_at_check = false;  // ev1 was recorded behind the last explicit check's kernels (and that check has been waited for)
    // Speculative epilogue: work the caller wants behind a PASSING explicit check (machip_fw_step: gradient, top-K, Frank-Wolfe
    // bookkeeping) is enqueued behind EVERY check's kernels, before the host waits for the check -- if the check passes, its
    // results are there at the same wait (one host round trip less per iteration); if not, it runs again behind the next check.
    std::function<void()> after_check;
    long check_seq = 0, hook_seq = -1, final_check_seq = -2;    // hook results are valid iff hook_seq == final_check_seq
    bool spec_likely = true;    // the check about to run is expected to pass (its residual estimate is below the tolerance): only then is the
                                // epilogue worth enqueueing behind it -- behind a failing check its kernels only delay the next chunk
    std::vector<hipEvent_t> ev_pool;
    // cached chunk graphs: (variant, width, grid, steps) -> exec
    std::map<std::tuple<int, int, int, int, int>, std::array<hipGraphExec_t, 2>> graphs;
    unsigned graph_flip = 0;
    const void* graph_csr_key = nullptr;
    bool profiled = false;      // a rocprofiler-sdk is attached to the process (graphs are then off unless option "graph" = 1)
    // Captured chunks or eager launches (option `graph`: 1 / 0; unset = automatic).  Rounds 1-4 captured every chunk: the host had to
    // get a whole chunk out at each end-game hop.  With streamed records (round 5) the queue is fed continuously, and eager launches
    // -- no submission seam per graph -- are faster wherever a launch runs longer than the host needs to issue one (~4.4 us):
    // configs[3] 241.5 -> 246.0 it/s, configs[1] 706 -> 721, two-lane sweeps c4s 265 -> 293, c2s 923 -> 1 178; city10000's 4.2 us
    // steps lose 3.7 % and keep their graphs (tools/r5_eager.sh).  launch_us: measured in-solve time per launch of this handle's last
    // solve in the same step form, the model before there is one; either way the results are the same bits.
    double launch_us_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // by step form (SpmvPlan::variant)
    double cur_launch_us = 0.0;                           // what the running solve's fused chunks go by
    bool use_graph(double launch_us = 0.0) const {
        if (opt.is_set(kOpt_graph)) return OPT(graph, 1) != 0;
        if (profiled) return false;
        // (evaluation lanes never capture: a lane's hipGraph executables cost the other lanes their hardware queues -- configs[1] as a
        // 4-lane sweep 1 185 -> 635 it/s after one captured solve per lane -- and city10000's lanes run as fast eagerly, 1 216 vs 1 224)
        if (throughput_lane) return false;
        return launch_us > 0.0 && launch_us < 5.2;
    }
    // host copies of T
    std::vector<double> ha, hb, hl1;
    std::vector<double> wk, guess;
    tri::Smallest sm;
    int J_last = 0;             // dimension of the Krylov space behind the current yvec
    long last_steps = 0;        // steps the previous solve needed (chunk sizing hint)
    int maxlen_hint = 0;        // longest row of the matrix about to be solved (0 = unknown)
    // preconditioned mode (precond.h): allocated on first use
    double *lx_x = nullptr, *lx_Lx = nullptr, *lx_p = nullptr, *lx_Lp = nullptr, *lx_Lw = nullptr;
    double *lx_rT = nullptr, *lx_wT = nullptr, *lx_tl = nullptr, *lx_tdinv = nullptr, *lx_tcu = nullptr;
    double *lx_ys = nullptr, *lx_pas = nullptr, *lx_as = nullptr, *lx_bs = nullptr;   // big-n solver scratch
    double* lx_maps = nullptr;   // 4 x stride chunk maps of the multi-workgroup tridiagonal solve
    double *lx_ba = nullptr, *lx_bd = nullptr, *lx_bu = nullptr;   // tridiagonal band of L in the solver's layout
    // exact chain + closures preconditioner (woodbury.h)
    int *wb_ui = nullptr, *wb_uj = nullptr, *wb_counts = nullptr;
    double *wb_uc = nullptr, *wb_g = nullptr, *wb_h = nullptr, *wb_Zt = nullptr, *wb_Cm = nullptr;
    size_t wb_Zt_cap = 0, wb_pas_cap = 0;
    double *wb_pas = nullptr, *wb_maps = nullptr;   // batched multi-workgroup column solves (n > 16 384)
    double* wb_Cm2 = nullptr;    // second buffer of the blocked Gauss-Jordan inversion (ping-pong)
    double* wb_piv = nullptr;    // 2 x (32 x 32): look-ahead inverse of the next pivot block (k_gj_step)
    WbView wb_active{};          // s > 0 while the running solve uses it
    double *lx_part = nullptr, *lx_partR = nullptr;
    int *lx_colT = nullptr, *lx_bad = nullptr;
    PersistPack ppack{};         // packed registers / LDS image of the single-workgroup kernel (persist.h), per matrix
    int* ell_col = nullptr;      // padded fixed-width copy of a short-row matrix (k_ell_build; 16 slots per row at most)
    double* ell_val = nullptr;
    CsrView ell_view{};          // {n, nullptr, ell_col, ell_val} while the current solve steps on it
    size_t lx_colT_cap = 0;
    LobState* lx_st = nullptr;
    double *h_lrec = nullptr, *d_hlrec = nullptr;
    bool lob_ready = false, last_was_lob = false;
    // what the last solve's steps were (machip_solve_mode; bench.py prices the roofline of THAT launch group):
    // 1 fused gather step (k_pipe_vec on the CSR), 2 column-panel step, 3 padded fixed-width step, 4 single-workgroup kernel,
    // 5 classic two-kernel step, 6 preconditioned by the tridiagonal chain, 7 exact chain + closures mode, 8 fp32 + fp64 sequences
    int last_mode = 0;
    long last_wb_s = 0;          // closures of the exact mode's capacitance matrix in that solve
    int solver_mode = 0;        // 0 = auto, 1 = Lanczos, 2 = preconditioned (LOBPCG + tridiagonal solve)
    bool throughput_lane = false;   // this solver serves an evaluation lane (machip_eval_batch / machip_fw_sweep): many solves run at
                                    // once, so the automatic mode keeps to kernels that occupy ONE CU per solve where it can
    int precision = 0;          // 0 = fp64; 1 = fp32 Krylov iterate + fp64 refinement (machip_set_precision)
    bool chain_like = false;    // the fixed edges contain (nearly) the whole chain (i, i+1)
    long chain_edges = 0;       // how many of them
    long support_hint = -1;     // active candidate edges of the matrix about to be solved (-1 = unknown)
    long hist_lan_steps = -1, hist_lob_iters = -1;   // steps / iterations of the last solve in each mode
    long hist_exact_iters = -1;                       // ... of the last solve the exact chain + closures preconditioner served
    // Lanczos -> exact hand-over (solve(): chain-like graphs beyond the single-workgroup sizes with at most wb_soft() closures): the
    // Lanczos loop gives up when its own forecast of the remaining steps costs more than 1.3x the exact mode's estimate, twice in a row
    static constexpr int kSwitchToExact = 1000;       // internal status of solve_lanczos, never leaves solve()
    double switch_est_us = 0.0;                       // > 0: the hand-over is allowed, the exact mode's estimated cost
    double switch_to_go = 0.0;                        // forecast at the hand-over (history for the next solve)
    static constexpr int kLobCap = 100000;
    // column-panel step (panel.h): the panel form of the matrix being solved, rebuilt per solve from its CSR
    bool seq_sharded = false;      // the running Krylov sequence is row-partitioned (its basis is spread over the ranks)
    bool seq_ipc = false;          // ... between processes (IpcGroup): every rank runs this host loop itself
    SpmvPlan seq_plan;             // ... with this launch shape
    ShardGroup* shard = nullptr;   // set (by the leader's handle) for the duration of a row-partitioned solve
    IpcGroup* ipc = nullptr;       // set while this handle belongs to an inter-process communicator with a row-partitioned eigen-solve
    bool last_seq_sharded = false; // the basis of the last sequence is spread over the ranks (ritz_block cannot read it)
    bool pan_allowed = false;   // the CSR is one this library assembled (diagonal first, other columns ascending)
    PanPlan pan;
    PanView panv{};
    bool pan_live = false;         // the panel form (shifted-recurrence shape) of the matrix being solved is built: plain products may use it (panel_spmv)
    bool pan_u = false;            // the running sequence's panel steps are those of the shifted recurrence (panel_u.h): 8-byte operand
    bool pan_u_off = false;        // ... ruled out for the rest of this solve (the drift monitor tripped)
    PanU pu{};
    PeerSet* d_ps = nullptr; PeerSet h_ps{};      // the row-partitioned panel step's peer set, in device memory
    double *pu_sig = nullptr, *pu_U0 = nullptr, *pu_U1 = nullptr;     // (pu.U0 / U1 point at these, or at the record buffers under an inter-process communicator)
    double last_amp = 0.0;         // largest accumulated drift factor of the last solve's shifted sequences (solve stats / tests)
    // mixed mode of the panel step (machip_set_precision(1), round 6): the LATE steps of a sequence read the tile values rounded to fp32
    bool pan32 = false;            // this solve may switch (shifted recurrence, eager launches, an instantiated shape)
    int pan32_from = INT_MAX;      // ... from this step of the running sequence on (decided by the host from the records; INT_MAX: not yet)
    float* pan_bv32 = nullptr;     // the panel form's value array rounded to fp32 (same layout as panv.bval)
    size_t pan_bv32_cap = 0;
    size_t pan_cap = 0, pan_nt_cap = 0, pan_y_cap = 0, pan_band_cap = 0;
    std::array<int, 5> pan_shape_key{0, 0, 0, 0, 0};     // shape the cells' static ranges (panv.cbase) were computed for
    PatternView pat{};          // union pattern of the handle (machip_create); none on a CSR-only handle
    bool pan_rows_ready = false;     // the last assembly wrote the per-row panel tables (PanSpec) ...
    int pan_rows_NP = 0, pan_rows_C = 0; bool pan_rows_band = false;     // ... for this shape

    // budget_mb > 0: HBM budget of the Krylov basis V for this instance (evaluation lanes take a share each)
    int init(int n_, hipStream_t s, int budget_mb = 0) {
        n = n_;
        stream = s;
        size_t budget = (size_t)(budget_mb > 0 ? budget_mb : OPT(vbudget_mb, 4096)) * (size_t)(1 << 20);
        vcap = budget / (sizeof(double) * (size_t)std::max(n, 1));
        vcap = std::max<size_t>(std::min<size_t>(vcap, 16384), 64);
        vcap = std::min<size_t>(vcap, (size_t)n + 10);   // a Krylov sequence never exceeds n - 1 + 8 columns (jcap)
        vcap = std::max<size_t>(vcap, 64);
        vcap = (size_t)OPT(vcap, (int)vcap);
        // rocprofiler-sdk (ROCm 7.2) segfaults inside its HSA interception when short graphs are
        // launched in quick succession (reproduced under rocprofv3 --kernel-trace on the pose-graph
        // tests; eager launches of the same kernels are fine): with a profiler attached fall back to
        // eager launches unless MACHIP_GRAPH says otherwise.
        profiled = dlopen("librocprofiler-sdk.so", RTLD_NOLOAD | RTLD_LAZY) != nullptr ||
                   dlopen("librocprofiler-sdk.so.1", RTLD_NOLOAD | RTLD_LAZY) != nullptr;
        ST_TRY(dev_alloc(&u, n));
        ST_TRY(dev_alloc(&V, (size_t)n * vcap));
        ST_TRY(dev_alloc(&tri, 3 * (vcap + 2)));
        ST_TRY(dev_alloc(&part, 2 * kNP * kMaxGrid));
        ST_TRY(dev_alloc(&Z0, n)); ST_TRY(dev_alloc(&Z1, n));
        ST_TRY(dev_alloc(&st, 1)); ST_TRY(dev_alloc(&st2, 1));
        HIP_TRY(hipMemsetAsync(st2, 0, sizeof(LanState), stream));      // (the explicit check's "column 0" counters stay {0, 0}: nothing advances them)
        ST_TRY(dev_alloc(&y_raw, (size_t)n + 2)); ST_TRY(dev_alloc(&w2, (size_t)n + 2)); ST_TRY(dev_alloc(&yvec, n));      // (+2: a panel's 16-byte operand loads may touch one double past n)
        ST_TRY(dev_alloc(&ypart, (size_t)n * ks_max)); ST_TRY(dev_alloc(&sdev, vcap + 2));
        ST_TRY(dev_alloc(&part_c, 3 * kMaxGrid)); ST_TRY(dev_alloc(&part_a2, kMaxGrid));
        ST_TRY(dev_alloc(&part_r, kMaxGrid)); ST_TRY(dev_alloc(&scratch3, 8)); ST_TRY(dev_alloc(&rq_dev, 1));
        ST_TRY(dev_alloc(&start, n));
        ST_TRY(dev_alloc(&wc, n)); ST_TRY(dev_alloc(&ctri, 3 * (vcap + 2)));
        ST_TRY(dev_alloc(&part_u, 3 * kMaxGrid)); ST_TRY(dev_alloc(&part_a, kMaxGrid)); ST_TRY(dev_alloc(&stc, 1));
        HIP_TRY(hipHostMalloc((void**)&h_tri, 3 * (vcap + 2) * sizeof(double), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&d_htri, h_tri, 0));
        HIP_TRY(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&d_hflag, h_flag, 0));
        *h_flag = 0;
        HIP_TRY(hipHostMalloc((void**)&h_pin, (4 * (vcap + 2) + 2 * kMaxGrid + 64) * sizeof(double), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&d_hpin, h_pin, 0));
        HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1));
        HIP_TRY(hipEventCreate(&evs0)); HIP_TRY(hipEventCreate(&evs1));
        return MACHIP_OK;
    }
    void destroy() {
        for (auto& kv : graphs) for (hipGraphExec_t ge : kv.second) if (ge) (void)hipGraphExecDestroy(ge);
        graphs.clear();
        void* ptrs[] = {u, V, tri, part, Z0, Z1, st, st2, y_raw, w2, yvec, ypart, sdev,
                        part_c, part_a2, part_r, scratch3, rq_dev, start, wc, ctri, part_u, part_a, stc,
                        lx_x, lx_Lx, lx_p, lx_Lp, lx_Lw, lx_rT, lx_wT, lx_tl, lx_tdinv, lx_tcu, lx_part, lx_partR,
                        lx_ys, lx_pas, lx_as, lx_bs, lx_maps, lx_ba, lx_bd, lx_bu, wb_ui, wb_uj, wb_counts, wb_uc, wb_g, wb_h, wb_Zt, wb_Cm, wb_Cm2, wb_piv, wb_pas, wb_maps,
                        lx_colT, lx_bad, lx_st, valf};
        {
            void* pb[] = {panv.tptr, panv.thead, panv.bval, panv.bcol, panv.ypart, panv.coef, panv.cbase, panv.ps, panv.tick, panv.claim, panv.ovf, panv.bd, panv.bpk};
            for (void* q : pb) if (q) (void)hipFree(q);
            void* pu_[] = {pu_U0, pu_U1, pu.W, pu_sig, pan_bv32, d_ps};
            for (void* q : pu_) if (q) (void)hipFree(q);
            void* pk[] = {ppack.band, ppack.cc, ppack.crow, ppack.ccol, ppack.cval, ell_col, ell_val};
            for (void* q : pk) if (q) (void)hipFree(q);
        }
        if (h_lrec) (void)hipHostFree(h_lrec);
#ifdef MACHIP_EXPERIMENTS
        if (h_brec) (void)hipHostFree(h_brec);
        if (h_bs) (void)hipHostFree(h_bs);
        { void* pb2[] = {bZ0, bZ1, bpart, bU0, bwarm, bsdev, brec, (void*)bclk}; for (void* q : pb2) if (q) (void)hipFree(q); }
#endif
        for (void* p : ptrs) if (p) (void)hipFree(p);
        if (h_tri) (void)hipHostFree(h_tri);
        if (h_flag) (void)hipHostFree(h_flag);
        if (h_pin) (void)hipHostFree(h_pin);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (evs0) (void)hipEventDestroy(evs0);
        if (evs1) (void)hipEventDestroy(evs1);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    }

    // Cold start: u (the stored start vector or the pseudo-random fill, already in place) is multiplied by (landscape / max)^p after
    // `start_land` Jacobi sweeps (kernels.h).  Scratch: wc (diagonal), y_raw / w2 (sweeps), part_c (maxima) -- all idle until the
    // first explicit check.  Deterministic (fixed-order maxima), identical on every rank of a partitioned solve.
    bool start_guess = false;   // the start vector of this solve is the caller's own guess (machip_fiedler x0): left as it is
    // A warm start (MAC.Cache made real) only pays where consecutive Fiedler vectors resemble each other.  On the localised vectors of the
    // Erdos-Renyi configs they do not (overlap < 0.005: the weak spot moves every iteration), and the weighted cold start is 13 % faster.
    // The overlap of a solve's start vector with its result is free -- |s_0|, the first Ritz coefficient -- so a warm-started solve that
    // finds it below 2 / sqrt(n) -- no better than a random vector's -- sends the next `kWarmSkip` warm requests to the weighted cold start,
    // then probes again.  (Not a larger threshold: city10000's consecutive vectors overlap by 0.03 - 0.09 only and its warm start still
    // saves 16 % of the steps -- a smooth vector is rich in the low end of the spectrum.)  A function of the records alone: every rank of
    // a partitioned solve takes the same turn.
    static constexpr int kWarmSkip = 7;      // doubled by every further probe that fails in a row (at most 63): a trajectory whose vectors keep
    int warm_skip = 0, warm_fails = 0;       // moving pays for ever fewer probes
    double last_pr = 0.0;                    // participation ratio 1 / sum v^4 of the last explicitly checked vector (0: none yet on this handle)
    // the landscape after `sweeps` Jacobi sweeps (in y_raw or w2; per-workgroup maxima of the last sweep in part_c[0 .. pl.grid))
    // (the sweeps' only per-workgroup output are the maxima in part_c, 3 x kMaxGrid doubles: their grid may exceed kMaxGrid)
    SpmvPlan landscape_plan(long nnz) const { return plan_spmv(opt, n, nnz, kAuto, 3 * kMaxGrid); }
    // (*grid_out: how many workgroups left a maximum in part_c -- the CSR product's launch shape, or the panel row kernel's)
    const double* landscape_field(const CsrView& A, const SpmvPlan& pl, int sweeps, int* grid_out = nullptr) {
        k_land_init<<<vgrid(), kBlock, 0, stream>>>(A, wc, y_raw);
        double *src = y_raw, *dst = w2;
        const bool via_panel = pan_live && OPT(panel_ops, 1) != 0;
        for (int s = 0; s < sweeps; ++s) {
            OpLand op{src, dst, wc, part_c, 0.0};
            if (via_panel) panel_spmv(src, op, vgrid());
            else launch_spmv(pl, stream, A, src, op);
            std::swap(src, dst);
        }
        if (grid_out) *grid_out = via_panel ? vgrid() : pl.grid;
        return src;
    }
    // y = L x for a plain vector on the panel form of the running solve (k_pan_mul8 without the recurrence's prologue + k_pan_rowop<Op>)
    template <class Op>
    void panel_spmv(const double* x, const Op& op, int grid) {
        const int g1 = pan.NB * pan.NP;
        const PipeView L = pview(SpmvPlan());
        switch (pan.LPT * 10 + pan.TWT) {
#define MACHIP_PANU_CASE(LP, TW) case LP * 10 + TW: k_pan_mul8<LP, TW, double><<<g1, kPanThreads, 0, stream>>>(x, panv.tptr, panv.thead, panv.bval, panv.bcol, panv.n, panv.C, panv.NP | (panv.TWW << 16), panv.NTB << 16, panv, L, -1); break;
#define MACHIP_PANU_ROW(LP) MACHIP_PANU_CASE(LP, 3) MACHIP_PANU_CASE(LP, 5) MACHIP_PANU_CASE(LP, 8)
            MACHIP_PANU_ROW(1) MACHIP_PANU_ROW(2) MACHIP_PANU_ROW(3) MACHIP_PANU_ROW(4) MACHIP_PANU_ROW(5) MACHIP_PANU_ROW(6)
            MACHIP_PANU_ROW(7) MACHIP_PANU_ROW(8) MACHIP_PANU_CASE(9, 3)
#undef MACHIP_PANU_ROW
#undef MACHIP_PANU_CASE
            default: break;
        }
        k_pan_rowop<Op><<<grid, kBlock, 0, stream>>>(panv, x, op);
    }
    int landscape_start(const CsrView& A, const SpmvPlan& pl) {
        const int sweeps = std::min(16, OPT(start_land, 3));
        if (sweeps <= 0) return MACHIP_OK;
        const double pw = (double)std::max(1, OPT(start_pow, 128));
        int gmax = pl.grid;
        const double* f = landscape_field(A, pl, sweeps, &gmax);
        const double floor_w = 1e-6 * (double)std::max(0, OPT(start_floor_e6, 1000));      // (every entry keeps this share of its draw: kernels.h)
        k_land_weight<<<vgrid(), kBlock, 0, stream>>>(f, part_c, gmax, u, n, pw, floor_w);
        HIP_TRY(hipGetLastError());
        return MACHIP_OK;
    }
    int vgrid() const { return (int)std::min<long>(kMaxGrid, ((long)n + kBlock - 1) / kBlock); }

    template <typename T = double>
    PipeViewT<T> pview(const SpmvPlan& pl) const {
        PipeViewT<T> L;     // T = float: records and basis live in the same buffers, read as fp32
        L.n = n; L.st = st; L.Z0 = reinterpret_cast<ZRec<T>*>(Z0); L.Z1 = reinterpret_cast<ZRec<T>*>(Z1); L.V = reinterpret_cast<T*>(V); L.tri = tri; L.htri = d_htri; L.hflag = d_hflag; L.part = part; L.P = pl.grid; L.chunk = 0; L.pub = 0; L.pubstep = 0;
        L.sig = (pl.variant == kPanel && pan_u) ? pu_sig : nullptr;      // (shifted records: panel_u.h)
        return L;
    }
    LanView check_view(const SpmvPlan& pl) const {   // "column 0" machinery for the explicit check
        LanView L;
        L.n = n; L.st = st2; L.u = y_raw; L.w = w2; L.V = yvec; L.alpha = scratch3; L.beta = scratch3 + 2;
        L.l1 = scratch3 + 4; L.part_u = part_c; L.P_u = vgrid(); L.part_a = part_a2; L.P_a = pl.grid;
        return L;
    }

    LanView classic_view(const SpmvPlan& pl) const {
        LanView L;
        L.n = n; L.st = stc; L.u = u; L.w = wc; L.V = V; L.alpha = ctri; L.beta = ctri + (vcap + 2);
        L.l1 = ctri + 2 * (vcap + 2); L.part_u = part_u; L.P_u = vgrid(); L.part_a = part_a; L.P_a = pl.grid;
        return L;
    }
    // Classic Lanczos chunk: per step one SpMV kernel (v_j = (u - mean)/||u|| with the norm taken
    // from the vector itself, w = L v_j, alpha partials) and one update kernel (u <- w - alpha v_j
    // - beta v_{j-1}).  Twice the launches of the pipelined form, but beta_j is computed from u_j
    // directly, so it stays accurate when ||u_j|| << ||L v_j|| (restart from a good vector, Krylov
    // space nearly exhausted), where the pipelined quadratic form has lost its digits.
    void enqueue_classic(const CsrView& A, const SpmvPlan& pl, int steps) {
        OpLanczos op;
        op.L = classic_view(pl);
        const int g2 = vgrid();
        for (int s = 0; s < steps; ++s) {
            launch_spmv(pl, stream, A, op.L.u, op);
            k_lan_update<<<g2, kBlock, 0, stream>>>(op.L);
        }
        k_lan_tail<<<1, kBlock, 0, stream>>>(op.L);
    }

    // ---- small graphs: a whole chunk of Lanczos steps in one single-workgroup launch (persist.h) ----
    template <typename T = double>
    PersistViewT<T> persist_view() const {
        PersistViewT<T> L;
        L.n = n; L.st = st; L.u = u; L.vprev = wc; L.V = reinterpret_cast<T*>(V); L.tri = tri; L.htri = d_htri; L.hflag = d_hflag;
        return L;
    }
    // Build the packed form of A for this solve (one single-workgroup launch; the chunks then start from coalesced loads).
    int pack_persist(const CsrView& A) {
        if (!ppack.band) {
            ST_TRY(dev_alloc(&ppack.band, 5 * (size_t)kPersistPad)); ST_TRY(dev_alloc(&ppack.cc, 2 * (size_t)kPersistPad));
            ST_TRY(dev_alloc(&ppack.crow, (size_t)kPersistPad + 1));
            ST_TRY(dev_alloc(&ppack.ccol, kPersistPackEntries)); ST_TRY(dev_alloc(&ppack.cval, kPersistPackEntries));
        }
        k_persist_pack<<<1, 1024, 0, stream>>>(A, ppack);
        HIP_TRY(hipGetLastError());
        return MACHIP_OK;
    }
    template <typename T>
    void launch_persist_t(const CsrView& A, int steps) {
        const PersistViewT<T> L = persist_view<T>();
        switch ((n + kPersistThreads - 1) / kPersistThreads) {   // rows per thread (sphere2500: 5 -- every row slot costs LDS reads in every step)
            case 1: case 2: k_lan_persist<2, T><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps); break;
            case 3: k_lan_persist<3, T><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps); break;
            case 4: k_lan_persist<4, T><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps); break;
            case 5: k_lan_persist<5, T><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps); break;
            default: k_lan_persist<6, T><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps); break;
        }
    }
    void launch_persist(const CsrView& A, int steps, bool f32 = false, const PersistCheb& ch = PersistCheb()) {
#ifdef MACHIP_EXPERIMENTS
        if (ch.deg >= 2) {       // Chebyshev-filtered recurrence (fp64; measured slower: DESIGN "negatives")
            const PersistView L = persist_view<double>();
            switch ((n + 2 * kPersistThreads - 1) / (2 * kPersistThreads)) {
                case 1: k_lan_persist<2, double, true><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps, ch); break;
                case 2: k_lan_persist<4, double, true><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps, ch); break;
                default: k_lan_persist<6, double, true><<<1, kPersistThreads, 0, stream>>>(ppack, L, steps, ch); break;
            }
            return;
        }
#endif
        (void)ch;
        if (f32) launch_persist_t<float>(A, steps); else launch_persist_t<double>(A, steps);
    }

    // ---- column-panel step (panel.h) ---------------------------------------------------------------
    // What the assembly's fill pass writes on the side for the panel form (kernels.h PanSpec): the panel shape is a function of n
    // (and of the handle's options) alone, so the per-row table exists before the host has seen nnz.  NP = 0: nothing to write.
    int pan_spec_prepare(PanSpec* out) {
        *out = PanSpec();
        pan_rows_ready = false;
        if (!pan_allowed || !pat.prow) return MACHIP_OK;
        const PanPlan sh = plan_panel(opt, n, 0, 1, true, (long)csr_cap, true, OPT(stream, 1) != 0);      // (the shape solve() will plan)
        if (!sh.on) return MACHIP_OK;
        ST_TRY(pan_row_buffers(sh, sh.band));
        HIP_TRY(hipMemsetAsync(panv.ovf, 0, sizeof(int), stream));
        out->NP = sh.NP; out->C = sh.C; out->band = sh.band ? 1 : 0;
        out->ps = panv.ps; out->bd = panv.bd; out->bpk = panv.bpk; out->ovf = panv.ovf;
        pan_rows_ready = true; pan_rows_NP = sh.NP; pan_rows_C = sh.C; pan_rows_band = sh.band;
        return MACHIP_OK;
    }
    template <class T>
    int pan_regrow(T** ptr, size_t count, bool* dropped) {
        if (*ptr) { if (!*dropped) { HIP_TRY(hipStreamSynchronize(stream)); drop_graphs(); *dropped = true; } (void)hipFree(*ptr); *ptr = nullptr; }
        return dev_alloc(ptr, count);
    }
    // per-row tables of the panel form (ps, band) and the per-panel partial products
    int pan_row_buffers(const PanPlan& pn, bool band) {
        bool dropped = false;
        if ((size_t)pn.NP * (size_t)n > pan_y_cap) {
            ST_TRY(pan_regrow(&panv.ypart, (size_t)pn.NP * ((size_t)n + 2), &dropped));     // (k_pan_step: even plane stride, pairs of rows)
            ST_TRY(pan_regrow(&panv.ps, ((size_t)pn.NP + 1) * (size_t)n, &dropped));
            pan_y_cap = (size_t)pn.NP * (size_t)n;
            pan_rows_ready = false;
        }
        if (!panv.ovf) ST_TRY(dev_alloc(&panv.ovf, 1));
        if (band && (size_t)n > pan_band_cap) {
            ST_TRY(pan_regrow(&panv.bd, 3 * (size_t)n, &dropped)); ST_TRY(pan_regrow(&panv.bpk, (size_t)n, &dropped));
            pan_band_cap = (size_t)n;
            pan_rows_ready = false;
        }
        return MACHIP_OK;
    }
    // (Re)build the panel form of A on the stream: buffers grow on demand (cached chunk graphs carry their addresses).
    // rows_ready: the assembly has already written the per-row tables (ps, band) for this shape -- k_pan_rows is not needed.
    int ensure_panel(const CsrView& A, long nnz, const PanPlan& pn, bool band = false, bool rows_ready = false) {
        const size_t cells = (size_t)pn.NB * pn.NP, NTP = (size_t)kPanWork * pn.TWW;
        const size_t NT = cells * NTP;
        bool dropped = false;
        if (NT > pan_nt_cap) {
            ST_TRY(pan_regrow(&panv.tptr, cells * (NTP + 1) + 1, &dropped)); ST_TRY(pan_regrow(&panv.thead, NT * 64, &dropped));
            pan_nt_cap = NT;
        }
        ST_TRY(pan_row_buffers(pn, band));
        if (!panv.coef) ST_TRY(dev_alloc(&panv.coef, 8));
        if (!panv.tick) { ST_TRY(dev_alloc(&panv.tick, 256)); ST_TRY(dev_alloc(&panv.claim, 4096)); }     // (NB <= 256, NB NP <= 4096: plan_panel)
        // band mode (Lanczos form, two launches): diagonal and chain neighbours stay out of the tiles -- k_pan_fin adds them
        panv.band = band ? 1 : 0;
        panv.rev = OPT(panel_rev, 1) != 0 ? 1 : 0;     // odd steps of the shifted form walk each wave's chunks backwards (panel_u.h: what the XCD's L2 still holds comes first); 0: forwards always
        panv.spin_ticks = OPT(panel_spin_us, 20) * 100;
#ifdef PAN_CLOCKS
        if (!panv.clk) { ST_TRY(dev_alloc(&panv.clk, (size_t)16 * kMaxGrid)); HIP_TRY(hipMemsetAsync(panv.clk, 0, sizeof(long long) * 16 * kMaxGrid, stream)); }
#endif
        panv.n = n; panv.NP = pn.NP; panv.C = pn.C; panv.NB = pn.NB; panv.NTB = pn.NTB; panv.TWW = pn.TWW; panv.CELLS = pn.cells;
        // static ranges of the cells in the value / column arrays: pattern slots of the cell + its rows (the diagonal, when the band
        // stays in the tiles) + 64 x 128 entries of zero padding (the tile heights telescope), whole 64-entry chunks; once per shape
        const std::array<int, 5> key{pn.NP, pn.C, pn.NB, pn.NTB, pn.TWW};
        if (!panv.cbase || key != pan_shape_key) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (!dropped) { drop_graphs(); dropped = true; }
            if (panv.cbase) { (void)hipFree(panv.cbase); panv.cbase = nullptr; }
            ST_TRY(dev_alloc(&panv.cbase, cells + 1));
            std::vector<int> cnt(cells + 1, 0);
            if (pat.prow) {
                HIP_TRY(hipMemsetAsync(panv.cbase, 0, sizeof(int) * (cells + 1), stream));
                k_pan_cellcap<<<(n + kBlock - 1) / kBlock, kBlock, 0, stream>>>(pat, 64 * pn.NTB, pn.C, pn.NP, panv.cbase);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemcpyAsync(cnt.data(), panv.cbase, sizeof(int) * cells, hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            } else {
                return fail(MACHIP_BAD_ARG, "column-panel form without a pattern");
            }
            std::vector<int> base(cells + 1, 0);
            size_t run = 0;
            for (size_t c = 0; c < cells; ++c) {
                base[c] = (int)run;
                run += (((size_t)cnt[c] + (size_t)64 * pn.NTB + 63) / 64) * 64 + (size_t)64 * (kPanMaxLen + 1);
                if (run > 2000000000ull) return fail(MACHIP_BAD_ARG, "column-panel form exceeds int32 indexing");
            }
            base[cells] = (int)run;
            HIP_TRY(hipMemcpyAsync(panv.cbase, base.data(), sizeof(int) * (cells + 1), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (run + kPanSlack > pan_cap) {
                ST_TRY(pan_regrow(&panv.bval, run + kPanSlack, &dropped)); ST_TRY(pan_regrow(&panv.bcol, run + kPanSlack, &dropped));
                pan_cap = run + kPanSlack;
            }
            pan_shape_key = key;
        }
        if (!rows_ready) {
            HIP_TRY(hipMemsetAsync(panv.ovf, 0, sizeof(int), stream));
            k_pan_rows<<<(n + kBlock - 1) / kBlock, kBlock, 0, stream>>>(A, panv);
            // (the per-row tables now describe THIS shape: a later solve of the same matrix in the shape the assembly had written them for
            // must not take the old flag for them -- advisor finding on round 5)
            pan_rows_ready = true; pan_rows_NP = pn.NP; pan_rows_C = pn.C; pan_rows_band = band;
        }
        {
            const int g = pan_build_grid(pn.NB, pn.NP);
            switch ((64 * pn.NTB + kPanThreads - 1) / kPanThreads) {      // rows per thread of the build kernel
                case 1: k_pan_build<1><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                case 2: k_pan_build<2><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                case 3: k_pan_build<3><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                case 4: k_pan_build<4><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                case 5: k_pan_build<5><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                case 6: k_pan_build<6><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                case 7: k_pan_build<7><<<g, kPanThreads, 0, stream>>>(A, panv); break;
                default: k_pan_build<8><<<g, kPanThreads, 0, stream>>>(A, panv); break;
            }
        }
        HIP_TRY(hipGetLastError());
        (void)nnz;
        return MACHIP_OK;
    }
#ifdef PAN_CLOCKS
#define MACHIP_FINU_CLK , (const PeerSet*)nullptr, 0, 0, panv.clk
#else
#define MACHIP_FINU_CLK
#endif
    void launch_pan_step(const PipeView& L, int s, int jhost = -1) {
        const int g1 = pan.NB * pan.NP;
#ifdef MACHIP_EXPERIMENTS
        if (pan.fused) {     // one launch per step (k_pan_step; measured slower: profiles/r4_c4_one_launch_step.md)
            switch (pan.RPT) {
#define MACHIP_PAN_CASE(R) case R: k_pan_step<R><<<g1, kPanThreads, 0, stream>>>(PAN_STEP_ARGS(panv, L, s)); break;
                MACHIP_PAN_CASE(1) MACHIP_PAN_CASE(2) MACHIP_PAN_CASE(3) MACHIP_PAN_CASE(4) MACHIP_PAN_CASE(5) MACHIP_PAN_CASE(6)
                MACHIP_PAN_CASE(7) MACHIP_PAN_CASE(8) MACHIP_PAN_CASE(9) MACHIP_PAN_CASE(10) MACHIP_PAN_CASE(11) MACHIP_PAN_CASE(12)
#undef MACHIP_PAN_CASE
                default: k_pan_step<13><<<g1, kPanThreads, 0, stream>>>(PAN_STEP_ARGS(panv, L, s)); break;
            }
            return;
        }
#endif
        if (pan_u) {             // shifted recurrence (panel_u.h): 8-byte operand, no prologue in the matrix kernel; the row kernel leads
            if (pan32 && jhost >= 0 && jhost >= pan32_from) {      // mixed mode: this step reads fp32 tile values (TWT = 3 shapes only: solve() checks)
                switch (pan.LPT) {
#define MACHIP_PANU32_CASE(LP) case LP: k_pan_mul8<LP, 3, float><<<g1, kPanThreads, 0, stream>>>(PAN_MUL8_ARGS(panv, pu, L, s), pan_bv32); break;
                    MACHIP_PANU32_CASE(1) MACHIP_PANU32_CASE(2) MACHIP_PANU32_CASE(3) MACHIP_PANU32_CASE(4) MACHIP_PANU32_CASE(5) MACHIP_PANU32_CASE(6)
                    MACHIP_PANU32_CASE(7) MACHIP_PANU32_CASE(8) MACHIP_PANU32_CASE(9)
#undef MACHIP_PANU32_CASE
                    default: break;
                }
            } else
            switch (pan.LPT * 10 + pan.TWT) {
#define MACHIP_PANU_CASE(LP, TW) case LP * 10 + TW: k_pan_mul8<LP, TW><<<g1, kPanThreads, 0, stream>>>(PAN_MUL8_ARGS(panv, pu, L, s)); break;
#define MACHIP_PANU_ROW(LP) MACHIP_PANU_CASE(LP, 3) MACHIP_PANU_CASE(LP, 5) MACHIP_PANU_CASE(LP, 8)
                MACHIP_PANU_ROW(1) MACHIP_PANU_ROW(2) MACHIP_PANU_ROW(3) MACHIP_PANU_ROW(4) MACHIP_PANU_ROW(5) MACHIP_PANU_ROW(6)
                MACHIP_PANU_ROW(7) MACHIP_PANU_ROW(8) MACHIP_PANU_CASE(9, 3)
#undef MACHIP_PANU_ROW
#undef MACHIP_PANU_CASE
                default: break;      // (plan_panel only hands out instantiated shapes)
            }
            const int npm = pan.NP <= 6 ? 6 : pan.NP <= 8 ? 8 : pan.NP <= 12 ? 12 : 16;
            switch (pan.block2 * 100 + npm) {
#define MACHIP_FINU_CASE(B, M) case B * 100 + M: k_pan_finu<B, M><<<pan.grid2, B, 0, stream>>>(PAN_FINU_ARGS(panv, pu, L, s, jhost) MACHIP_FINU_CLK); break;
#define MACHIP_FINU_ROW(B) MACHIP_FINU_CASE(B, 6) MACHIP_FINU_CASE(B, 8) MACHIP_FINU_CASE(B, 12) MACHIP_FINU_CASE(B, 16)
                MACHIP_FINU_ROW(256) MACHIP_FINU_ROW(512) MACHIP_FINU_ROW(1024)
#undef MACHIP_FINU_ROW
#undef MACHIP_FINU_CASE
                default: break;
            }
            return;
        }
        if (pan.cells > 1) {     // several row blocks per workgroup, the panel loaded once (k_pan_mul_multi)
            const int gm = pan.NP * ((pan.NB + pan.cells - 1) / pan.cells);
            switch (pan.RPT) {
#define MACHIP_PAN_CASE(R) case R: k_pan_mul_multi<R><<<gm, kPanThreads, 0, stream>>>(PAN_MUL_ARGS(panv, L, s)); break;
                MACHIP_PAN_CASE(1) MACHIP_PAN_CASE(2) MACHIP_PAN_CASE(3) MACHIP_PAN_CASE(4) MACHIP_PAN_CASE(5) MACHIP_PAN_CASE(6)
                MACHIP_PAN_CASE(7) MACHIP_PAN_CASE(8) MACHIP_PAN_CASE(9) MACHIP_PAN_CASE(10) MACHIP_PAN_CASE(11) MACHIP_PAN_CASE(12)
#undef MACHIP_PAN_CASE
                default: k_pan_mul_multi<13><<<gm, kPanThreads, 0, stream>>>(PAN_MUL_ARGS(panv, L, s)); break;
            }
        } else switch (pan.RPT) {
#define MACHIP_PAN_CASE(R) case R: k_pan_mul<R><<<g1, kPanThreads, 0, stream>>>(PAN_MUL_ARGS(panv, L, s)); break;
            MACHIP_PAN_CASE(1) MACHIP_PAN_CASE(2) MACHIP_PAN_CASE(3) MACHIP_PAN_CASE(4) MACHIP_PAN_CASE(5) MACHIP_PAN_CASE(6)
            MACHIP_PAN_CASE(7) MACHIP_PAN_CASE(8) MACHIP_PAN_CASE(9) MACHIP_PAN_CASE(10) MACHIP_PAN_CASE(11) MACHIP_PAN_CASE(12)
#undef MACHIP_PAN_CASE
            default: k_pan_mul<13><<<g1, kPanThreads, 0, stream>>>(PAN_MUL_ARGS(panv, L, s)); break;
        }
        if (pan.block2 == 1024) k_pan_fin<1024><<<pan.grid2, 1024, 0, stream>>>(PAN_FIN_ARGS(panv, L, s));
        else if (pan.block2 == 512) k_pan_fin<512><<<pan.grid2, 512, 0, stream>>>(PAN_FIN_ARGS(panv, L, s));
        else k_pan_fin<256><<<pan.grid2, 256, 0, stream>>>(PAN_FIN_ARGS(panv, L, s));
    }

#ifdef MACHIP_EXPERIMENTS
    // y_p = L[block, panel p] w for a plain operand vector (k_pan_mul<RPT, RAW = true>): the diagonally preconditioned mode's product
    void launch_pan_mul_raw(const double* w) {
        const int g1 = pan.NB * pan.NP;
        PipeView L{};        // (unused by the RAW instantiation: no records, no prologue)
        L.n = n; L.part = lx_part + (size_t)2 * kLobNS * kMaxGrid; L.st = st;      // (part: w^T L w per (row block, panel) cell)
        const Z2* wz = reinterpret_cast<const Z2*>(w);
        switch (pan.RPT) {
#define MACHIP_PAN_CASE(R) case R: k_pan_mul<R, true><<<g1, kPanThreads, 0, stream>>>(wz, L.part, L.st, panv.tptr, panv.thead, panv.n, panv.C, panv.NP, panv.TWW, panv, L, 0); break;
            MACHIP_PAN_CASE(1) MACHIP_PAN_CASE(2) MACHIP_PAN_CASE(3) MACHIP_PAN_CASE(4) MACHIP_PAN_CASE(5) MACHIP_PAN_CASE(6)
            MACHIP_PAN_CASE(7) MACHIP_PAN_CASE(8) MACHIP_PAN_CASE(9) MACHIP_PAN_CASE(10) MACHIP_PAN_CASE(11) MACHIP_PAN_CASE(12)
#undef MACHIP_PAN_CASE
            default: k_pan_mul<13, true><<<g1, kPanThreads, 0, stream>>>(wz, L.part, L.st, panv.tptr, panv.thead, panv.n, panv.C, panv.NP, panv.TWW, panv, L, 0); break;
        }
    }
#endif

    // ---- row-partitioned chunk: per step one launch per rank, ordered by events (ShardGroup) ----
    void shard_split(const SpmvPlan& pl) {
        const int R = (int)shard->rk.size();
        for (int r = 0; r < R; ++r) { shard->rk[(size_t)r].g0 = (int)((long)pl.grid * r / R); shard->rk[(size_t)r].g1 = (int)((long)pl.grid * (r + 1) / R); }
    }
    PeerSet shard_peers(const ShardRank& me, const SpmvPlan& pl) const {
        PeerSet PS;
        PS.n = (int)shard->rk.size(); PS.first = me.g0; PS.total = pl.grid;
        for (int q = 0; q < PS.n; ++q) { PS.Z0[q] = shard->rk[(size_t)q].Z0; PS.Z1[q] = shard->rk[(size_t)q].Z1; PS.part[q] = shard->rk[(size_t)q].part; }
        return PS;
    }
    void launch_chunk_sharded(const SpmvPlan& pl, int steps) {
        ShardGroup& G = *shard;
        const int R = (int)G.rk.size();
        shard_split(pl);
        (void)hipEventRecord(G.fork, stream);
        for (int q = 1; q < R; ++q) (void)hipStreamWaitEvent(G.rk[(size_t)q].stream, G.fork, 0);
        for (int s = 0; s < steps; ++s) {
            for (int r = 0; r < R; ++r) {
                ShardRank& me = G.rk[(size_t)r];
                if (s > 0) for (int q = 0; q < R; ++q) if (q != r) (void)hipStreamWaitEvent(me.stream, G.rk[(size_t)q].ev[(s - 1) & 1], 0);
                if (me.g1 > me.g0) {
                    if (!G.same_device) (void)hipSetDevice(me.device);
                    PipeView L = pview(pl);                       // the leader's counters / tridiagonal records ...
                    L.Z0 = me.Z0; L.Z1 = me.Z1; L.V = me.V; L.part = me.part;   // ... this rank's operand, basis rows and partial sums
                    launch_pipe_shard(pl, me.stream, me.A, L, s, shard_peers(me, pl), me.g1 - me.g0);
                }
                (void)hipEventRecord(me.ev[s & 1], me.stream);
            }
        }
        for (int q = 1; q < R; ++q) (void)hipStreamWaitEvent(stream, G.rk[(size_t)q].ev[(steps - 1) & 1], 0);
        if (!G.same_device) (void)hipSetDevice(G.rk[0].device);
        k_pipe_tail<<<1, 64, 0, stream>>>(pview(pl), steps);
    }
    // ---- row-partitioned chunk between processes: my workgroups only, ordered by flags in device memory (IpcGroup) ----
    PeerSet ipc_peers(const SpmvPlan& pl) const {
        PeerSet PS;
        PS.n = ipc->nranks; PS.first = ipc->g0; PS.total = pl.grid;
        for (int q = 0; q < PS.n; ++q) { PS.Z0[q] = ipc->Z0[q]; PS.Z1[q] = ipc->Z1[q]; PS.part[q] = ipc->part[q]; }
        return PS;
    }
    void ipc_split(const SpmvPlan& pl) {
        ipc->g0 = (int)((long)pl.grid * ipc->rank / ipc->nranks);
        ipc->g1 = (int)((long)pl.grid * (ipc->rank + 1) / ipc->nranks);
    }
    void launch_chunk_ipc(const CsrView& A, const SpmvPlan& pl, int steps) {
        ipc_split(pl);
        const PipeView L = pview(pl);
        const PeerSet PS = ipc_peers(pl);
        k_ipc_wait<<<1, 64, 0, stream>>>(ipc->view, 0);              // (whatever the previous chunk / the sequence start left pending)
        for (int s = 0; s < steps; ++s) {
            launch_pipe_shard(pl, stream, A, L, s, PS, ipc->g1 - ipc->g0);
            k_ipc_pubwait<<<1, 64, 0, stream>>>(ipc->view, 0);       // mine delivered; every peer has delivered its records / partial sums of this step
        }
        k_pipe_tail<<<1, 64, 0, stream>>>(L, steps);
    }
    // The same for the column-panel step of the shifted recurrence (round 6): rank r owns the row kernel's workgroups [g0, g1) -- rows
    // [g0 B, g1 B) -- and launches the matrix kernel for the row blocks those rows lie in, all panels each (a block that straddles two ranks
    // is multiplied by both: 1 of 42 at configs[3]); the row kernel writes its rows of the next operand and its six sums into every rank's
    // copy.  Cells, row sums, partial sums and their order are those of the un-partitioned launch: bit-identical.
    void launch_chunk_ipc_pan(const SpmvPlan& pl, int steps, int j0) {
        ipc_split(pl);
        const PipeView L = pview(pl);
        const PeerSet PS = ipc_peers(pl);
        if (!d_ps) { if (dev_alloc(&d_ps, 1) != MACHIP_OK) return; }
        if (memcmp(&PS, &h_ps, sizeof(PeerSet)) != 0) {       // (the row kernel reads its peer set from device memory: panel_u.h)
            h_ps = PS;
            (void)hipMemcpyAsync(d_ps, &h_ps, sizeof(PeerSet), hipMemcpyHostToDevice, stream);
        }
        const int Rb = 64 * pan.NTB;
        const long r_lo = (long)ipc->g0 * pan.block2, r_hi = std::min<long>((long)ipc->g1 * pan.block2, (long)n);
        const int b0 = (int)(r_lo / Rb), b1 = (int)((r_hi + Rb - 1) / Rb);
        const int g1m = std::max(1, b1 - b0) * pan.NP, gf = ipc->g1 - ipc->g0;
        const int npm = pan.NP <= 6 ? 6 : pan.NP <= 8 ? 8 : pan.NP <= 12 ? 12 : 16;
        k_ipc_wait<<<1, 64, 0, stream>>>(ipc->view, 0);              // (whatever the previous chunk / the sequence start left pending)
        for (int s = 0; s < steps; ++s) {
            const int jhost = j0 >= 0 ? j0 + s : -1;
            switch (pan.LPT * 10 + pan.TWT) {
#define MACHIP_PANU_CASE(LP, TW) case LP * 10 + TW: k_pan_mul8<LP, TW, double><<<g1m, kPanThreads, 0, stream>>>(PAN_MUL8_ARGS_AT(panv, pu, L, s, b0)); break;
#define MACHIP_PANU_ROW(LP) MACHIP_PANU_CASE(LP, 3) MACHIP_PANU_CASE(LP, 5) MACHIP_PANU_CASE(LP, 8)
                MACHIP_PANU_ROW(1) MACHIP_PANU_ROW(2) MACHIP_PANU_ROW(3) MACHIP_PANU_ROW(4) MACHIP_PANU_ROW(5) MACHIP_PANU_ROW(6)
                MACHIP_PANU_ROW(7) MACHIP_PANU_ROW(8) MACHIP_PANU_CASE(9, 3)
#undef MACHIP_PANU_ROW
#undef MACHIP_PANU_CASE
                default: break;
            }
            switch (pan.block2 * 100 + npm) {
#define MACHIP_FINU_CASE(B, M) case B * 100 + M: k_pan_finu<B, M, true><<<gf, B, 0, stream>>>(PAN_FINU_ARGS(panv, pu, L, s, jhost), d_ps, PS.first, PS.total); break;
#define MACHIP_FINU_ROW(B) MACHIP_FINU_CASE(B, 6) MACHIP_FINU_CASE(B, 8) MACHIP_FINU_CASE(B, 12) MACHIP_FINU_CASE(B, 16)
                MACHIP_FINU_ROW(256) MACHIP_FINU_ROW(512) MACHIP_FINU_ROW(1024)
#undef MACHIP_FINU_ROW
#undef MACHIP_FINU_CASE
                default: break;
            }
            k_ipc_pubwait<<<1, 64, 0, stream>>>(ipc->view, 0);       // mine delivered; every peer has delivered its operand rows / partial sums of this step
        }
        k_pipe_tail<<<1, 64, 0, stream>>>(L, steps);
    }
    // y_raw = V[:, :J] s: my rows of the basis -> my rows of EVERY rank's y_raw, then everybody holds the whole vector
    int ipc_ritz(const SpmvPlan& pl, int J, const double* s_host) {
        const int g2 = vgrid();
        const int KS = std::max(1, std::min(ks_max, J / 8));
        memcpy(h_pin, s_host, sizeof(double) * (size_t)J);
        HIP_TRY(hipMemcpyAsync(sdev, h_pin, sizeof(double) * (size_t)J, hipMemcpyHostToDevice, stream));
        ipc_split(pl);
        RowOwner own; own.gpb = pl.variant == kPanel ? pl.block : pipe_gpb(pl); own.gtot = pl.grid; own.g0 = ipc->g0; own.g1 = ipc->g1;      // (panel form: the row kernel's workgroups own the rows)
        PeerVecs all; all.n = ipc->nranks;
        for (int q = 0; q < all.n; ++q) all.v[q] = ipc->yraw[q];
        k_ritz_partial<<<dim3(g2, KS), kBlock, 0, stream>>>(V, n, J, sdev, ypart, own);
        k_ritz_own_rows<<<g2, kBlock, 0, stream>>>(ypart, n, KS, y_raw, own, all);
        k_ipc_pubwait<<<1, 64, 0, stream>>>(ipc->view, 1);
        k_vec_sums<<<g2, kBlock, 0, stream>>>(y_raw, n, part_c);
        HIP_TRY(hipGetLastError());
        return MACHIP_OK;
    }
    int ipc_check_err(const char* where) {
        if (ipc && ipc->h_err && *(volatile int*)ipc->h_err)
            return fail(MACHIP_RCCL_ERROR, std::string("inter-process communicator: ") + (*(volatile int*)ipc->h_err == 2 ? "a peer rank raised abort" : "a peer rank did not deliver within the time limit (stalled or dead)") + " (" + where + ")");
        return MACHIP_OK;
    }

    // start of a sequence: the leader's k_pipe_init output (records of u, first partial sums) into every rank's copy
    int shard_broadcast_init() {
        ShardGroup& G = *shard;
        for (size_t q = 1; q < G.rk.size(); ++q) {
            HIP_TRY(hipMemcpyAsync(G.rk[q].Z0, Z0, sizeof(Z2) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipMemcpyAsync(G.rk[q].part, part, sizeof(double) * 2 * kNP * kMaxGrid, hipMemcpyDeviceToDevice, stream));
        }
        return MACHIP_OK;
    }
    // y_raw = V[:, :J] s with V spread over the ranks by rows; sums of y_raw in part_c (the layout check_vector expects)
    int shard_ritz(const SpmvPlan& pl, int J, const double* s_host) {
        ShardGroup& G = *shard;
        const int R = (int)G.rk.size(), g2 = vgrid();
        const int KS = std::max(1, std::min(ks_max, J / 8));
        memcpy(h_pin, s_host, sizeof(double) * (size_t)J);
        shard_split(pl);
        HIP_TRY(hipEventRecord(G.fork, stream));
        for (int r = 0; r < R; ++r) {
            ShardRank& me = G.rk[(size_t)r];
            if (r) HIP_TRY(hipStreamWaitEvent(me.stream, G.fork, 0));
            if (!G.same_device) HIP_TRY(hipSetDevice(me.device));
            HIP_TRY(hipMemcpyAsync(me.sdev, h_pin, sizeof(double) * (size_t)J, hipMemcpyHostToDevice, me.stream));
            RowOwner own; own.gpb = pipe_gpb(pl); own.gtot = pl.grid; own.g0 = me.g0; own.g1 = me.g1;
            k_ritz_partial<<<dim3(g2, KS), kBlock, 0, me.stream>>>(me.V, n, J, me.sdev, me.ypart, own);
            k_ritz_own_rows<<<g2, kBlock, 0, me.stream>>>(me.ypart, n, KS, y_raw, own);
            HIP_TRY(hipEventRecord(me.ev[0], me.stream));
        }
        for (int q = 1; q < R; ++q) HIP_TRY(hipStreamWaitEvent(stream, G.rk[(size_t)q].ev[0], 0));
        if (!G.same_device) HIP_TRY(hipSetDevice(G.rk[0].device));
        k_vec_sums<<<g2, kBlock, 0, stream>>>(y_raw, n, part_c);
        HIP_TRY(hipGetLastError());
        return MACHIP_OK;
    }

    // ---- one chunk = `steps` step kernels + the tail kernel ------------------------------------
    // tailless: no tail kernel -- the chunk's last step advances the counters and the NEXT chunk's first step (pub = this chunk's
    // step count there) hands the records to the host; a chunk that turns out to have no successor gets its tail from flush_tail.
    // j0 >= 0: the sequence's step index of the chunk's first step, known to the host (eager launches only: a captured chunk is replayed at any base)
    void launch_chunk(const CsrView& A, const SpmvPlan& pl, int steps, bool f32 = false, bool tailless = false, int pub = 0, int j0 = -1) {
        if (pl.variant == kPanel && seq_ipc) { launch_chunk_ipc_pan(pl, steps, j0); return; }
        if (pl.variant == kPanel) {
            PipeView L = pview(pl);
            L.chunk = tailless ? steps : 0; L.pub = std::max(pub, 0); L.pubstep = pub < 0;
            for (int s = 0; s < steps; ++s) launch_pan_step(L, s, j0 >= 0 ? j0 + s : -1);
            if (!tailless) k_pipe_tail<<<1, 64, 0, stream>>>(L, steps);
            return;
        }
        if (shard && pl.variant == kVec && !f32) { launch_chunk_sharded(pl, steps); return; }
        if (seq_ipc && pl.variant == kVec && !f32) { launch_chunk_ipc(A, pl, steps); return; }
        if (f32) {
            PipeViewT<float> L = pview<float>(pl);
            L.chunk = tailless ? steps : 0; L.pub = std::max(pub, 0); L.pubstep = pub < 0;
            const CsrViewT<float> Af{A.n, A.rowptr, A.col, valf};
            for (int s = 0; s < steps; ++s) launch_pipe(pl, stream, Af, L, s);
            if (!tailless) k_pipe_tail<<<1, 64, 0, stream>>>(L, steps);
            return;
        }
        PipeView L = pview(pl);
        L.chunk = tailless ? steps : 0; L.pub = std::max(pub, 0); L.pubstep = pub < 0;
        const CsrView& As = pl.variant == kEll ? ell_view : A;
        for (int s = 0; s < steps; ++s) launch_pipe(pl, stream, As, L, s);
        if (!tailless) k_pipe_tail<<<1, 64, 0, stream>>>(L, steps);
    }
    void flush_tail(const SpmvPlan& pl, int steps, bool f32) {
        if (f32) k_pipe_tail<<<1, 64, 0, stream>>>(pview<float>(pl), steps);
        else k_pipe_tail<<<1, 64, 0, stream>>>(pview(pl), steps);
    }
    int enqueue_chunk(const CsrView& A, const SpmvPlan& pl, int steps, bool f32 = false, bool tailless = false, int pub = 0, int j0 = -1) {
        const bool sharded = shard && pl.variant == kVec && !f32;
        // row-partitioned chunks are launched eagerly: a captured chunk would be one graph of steps x ranks kernel nodes with
        // ranks - 1 cross-stream dependencies each (ROCm 7.2 crashes on it from 8 ranks on one device), and across devices
        // a single graph is not an option anyway
        if (!use_graph(cur_launch_us) || sharded) { launch_chunk(A, pl, steps, f32, tailless, pub, j0); HIP_TRY(hipGetLastError()); return MACHIP_OK; }
        if (graph_csr_key != (const void*)A.val) {   // different matrix buffers: cached graphs are stale
            for (auto& kv : graphs) for (hipGraphExec_t ge : kv.second) if (ge) (void)hipGraphExecDestroy(ge);
            graphs.clear();
            graph_csr_key = (const void*)A.val;
        }
        const auto key = std::make_tuple(pl.variant + (f32 ? 100 : 0) + 1000 * pl.defer + (sharded ? 100000 * (int)shard->rk.size() : 0) + (seq_ipc && pl.variant == kVec && !f32 ? 10000000 : 0), pl.width * 10 + pl.unroll, pl.grid, pl.block, steps + 1000 * (tailless ? 1 : 0) + 10000 * pub);
        auto it = graphs.find(key);
        if (it == graphs.end()) it = graphs.emplace(key, std::array<hipGraphExec_t, 2>{nullptr, nullptr}).first;
        // two executables per shape, used alternately: with one chunk running ahead, the same
        // hipGraphExec is never in flight twice
        hipGraphExec_t& ge = it->second[(size_t)(graph_flip++ & 1)];
        if (!ge) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            launch_chunk(A, pl, steps, f32, tailless, pub);
            HIP_TRY(hipStreamEndCapture(stream, &g));
            HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
        }
        HIP_TRY(hipGraphLaunch(ge, stream));
        return MACHIP_OK;
    }

    // y = V[:, :J] s  -> normalised into yvec; w2 = L yvec; returns (rq, ||w2 - rq yvec||_1).
    int explicit_check(const CsrView& A, const SpmvPlan& pl, int J, const double* s_host, double* rq,
                       double* res_l1, bool f32 = false) {
        if (seq_sharded && seq_ipc) { ST_TRY(ipc_ritz(seq_plan, J, s_host)); return check_vector(A, pl, rq, res_l1); }
        if (seq_sharded) { ST_TRY(shard_ritz(seq_plan, J, s_host)); return check_vector(A, pl, rq, res_l1); }
        memcpy(h_pin, s_host, sizeof(double) * (size_t)J);
        HIP_TRY(hipMemcpyAsync(sdev, h_pin, sizeof(double) * (size_t)J, hipMemcpyHostToDevice, stream));
        const int g2 = vgrid();
        const int KS = std::max(1, std::min(ks_max, J / 8));
        // (fp32 basis: the combination is accumulated in fp64; everything after it -- normalisation, L y, Rayleigh
        // quotient, the reference's residual test -- is fp64 on the fp64 matrix)
        if (f32) k_ritz_partial<float><<<dim3(g2, KS), kBlock, 0, stream>>>(reinterpret_cast<const float*>(V), n, J, sdev, ypart);
        else k_ritz_partial<<<dim3(g2, KS), kBlock, 0, stream>>>(V, n, J, sdev, ypart);
        k_ritz_combine<<<g2, kBlock, 0, stream>>>(ypart, n, KS, y_raw, part_c);
        HIP_TRY(hipGetLastError());
        return check_vector(A, pl, rq, res_l1);
    }
    // Same, for a vector already in y_raw with its sums in part_c.
    int check_vector(const CsrView& A, const SpmvPlan& pl, double* rq, double* res_l1) {
        const int g2 = vgrid();
        OpLanczos op;
        op.L = check_view(pl);
        const bool via_panel = pan_live && OPT(panel_ops, 1) != 0;
        const int pa = via_panel ? vgrid() : pl.grid;            // alpha partials: one per workgroup of the kernel that finishes the rows
        if (via_panel) { op.L.P_a = pa; panel_spmv(y_raw, op, pa); }
        else launch_spmv(pl, stream, A, y_raw, op);
        // (the partials and the Rayleigh quotient go straight into mapped pinned memory: two copy kernels less per check)
        double* hp = h_pin + (vcap + 2);
        double* dhp = d_hpin + (vcap + 2);
        k_resid_l1<<<g2, kBlock, 0, stream>>>(w2, yvec, n, part_a2, pa, dhp, rq_dev, dhp + kMaxGrid, dhp + kMaxGrid + 8);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev1, stream));     // end of the solve's device time if this check passes (no second wait then)
        ++check_seq;
        if (after_check && spec_likely) { after_check(); hook_seq = check_seq; }
        HIP_TRY(hipStreamSynchronize(stream));
        ST_TRY(ipc_check_err("explicit check"));
        ev1_at_check = true;
        double s = 0.0, q4 = 0.0;
        for (int i = 0; i < g2; ++i) { s += hp[i]; q4 += hp[kMaxGrid + 8 + i]; }
        *res_l1 = s;
        *rq = hp[kMaxGrid];
        last_pr = q4 > 0.0 ? 1.0 / q4 : (double)n;      // participation ratio of the checked (unit) vector: ~ the number of vertices it lives on
        return MACHIP_OK;
    }

    struct Pending { int jend; hipEvent_t ev; int jstart; bool classic; int tailless; };   // tailless: step count of a chunk without tail kernel (0: it had one)

    // Spin on the pinned flag the tail kernel publishes; fall back to a stream query now and then so
    // a device fault cannot hang the host.
    int wait_flag(unsigned long long want) {
        volatile unsigned long long* f = h_flag;
        for (unsigned long spins = 0;; ++spins) {
            const unsigned long long v = *f;
            if ((v >> 32) == (want >> 32) && (v & 0xffffffffull) >= (want & 0xffffffffull)) return MACHIP_OK;
            if ((spins & 0xfffff) == 0xfffff) {
                const hipError_t q = hipStreamQuery(stream);
                if (q != hipSuccess && q != hipErrorNotReady)
                    return fail(MACHIP_HIP_ERROR, std::string("stream error while waiting for a Lanczos chunk: ") + hipGetErrorString(q));
                if (q == hipSuccess) {   // stream drained: the flag must be there now
                    const unsigned long long v2 = *f;
                    if ((v2 >> 32) == (want >> 32) && (v2 & 0xffffffffull) >= (want & 0xffffffffull)) return MACHIP_OK;
                    return fail(MACHIP_HIP_ERROR, "Lanczos chunk finished without publishing its flag");
                }
            }
            __builtin_ia32_pause();
        }
    }

    int get_event(hipEvent_t* e) {
        if (!ev_pool.empty()) { *e = ev_pool.back(); ev_pool.pop_back(); return MACHIP_OK; }
        HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        return MACHIP_OK;
    }

    // ---- preconditioned mode (precond.h) ------------------------------------------------------
    void drop_graphs() {
        for (auto& kv : graphs) for (hipGraphExec_t ge : kv.second) if (ge) (void)hipGraphExecDestroy(ge);
        graphs.clear();
    }
    int lob_alloc(long nnz) {
        if (!lob_ready) {
            ST_TRY(dev_alloc(&lx_x, n)); ST_TRY(dev_alloc(&lx_Lx, n)); ST_TRY(dev_alloc(&lx_p, n));
            ST_TRY(dev_alloc(&lx_Lp, n)); ST_TRY(dev_alloc(&lx_Lw, n));
            const bool jsave = lob_jacobi;
            lob_jacobi = false;
            const size_t tcap = std::max((size_t)lob_c() * (size_t)lob_stride(), (size_t)n);   // chunk-transposed, zero padded past n
            lob_jacobi = jsave;
            double** tr[] = {&lx_rT, &lx_wT, &lx_tl, &lx_tdinv, &lx_tcu, &lx_ba, &lx_bd, &lx_bu};
            for (double** q : tr) {
                ST_TRY(dev_alloc(q, tcap));
                HIP_TRY(hipMemsetAsync(*q, 0, sizeof(double) * tcap, stream));
            }
            if (n > kTriMaxN && n <= kTriBigMaxN) {
                const size_t fcap = (size_t)kTriThreads * (size_t)((n + kTriThreads - 1) / kTriThreads);
                ST_TRY(dev_alloc(&lx_ys, tcap)); ST_TRY(dev_alloc(&lx_pas, tcap));
                ST_TRY(dev_alloc(&lx_as, fcap)); ST_TRY(dev_alloc(&lx_bs, fcap));
                ST_TRY(dev_alloc(&lx_maps, 4 * (size_t)kMaxGrid));   // one map per workgroup and direction
            }
            ST_TRY(dev_alloc(&lx_part, (size_t)(2 * kLobNS + 1) * kMaxGrid)); ST_TRY(dev_alloc(&lx_partR, 2 * kMaxGrid));     // (two parities of the 15 sums + w^T L w per panel cell: k_lob_update_pan)
            ST_TRY(dev_alloc(&lx_bad, 1)); ST_TRY(dev_alloc(&lx_st, 1));
            HIP_TRY(hipHostMalloc((void**)&h_lrec, sizeof(double) * 4 * (size_t)(kLobCap + kLobMaxChunk + 4), hipHostMallocMapped));
            memset(h_lrec, 0, sizeof(double) * 4 * (size_t)(kLobCap + kLobMaxChunk + 4));
            HIP_TRY(hipHostGetDevicePointer((void**)&d_hlrec, h_lrec, 0));
            lob_ready = true;
        }
        if ((size_t)nnz > lx_colT_cap) {
            if (lx_colT) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(lx_colT); lx_colT = nullptr; drop_graphs(); }
            lx_colT_cap = (size_t)nnz + (size_t)nnz / 2 + 1024;
            ST_TRY(dev_alloc(&lx_colT, lx_colT_cap));
        }
        return MACHIP_OK;
    }
    // layout of the tridiagonal solver: up to n = 16 384 one workgroup, c = ceil(n/1024) unknowns per thread;
    // beyond, 4 unknowns per thread and as many 1024-thread workgroups as that takes
    bool lob_jacobi = false;   // the running preconditioned solve uses the diagonal preconditioner (natural layout, c = 1)
    bool lob_pan = false;      // ... and its product runs in column-panel form
    int lob_par0 = 0;          // parity of the iterate count at the start of the chunk being enqueued
    bool lob_pan2 = false;     // ... with two launches per iteration (k_lob_update_pan; MACHIP_LOB_PAN2=0: k_pan_find + k_lob_update)
    int lob_c() const { return lob_jacobi ? 1 : n > kTriMaxN ? kTriBigC : (n + kTriThreads - 1) / kTriThreads; }
    int lob_stride() const {
        if (lob_jacobi) return n;
        if (n <= kTriMaxN) return kTriThreads;
        const int q = (n + kTriBigC - 1) / kTriBigC;
        return (q + kTriThreads - 1) / kTriThreads * kTriThreads;
    }
    LobView lview(const SpmvPlan& pl) const {
        LobView L;
        L.n = n; L.c = lob_c(); L.stride = lob_stride();
        L.mapA = lx_maps; L.mapB = lx_maps ? lx_maps + kMaxGrid : nullptr;
        L.mapA2 = lx_maps ? lx_maps + 2 * kMaxGrid : nullptr; L.mapB2 = lx_maps ? lx_maps + 3 * kMaxGrid : nullptr;
        L.x = lx_x; L.Lx = lx_Lx; L.p = lx_p; L.Lp = lx_Lp; L.Lw = lx_Lw; L.rT = lx_rT; L.wT = lx_wT;
        L.tl = lx_tl; L.tdinv = lx_tdinv; L.tcu = lx_tcu; L.part = lx_part; L.partR = lx_partR;
        L.ys = lx_ys; L.pas = lx_pas;
        L.P_c = pl.grid; L.P_a = vgrid(); L.st = lx_st; L.hrec = d_hlrec; L.hflag = d_hflag;
        return L;
    }
    template <int CMAX>
    void lob_launch_chunk_t(const CsrView& AT, const SpmvPlan& pl, const LobView& L, int steps) {
        OpLob op;
        op.L = L;
        // Round 4: between two products the update of iteration s - 1, the tridiagonal solve of iteration s and g = U^T y run as ONE
        // single-workgroup launch (k_lob_fused; the same arithmetic per unknown): T W S (F W S)^(steps - 1) U instead of (T W S U)^steps
        // (small graphs only: with more unknowns per thread the one workgroup's strided reads of six vectors cost more than the two
        // launches saved -- city10000, C = 10, in the configs[4] sweep: 870 it/s fused against 1 223)
        const bool fuse = OPT(lob_fuse, 1) != 0 && CMAX <= 4;
        WbView W0 = wb_active;       // (s = 0 without the exact preconditioner: the fused kernel then skips g)
        for (int s = 0; s < steps; ++s) {
            if (s == 0 || !fuse) {
                k_tri_solve<CMAX><<<1, kTriThreads, 0, stream>>>(L, s);
                if (wb_active.s > 0) k_wb_g<<<(wb_active.s + kBlock - 1) / kBlock, kBlock, 0, stream>>>(L, wb_active);
            }
            if (wb_active.s > 0) {   // w <- y - Z C^-1 U^T y
                const WbView& W = wb_active;
                k_wb_h<<<std::min(kMaxGrid, (W.s + 3) / 4), kBlock, 0, stream>>>(W);
                k_wb_w<<<(int)std::min<size_t>(kMaxGrid, W.cap / kWbwRows), kWbwThreads, 0, stream>>>(L, W);
            }
            launch_spmv(pl, stream, AT, L.wT, op);
            if (fuse && s + 1 < steps) k_lob_fused<CMAX><<<1, kTriThreads, 0, stream>>>(L, W0, s);
            else k_lob_update<false><<<L.P_a, kBlock, 0, stream>>>(L, s);
        }
        k_lob_tail<<<1, 64, 0, stream>>>(L, steps);
    }
    void lob_launch_chunk(const CsrView& AT, const SpmvPlan& pl, const LobView& L, int steps) {
#ifdef MACHIP_EXPERIMENTS
        if (lob_jacobi) {          // diagonal preconditioner: two launches per iteration (three with the column-panel product)
            OpLob op;
            op.L = L;
            for (int s = 0; s < steps; ++s) {
                if (lob_pan && lob_pan2) {     // two launches per iteration: product (+ w^T L w per cell), then sums + Rayleigh-Ritz + update + next sums
                    launch_pan_mul_raw(L.wT);
                    const double* pw = lx_part + (size_t)2 * kLobNS * kMaxGrid;
                    if (pan.block2 == 512) k_lob_update_pan<512><<<pan.grid2, 512, 0, stream>>>(L, panv.ypart, panv.NP, pw, pan.NB * pan.NP, s, (lob_par0 + s + 1) & 1);
                    else if (pan.block2 == 1024) k_lob_update_pan<1024><<<pan.grid2, 1024, 0, stream>>>(L, panv.ypart, panv.NP, pw, pan.NB * pan.NP, s, (lob_par0 + s + 1) & 1);
                    else k_lob_update_pan<256><<<pan.grid2, 256, 0, stream>>>(L, panv.ypart, panv.NP, pw, pan.NB * pan.NP, s, (lob_par0 + s + 1) & 1);
                    continue;
                }
                if (lob_pan) {     // Lw = L w in column-panel form (panel.h: gathers served by LDS), then partial sums + inner products
                    launch_pan_mul_raw(L.wT);
                    k_pan_find<<<L.P_c, kBlock, 0, stream>>>(op, panv.ypart, panv.NP);
                } else launch_spmv(pl, stream, AT, L.wT, op);
                k_lob_update<true><<<L.P_a, kBlock, 0, stream>>>(L, s);
            }
            k_lob_tail<<<1, 64, 0, stream>>>(L, steps);
            return;
        }
#endif
        if (n > kTriMaxN) {
            OpLob op;
            op.L = L;
            const int gw = L.stride / kTriThreads;
            for (int s = 0; s < steps; ++s) {
                k_tri_big_fwd<<<gw, kTriThreads, 0, stream>>>(L);
                k_tri_big_mid<<<gw, kTriThreads, 0, stream>>>(L);
                k_tri_big_fin<<<gw, kTriThreads, 0, stream>>>(L);
                if (wb_active.s > 0) {
                    const WbView& W = wb_active;
                    k_wb_g<<<(W.s + kBlock - 1) / kBlock, kBlock, 0, stream>>>(L, W);
                    k_wb_h<<<std::min(kMaxGrid, (W.s + 3) / 4), kBlock, 0, stream>>>(W);
                    k_wb_w<<<(int)std::min<size_t>(kMaxGrid, W.cap / kWbwRows), kWbwThreads, 0, stream>>>(L, W);
                }
                launch_spmv(pl, stream, AT, L.wT, op);
                k_lob_update<false><<<L.P_a, kBlock, 0, stream>>>(L, s);
            }
            k_lob_tail<<<1, 64, 0, stream>>>(L, steps);
            return;
        }
        switch (L.c) {   // the solver kernel is instantiated per chunk length: no guards, everything in registers
#define MACHIP_LOB_CASE(C) case C: lob_launch_chunk_t<C>(AT, pl, L, steps); break;
            MACHIP_LOB_CASE(1) MACHIP_LOB_CASE(2) MACHIP_LOB_CASE(3) MACHIP_LOB_CASE(4) MACHIP_LOB_CASE(5) MACHIP_LOB_CASE(6)
            MACHIP_LOB_CASE(7) MACHIP_LOB_CASE(8) MACHIP_LOB_CASE(9) MACHIP_LOB_CASE(10) MACHIP_LOB_CASE(11) MACHIP_LOB_CASE(12)
            MACHIP_LOB_CASE(13) MACHIP_LOB_CASE(14) MACHIP_LOB_CASE(15)
#undef MACHIP_LOB_CASE
            default: lob_launch_chunk_t<16>(AT, pl, L, steps); break;
        }
    }
    int lob_enqueue_chunk(const CsrView& A, const CsrView& AT, const SpmvPlan& pl, const LobView& L, int steps) {
        if (!use_graph() || wb_active.s > 0) { lob_launch_chunk(AT, pl, L, steps); return MACHIP_OK; }   // (closure count is baked into the launches)
        if (graph_csr_key != (const void*)A.val) { drop_graphs(); graph_csr_key = (const void*)A.val; }
        const auto key = std::make_tuple(1000 + pl.variant, pl.width, pl.grid, L.c + (n > kTriMaxN ? 100 : 0) + (lob_jacobi ? 1000 : 0) + (lob_pan ? 2000 + 10000 * pan.NP + 1000000 * pan.NB : 0) + (lob_pan2 ? 500 + 250 * lob_par0 : 0), steps);
        auto it = graphs.find(key);
        if (it == graphs.end()) it = graphs.emplace(key, std::array<hipGraphExec_t, 2>{nullptr, nullptr}).first;
        hipGraphExec_t& ge = it->second[(size_t)(graph_flip++ & 1)];
        if (!ge) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            lob_launch_chunk(AT, pl, L, steps);
            HIP_TRY(hipStreamEndCapture(stream, &g));
            HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
        }
        HIP_TRY(hipGraphLaunch(ge, stream));
        return MACHIP_OK;
    }
    // Most closures the exact (Woodbury) preconditioner takes: the dense s x s factorisation grows as s^3 (rocSOLVER:
    // 7 ms at 2 048), its memory as s^2 + n s; MACHIP_WB_MAX raises the default for graphs that converge no other way.
    // Two tiers: up to wb_soft() closures the exact preconditioner is built at once; between wb_soft() and wb_hard() the
    // tridiagonal one runs first and the solve ESCALATES to the exact one only when it turns out to crawl (the dense
    // factorisation then costs tens of ms -- against seconds, or no convergence at all: fuzz seed 129, n = 36 874,
    // 3 900 active closures, lambda_2 / ||L|| = 1e-10: 200 000 iterations without converging -> 27 iterations, 36 ms).
    int wb_soft() const { return std::max(64, std::min(16384, OPT(wb_max, kWbMaxS))); }
    // (evaluation lanes: 4 096 -- an s x s inverse beyond that, 2 s^3 flops on the whole chip, starves the other lanes: city10000, nine
    // budgets on four lanes, 423 it/s with escalations up to 16 384 closures against 669 with the cap; profiles/r4_exact_small.md)
    int wb_hard() const { return std::max(wb_soft(), std::min(throughput_lane ? 4096 : 16384, OPT(wb_hard, 16384))); }   // (11 600 closures on 30 000 nodes at lambda_2/||L|| = 3e-9: 0.30 s escalated against 0.75-0.9 s)
    int wb_limit_now = kWbMaxS;   // the tier this solve_lob call may use
    bool lob_escalate = false;    // set by solve_lob when it gives up early in favour of the exact preconditioner
    int wb_cap_s = 0;      // what the buffers below were sized for
    int wb_alloc() {
        // sized for the closures this solve actually has (+25 %, whole 256s), not for the tier's limit: 16 384 would mean
        // a 2 GiB capacitance matrix held for the life of the handle (and of every evaluation lane)
        long need = support_hint > 0 ? support_hint + support_hint / 4 : 256;
        need = std::min<long>(wb_limit_now, std::max<long>(256, (need + 255) / 256 * 256));
        const int want = (int)need;
        if (wb_ui && wb_cap_s >= want) return MACHIP_OK;
        if (wb_ui) {
            HIP_TRY(hipStreamSynchronize(stream));
            drop_graphs();          // cached chunk graphs carry the old addresses
            void* old[] = {wb_ui, wb_uj, wb_counts, wb_uc, wb_g, wb_h, wb_Cm, wb_Cm2, wb_maps};
            for (void* q : old) if (q) (void)hipFree(q);
            wb_ui = wb_uj = wb_counts = nullptr; wb_uc = wb_g = wb_h = wb_Cm = wb_Cm2 = wb_maps = nullptr;
        }
        wb_cap_s = want;
        ST_TRY(dev_alloc(&wb_ui, (size_t)want)); ST_TRY(dev_alloc(&wb_uj, (size_t)want)); ST_TRY(dev_alloc(&wb_counts, 2));
        ST_TRY(dev_alloc(&wb_uc, (size_t)want)); ST_TRY(dev_alloc(&wb_g, (size_t)want)); ST_TRY(dev_alloc(&wb_h, (size_t)want));
        const size_t ldw = ((size_t)want + kGjT - 1) / kGjT * kGjT;      // whole 64 x 64 tiles (k_gj_step)
        ST_TRY(dev_alloc(&wb_Cm, ldw * ldw)); ST_TRY(dev_alloc(&wb_Cm2, ldw * ldw));
        return MACHIP_OK;
    }
    // Z = T^-1 U, C = D^-1 + U^T Z, C^-1 (rocSOLVER).  MACHIP_NOT_CONVERGED when C is not positive definite.
    int wb_build(const LobView& L, int s) {
        const size_t cap = (size_t)L.c * (size_t)L.stride;
        if (cap * (size_t)s > wb_Zt_cap) {
            if (wb_Zt) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(wb_Zt); wb_Zt = nullptr; }
            wb_Zt_cap = cap * (size_t)std::min(wb_cap_s, std::max(s + s / 2, 64));
            ST_TRY(dev_alloc(&wb_Zt, wb_Zt_cap));
        }
        WbView W;
        W.s = s; W.cap = cap; W.ui = wb_ui; W.uj = wb_uj; W.uc = wb_uc; W.Zt = wb_Zt; W.Cm = wb_Cm; W.g = wb_g; W.h = wb_h;
        W.ld = (s + kGjT - 1) / kGjT * kGjT;
        if (n > kTriMaxN) {   // batched multi-workgroup solves: grid = (workgroups per system, closures)
            const int gw = L.stride / kTriThreads;
            if (cap * (size_t)s > wb_pas_cap) {
                if (wb_pas) { HIP_TRY(hipStreamSynchronize(stream)); (void)hipFree(wb_pas); wb_pas = nullptr; }
                wb_pas_cap = wb_Zt_cap;
                ST_TRY(dev_alloc(&wb_pas, wb_pas_cap));
            }
            if (!wb_maps) ST_TRY(dev_alloc(&wb_maps, (size_t)4 * kMaxGrid * (size_t)wb_cap_s));
            WbBig Bg{wb_pas, wb_maps};
            k_wb_big_fwd<<<dim3(gw, s), kTriThreads, 0, stream>>>(L, W, Bg);
            k_wb_big_mid<<<dim3(gw, s), kTriThreads, 0, stream>>>(L, W, Bg);
            k_wb_big_fin<<<dim3(gw, s), kTriThreads, 0, stream>>>(L, W, Bg);
        } else switch (L.c) {
#define MACHIP_LOB_CASE(C) case C: k_wb_cols<C><<<s, kTriThreads, 0, stream>>>(L, W); break;
            MACHIP_LOB_CASE(1) MACHIP_LOB_CASE(2) MACHIP_LOB_CASE(3) MACHIP_LOB_CASE(4) MACHIP_LOB_CASE(5) MACHIP_LOB_CASE(6)
            MACHIP_LOB_CASE(7) MACHIP_LOB_CASE(8) MACHIP_LOB_CASE(9) MACHIP_LOB_CASE(10) MACHIP_LOB_CASE(11) MACHIP_LOB_CASE(12)
            MACHIP_LOB_CASE(13) MACHIP_LOB_CASE(14) MACHIP_LOB_CASE(15)
#undef MACHIP_LOB_CASE
            default: k_wb_cols<16><<<s, kTriThreads, 0, stream>>>(L, W); break;
        }
        const int g2s = (int)std::min<long>(kMaxGrid, ((long)W.ld * W.ld + kBlock - 1) / kBlock);
        k_wb_cap<<<g2s, kBlock, 0, stream>>>(L, W);
        // C^-1: ld / 32 blocked Gauss-Jordan steps on the matrix cores, ping-pong between the two buffers (woodbury.h)
        int* info = wb_counts + 1;
        HIP_TRY(hipMemsetAsync(info, 0, sizeof(int), stream));
        const int tiles = W.ld / kGjT;
        double *src = wb_Cm, *dst = wb_Cm2;
        // (from MACHIP_GJ_LOOK_MIN = 1 024 rows on, look-ahead: the workgroup holding the next pivot block inverts it for the next launch;
        // below that all tiles run in one round of workgroups and the redundant inversion is off nobody's critical path)
        const bool look = W.ld >= OPT(gj_look_min, 1024);
        if (look && !wb_piv) ST_TRY(dev_alloc(&wb_piv, (size_t)2 * kGjB * kGjB));
        for (int kb = 0, k = 0; kb < W.ld; kb += kGjB, ++k) {
            if (look) k_gj_step<0><<<dim3(tiles, tiles), 256, 0, stream>>>(src, dst, W.ld, kb, info, k ? wb_piv + (size_t)(k & 1) * kGjB * kGjB : nullptr,
                                                                        wb_piv + (size_t)((k + 1) & 1) * kGjB * kGjB);
            else k_gj_step<0><<<dim3(tiles, tiles), 256, 0, stream>>>(src, dst, W.ld, kb, info);
            std::swap(src, dst);
        }
        HIP_TRY(hipGetLastError());
        W.Cm = src;                                  // (the last step's output)
        int hinfo = 0;
        HIP_TRY(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (hinfo != 0) return MACHIP_NOT_CONVERGED;     // a non-positive pivot: C is not positive definite numerically
        wb_active = W;
        return MACHIP_OK;
    }

    // Returns MACHIP_OK (converged: yvec, *lam, *res set), MACHIP_NOT_CONVERGED (caller falls back to
    // Lanczos) or an error.
    int solve_lob(const CsrView& A, long nnz, double lnorm, double tol, int max_steps, int start_mode,
                  double* lam, double* res, long* iters, long* spmvs, long* restarts_out, bool jacobi = false) {
#ifndef MACHIP_EXPERIMENTS
        jacobi = false;
#endif
        lob_jacobi = jacobi;
        lob_pan = false;
        ST_TRY(lob_alloc(jacobi ? 0 : nnz));
        // (explicit-check kernels may use the whole chip at large n, cf. solve_lanczos)
        SpmvPlan pl = plan_spmv(opt, n, nnz, kAuto, jacobi && n > 32768 ? kMaxGrid : 0);
#ifdef MACHIP_EXPERIMENTS
        if (jacobi) {
            pan = plan_panel(opt, n, nnz, maxlen_hint, pan_allowed && precision == 0 && !shard && !ipc, (long)csr_cap, false, false);
            lob_pan = false; lob_pan2 = false;
            if (pan.on && !pan.verify) {
                ST_TRY(ensure_panel(A, nnz, pan));
                lob_pan = true;
                lob_pan2 = OPT(lob_pan2, 1) != 0 && pan.NB * pan.NP <= 256 && pan.grid2 <= 256 && pan.cells == 1;
            }
        }
#endif
        LobView L = lview(pl);
        if (lob_pan) L.P_c = std::min(256, vgrid());           // partial sums come from k_pan_find's workgroups
        if (lob_pan && lob_pan2) L.P_a = pan.grid2;            // ... ||r||_1 partials from k_lob_update_pan's (and k_lob_start's) pan.grid2 workgroups
        const int g2 = vgrid();
        const bool debug = OPT(debug, 0) != 0;
        const double scale = lnorm > 0 ? lnorm : 1.0;
        // ---- exact preconditioner (woodbury.h) when the graph is chain + at most kWbMaxS closures ----
        wb_active.s = 0;
        int wb_s = 0;
        lob_escalate = false;
        const bool wb_enabled = !jacobi && OPT(woodbury, 1) != 0 && chain_like && support_hint >= 0;
        const bool may_escalate = wb_enabled && support_hint > wb_limit_now && support_hint <= wb_hard();
        if (wb_enabled && support_hint <= wb_limit_now) {
            ST_TRY(wb_alloc());
            HIP_TRY(hipMemsetAsync(lx_bad, 0, sizeof(int), stream));
            k_wb_extract<<<1, kTriThreads, 0, stream>>>(A, (n + kTriThreads - 1) / kTriThreads, wb_cap_s, wb_ui, wb_uj, wb_uc, wb_counts, lx_bad);
            int hc[2] = {0, 0};
            HIP_TRY(hipMemcpyAsync(&hc[0], wb_counts, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(&hc[1], lx_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            if (hc[0] > wb_cap_s && hc[0] <= wb_limit_now && !hc[1]) {
                // more off-chain entries than the buffers were sized for (support_hint counts active CANDIDATES; fixed edges off
                // the chain are closures too): grow to what is there and extract again -- these are the stiff cases the exact
                // preconditioner exists for, it must not be skipped silently
                const long keep = support_hint;
                support_hint = hc[0];
                const int st_grow = wb_alloc();
                support_hint = keep;
                if (st_grow != MACHIP_OK) return st_grow;
                HIP_TRY(hipMemsetAsync(lx_bad, 0, sizeof(int), stream));
                k_wb_extract<<<1, kTriThreads, 0, stream>>>(A, (n + kTriThreads - 1) / kTriThreads, wb_cap_s, wb_ui, wb_uj, wb_uc, wb_counts, lx_bad);
                HIP_TRY(hipMemcpyAsync(&hc[0], wb_counts, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipMemcpyAsync(&hc[1], lx_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
            }
            if (hc[0] > 0 && hc[0] <= wb_cap_s && !hc[1]) wb_s = hc[0];
        }
        const int chain_only = wb_s > 0 ? 1 : 0;
        // ---- T = tridiag(L) + sigma I (chain Laplacian + sigma I under the exact preconditioner), factored on
        // the device; gather indices in the solver's layout ----
        const double sigma = (wb_s > 0 ? 1e-8 : 2.5e-7) * scale;
        HIP_TRY(hipMemsetAsync(lx_bad, 0, sizeof(int), stream));
#ifdef MACHIP_EXPERIMENTS
        if (jacobi) k_jac_dinv<<<g2, kBlock, 0, stream>>>(A, lx_tdinv, lx_bad);
        else
#endif
        if (n > kTriMaxN) k_tri_factor_big<<<1, kTriThreads, 0, stream>>>(A, L.stride, sigma, lx_tl, lx_tdinv, lx_tcu, lx_as, lx_bs, lx_bad, chain_only);
        else {
            k_tri_band<<<g2, kBlock, 0, stream>>>(A, L.c, L.stride, lx_ba, lx_bd, lx_bu);
            switch (L.c) {
#define MACHIP_LOB_CASE(C) case C: k_tri_factor<C><<<1, kTriThreads, 0, stream>>>(n, lx_ba, lx_bd, lx_bu, sigma, lx_tl, lx_tdinv, lx_tcu, lx_bad, chain_only); break;
            MACHIP_LOB_CASE(1) MACHIP_LOB_CASE(2) MACHIP_LOB_CASE(3) MACHIP_LOB_CASE(4) MACHIP_LOB_CASE(5) MACHIP_LOB_CASE(6)
            MACHIP_LOB_CASE(7) MACHIP_LOB_CASE(8) MACHIP_LOB_CASE(9) MACHIP_LOB_CASE(10) MACHIP_LOB_CASE(11) MACHIP_LOB_CASE(12)
            MACHIP_LOB_CASE(13) MACHIP_LOB_CASE(14) MACHIP_LOB_CASE(15)
#undef MACHIP_LOB_CASE
            default: k_tri_factor<16><<<1, kTriThreads, 0, stream>>>(n, lx_ba, lx_bd, lx_bu, sigma, lx_tl, lx_tdinv, lx_tcu, lx_bad, chain_only); break;
            }
        }
        CsrView AT = A;
        if (!jacobi) {
            k_lob_perm_cols<<<(int)std::min<long>(kMaxGrid, (nnz + kBlock - 1) / kBlock), kBlock, 0, stream>>>(A.col, nnz, L.c, L.stride, lx_colT);
            AT.col = lx_colT;
        }
        if (wb_s > 0) {
            const int st = wb_build(L, wb_s);
            if (st != MACHIP_OK && st != MACHIP_NOT_CONVERGED) return st;
            if (st == MACHIP_NOT_CONVERGED) return MACHIP_NOT_CONVERGED;   // capacitance not SPD numerically: Lanczos takes over
        }
        // ---- start vector: normalised into yvec, w2 = L yvec, Rayleigh quotient ----
        const double* src = (start_mode == 1 && have_prev) ? yvec : (have_start ? start : nullptr);
        if (!src) { k_fill_start<<<g2, kBlock, 0, stream>>>(u, n, 0x1234567ull); src = u; }
        HIP_TRY(hipMemcpyAsync(y_raw, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        k_vec_sums<<<g2, kBlock, 0, stream>>>(y_raw, n, part_c);
        double rq = 0.0, r1 = 0.0;
        ST_TRY(check_vector(A, pl, &rq, &r1));
        *spmvs = 1; *iters = 0; *restarts_out = 0;
        int hbad = 0;
        HIP_TRY(hipMemcpyAsync(&hbad, lx_bad, sizeof(int), hipMemcpyDeviceToHost, stream));   // (never the legacy stream: another lane's thread may be capturing a graph)
        HIP_TRY(hipStreamSynchronize(stream));
        if (hbad || !(rq == rq)) return MACHIP_NOT_CONVERGED;
        *lam = rq; *res = r1 / scale;
        if (*res < tol) return MACHIP_OK;
        const int cap = std::min(kLobCap, max_steps > 0 ? max_steps : kLobCap);
        const int patience = std::max(64, OPT(lob_patience, 10000));   // iterations without a new best residual
        // (exact preconditioner: a handful of iterations in all, each six launches -- short chunks, no speculation)
        const bool wb = wb_active.s > 0;
        const int chunk0 = wb ? 4 : std::min(kLobMaxChunk, std::max(1, OPT(lob_chunk, 16)));
        const double ltarget = std::log(std::max(tol * scale, 1e-300));
        int it_enq = 0, restarts = 0;
        double best = r1;
        int best_it = 0;
        while (true) {
            ++epoch;
#ifdef MACHIP_EXPERIMENTS
            if (jacobi && lob_pan && lob_pan2) k_lob_start<true, true><<<pan.grid2, kBlock, 0, stream>>>(L, yvec, w2, rq_dev, it_enq, epoch);     // (same grid as k_lob_update_pan: its prologue counts gridDim partials)
            else if (jacobi) k_lob_start<true><<<g2, kBlock, 0, stream>>>(L, yvec, w2, rq_dev, it_enq, epoch);
            else
#endif
            k_lob_start<false><<<g2, kBlock, 0, stream>>>(L, yvec, w2, rq_dev, it_enq, epoch);
            std::deque<int> pend;
            std::deque<std::pair<int, double>> hist;
            double to_go = 1e18, est = 1e300;
            bool check = false, bad = false;
            int ramp = wb ? 2 : 4;
            while (!check) {
                const bool near = to_go < 2.0 * chunk0;
                const int depth = (near || wb) ? 1 : 2;
                while ((int)pend.size() < depth && it_enq < cap) {
                    int chunk = std::min(chunk0, ramp);   // easy problems (T ~ L) converge in a handful of iterations
                    ramp = std::min(chunk0, ramp * 2);
                    if (near) chunk = std::min(chunk, std::max(2, (int)(0.75 * to_go) + 1));
                    chunk = std::min(chunk, cap - it_enq);
                    lob_par0 = it_enq & 1;      // (parity of the chunk's first iterate: baked into k_lob_update_pan's launches and the graph key)
                    ST_TRY(lob_enqueue_chunk(A, AT, pl, L, chunk));
                    it_enq += chunk;
                    *spmvs += chunk;
                    pend.push_back(it_enq);
                }
                if (pend.empty()) { check = true; break; }
                int jend = pend.front();
                pend.pop_front();
                ST_TRY(wait_flag(((unsigned long long)epoch << 32) | (unsigned long long)(unsigned int)jend));
                const unsigned long long fv = *(volatile unsigned long long*)h_flag;
                bad = (fv & 0x80000000ull) != 0;
                {   // the flag can overtake the record on its way to host memory (seen once in ~60 solves as a
                    // stale residual that triggered premature checks): wait for the record's own tag
                    volatile double* rec = h_lrec + 4 * (size_t)jend;
                    const double want = lob_tag(epoch, jend);
                    for (unsigned long spins = 0; rec[2] != want; ++spins) {
                        if (spins > 200000000ul) return fail(MACHIP_HIP_ERROR, "preconditioned solve: iteration record never arrived");
                        __builtin_ia32_pause();
                    }
                    est = rec[1];
                }
                if (debug) fprintf(stderr, "[machip] lobpcg it=%d theta=%.15g est=%.3e to_go=%.0f bad=%d pend=%zu\n", jend, h_lrec[4 * (size_t)jend], est / scale, std::min(to_go, 1e9), (int)bad, pend.size());
                if (est < best) { best = est; best_it = jend; }
                if (est > 0.0) {
                    hist.emplace_back(jend, std::log(est));
                    while (hist.size() > 2 && hist[1].first <= jend - 48) hist.pop_front();
                    to_go = 1e18;
                    if (hist.front().first < jend) {
                        const double slope = (hist.front().second - hist.back().second) / (double)(jend - hist.front().first);
                        if (slope > 1e-7) to_go = std::max(0.0, (hist.back().second - ltarget) / slope);
                    }
                }
                if (bad || !(est == est) || est < tol * scale || jend >= cap || jend - best_it > patience) check = true;
                // the tridiagonal preconditioner crawls and an exact one is affordable: hand back for the second tier
                if (!wb && may_escalate && !check && ((jend >= 1500 && to_go > 3000.0) || jend >= 6000)) {
                    HIP_TRY(hipStreamSynchronize(stream));
                    *iters = it_enq; *restarts_out = restarts;
                    lob_escalate = true;
                    if (debug) fprintf(stderr, "[machip] lobpcg it=%d: escalating to the exact preconditioner (to_go %.0f)\n", jend, std::min(to_go, 1e9));
                    return MACHIP_NOT_CONVERGED;
                }
            }
            // ---- explicit check of the current x (fresh SpMV), also the refresh point of a restart ----
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpyAsync(y_raw, lx_x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            k_vec_sums<<<g2, kBlock, 0, stream>>>(y_raw, n, part_c);
            ST_TRY(check_vector(A, pl, &rq, &r1));
            *spmvs += 1;
            *iters = it_enq;
            *restarts_out = restarts;
            if (debug) fprintf(stderr, "[machip]    lobpcg check it=%d rq=%.15g res=%.3e (tol %.1e)\n", it_enq, rq, r1 / scale, tol);
            if (!(rq == rq)) return MACHIP_NOT_CONVERGED;
            *lam = rq; *res = r1 / scale;
            if (*res < tol) return MACHIP_OK;
            if (it_enq >= cap || ++restarts > 12 || it_enq - best_it > patience) return MACHIP_NOT_CONVERGED;
        }
    }

#ifdef MACHIP_EXPERIMENTS      // (measured: ties with the scalar recurrence on city10000 -- profiles/r5_city_block.md)
    // ---- block Lanczos mode (blocklan.h, band.h) --------------------------------------------------------------------------------
    static constexpr int kBlockFallback = 1001;       // internal status of solve_block: the scalar recurrence takes over
    BRec* bZ0 = nullptr; BRec* bZ1 = nullptr;
    double* bpart = nullptr;      // 2 x kBQ x kMaxGrid partial sums
    double* bU0 = nullptr;        // start block, column-major n x 4
    double* bwarm = nullptr;      // Ritz vectors 2..4 of the last block solve (n x 3): columns 1..3 of the next start block
    double* bsdev = nullptr;      // their coefficient vectors (3 x (vcap + 2))
    double* h_bs = nullptr;       // ... pinned staging
    double* brec = nullptr;       // records in device memory
    double* h_brec = nullptr; double* d_hbrec = nullptr;      // records, kBRec doubles per block step (pinned, device-mapped)
    size_t brec_steps = 0;
    bool blk_have_warm = false;
    long long* bclk = nullptr;    // (probe)
    long hist_blk_steps = -1;     // block steps of the last block solve (counts only: the mode choice stays reproducible)
    band::Factor bfac; band::Smallest bsm; std::vector<double> bh, bguess, bwk;

    int blk_G(long nnz) const { const long mean = nnz / std::max(1, n); return mean < 5 ? 2 : mean < 12 ? 4 : 8; }
    int blk_threads(long) const { return kBlkThreads; }
    int blk_rpb(long nnz) const { return (blk_threads(nnz) - 64) / blk_G(nnz); }
    bool blk_fits(long nnz) const {
        const int rpb = blk_rpb(nnz);
        return n > 256 && (n + rpb - 1) / rpb <= kBlkMaxGrid && vcap >= 256 && precision == 0 && !shard && !ipc;
    }
    BlkView bview(int grid) const {
        BlkView L; L.n = n; L.st = st; L.Z0 = bZ0; L.Z1 = bZ1; L.V = V; L.rec = brec; L.hrec = d_hbrec; L.hflag = d_hflag; L.part = bpart; L.P = grid; L.clk = bclk; L.inv_n = 1.0 / (double)n;
        return L;
    }
    int blk_alloc() {
        if (bZ0) return MACHIP_OK;
        ST_TRY(dev_alloc(&bZ0, (size_t)n)); ST_TRY(dev_alloc(&bZ1, (size_t)n));
        ST_TRY(dev_alloc(&bpart, (size_t)2 * kBQ * kMaxGrid));
        ST_TRY(dev_alloc(&bU0, (size_t)n * kBW)); ST_TRY(dev_alloc(&bwarm, (size_t)n * (kBW - 1)));
        ST_TRY(dev_alloc(&bsdev, (size_t)(kBW - 1) * (vcap + 2)));
        brec_steps = vcap / kBW + 2;
        ST_TRY(dev_alloc(&brec, brec_steps * kBRec));
        HIP_TRY(hipHostMalloc((void**)&h_brec, brec_steps * kBRec * sizeof(double), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&d_hbrec, h_brec, 0));
        HIP_TRY(hipHostMalloc((void**)&h_bs, (size_t)(kBW - 1) * (vcap + 2) * sizeof(double), 0));
        return MACHIP_OK;
    }
    void blk_launch_chunk(const CsrView& A, int G, int threads, int grid, int steps) {
        const BlkView L = bview(grid);
        for (int s = 0; s < steps; ++s) {
            if (G == 2) k_blk_vec<2, kBlkThreads><<<grid, threads, 0, stream>>>(A, L, s);
            else if (G == 4) k_blk_vec<4, kBlkThreads><<<grid, threads, 0, stream>>>(A, L, s);
            else k_blk_vec<8, kBlkThreads><<<grid, threads, 0, stream>>>(A, L, s);
        }
        k_blk_tail<<<1, 64, 0, stream>>>(L, steps);
    }
    int blk_enqueue_chunk(const CsrView& A, int G, int threads, int grid, int steps) {
        if (!use_graph()) { blk_launch_chunk(A, G, threads, grid, steps); HIP_TRY(hipGetLastError()); return MACHIP_OK; }
        if (graph_csr_key != (const void*)A.val) {   // different matrix buffers: cached graphs are stale
            for (auto& kv : graphs) for (hipGraphExec_t ge : kv.second) if (ge) (void)hipGraphExecDestroy(ge);
            graphs.clear();
            graph_csr_key = (const void*)A.val;
        }
        const auto key = std::make_tuple(9000 + G, 0, grid, threads, steps);
        auto it = graphs.find(key);
        if (it == graphs.end()) it = graphs.emplace(key, std::array<hipGraphExec_t, 2>{nullptr, nullptr}).first;
        hipGraphExec_t& ge = it->second[(size_t)(graph_flip++ & 1)];
        if (!ge) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            blk_launch_chunk(A, G, threads, grid, steps);
            HIP_TRY(hipStreamEndCapture(stream, &g));
            HIP_TRY(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
        }
        HIP_TRY(hipGraphLaunch(ge, stream));
        return MACHIP_OK;
    }

    // Block Lanczos solve.  Returns MACHIP_OK (pair in yvec / *lambda2, passed the explicit check), kBlockFallback (breakdown of the
    // block, basis full, or no convergence within the step budget: the caller continues with the scalar recurrence), or an error.
    int solve_block(const CsrView& A, long nnz, double lnorm, double tol, int max_steps, int start_mode, double* lambda2,
                    machip_solve_stats* stats) {
        ST_TRY(blk_alloc());
        if (OPT(debug, 0) == 2 && !bclk) { HIP_TRY(hipMalloc((void**)&bclk, sizeof(long long) * 16 * 256)); HIP_TRY(hipMemset(bclk, 0, sizeof(long long) * 16 * 256)); }
        const SpmvPlan pl = plan_spmv(opt, n, nnz, 0, n > 32768 ? kMaxGrid : 0);
        const int G = blk_G(nnz), threads = blk_threads(nnz), rpb = blk_rpb(nnz), grid = (n + rpb - 1) / rpb, g2 = vgrid();
        const bool debug = OPT(debug, 0) != 0;
        const double tiny_l = (lnorm > 0 ? lnorm : 1.0);
        HIP_TRY(hipEventRecord(ev0, stream));
        ev1_at_check = false;
        // ---- start block: column 0 as the scalar recurrence would start, the others from the last block solve's Ritz vectors ----
        if (start_mode == 1 && have_prev) HIP_TRY(hipMemcpyAsync(bU0, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        else if (have_start) HIP_TRY(hipMemcpyAsync(bU0, start, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        else k_fill_start<<<g2, kBlock, 0, stream>>>(bU0, n, 0x1234567ull);
        if (blk_have_warm && start_mode == 1) HIP_TRY(hipMemcpyAsync(bU0 + (size_t)n, bwarm, sizeof(double) * (size_t)n * (kBW - 1), hipMemcpyDeviceToDevice, stream));
        else for (int c = 1; c < kBW; ++c) k_fill_start<<<g2, kBlock, 0, stream>>>(bU0 + (size_t)c * n, n, 0x1234567ull + 0x9E37ull * (unsigned long long)c);
        ++epoch;
        k_blk_init<<<grid, kBlock, 0, stream>>>(bview(grid), bU0, (int)epoch);
        HIP_TRY(hipGetLastError());
        const int cap_steps = (int)std::min<size_t>(std::min<size_t>((vcap - 2) / kBW, brec_steps - 2), (size_t)std::max(8, (n - 1) / kBW)) & ~1;
        const int chunk0 = std::max(2, OPT(blk_chunk, 16) & ~1), chunk_near = std::max(2, OPT(blk_chunk_near, 4) & ~1);
        const double trigger_slack = 0.01 * OPT(trigger_pct, 110);
        if (max_steps <= 0) max_steps = 200000;
        std::deque<std::pair<int, int>> bp;       // (jstart, jend) of the chunks in flight
        std::deque<std::pair<int, double>> hist;
        int J_enq = 0, J = 0;
        double to_go = 1e18, est_latest = 1e300, last_check_est = 1e300, theta_prev = 0.0, lam = 0.0, res = 0.0;
        bguess.clear();
        const double qnan = std::numeric_limits<double>::quiet_NaN();
        const double ltarget = std::log(std::max(tol * tiny_l, 1e-300));
        bool converged = false, fallback = false;
        long checks = 0;
        HIP_TRY(hipEventRecord(evs0, stream));
        while (!converged && !fallback) {
            const bool near = to_go < 2.0 * chunk0 || (to_go >= 1e17 && est_latest < 1e3 * tol * lnorm);
            const int depth = near ? 1 : 2;
            while ((int)bp.size() < depth && J_enq < cap_steps && J_enq < max_steps) {
                int chunk = near ? chunk_near : chunk0;
                if (near && to_go < 1e17) chunk = std::min(chunk0, std::max(chunk_near, ((int)(0.75 * to_go) + 1) & ~1));
                chunk = std::min(chunk, cap_steps - J_enq);
                if (chunk <= 0) break;
                const int hi = J_enq + chunk;
                // poison what this chunk delivers for the first time: B_j for j in (J_enq, hi] (and B_0), A_j / l1_j for j in [J_enq, hi)
                for (int j = J_enq ? J_enq + 1 : 0; j <= hi; ++j) for (int i = 0; i < 16; ++i) h_brec[(size_t)j * kBRec + 16 + i] = qnan;
                for (int j = J_enq; j < hi; ++j) { for (int i = 0; i < 16; ++i) h_brec[(size_t)j * kBRec + i] = qnan; for (int i = 0; i < 4; ++i) h_brec[(size_t)j * kBRec + 32 + i] = qnan; }
                ST_TRY(blk_enqueue_chunk(A, G, threads, grid, chunk));
                if (debug) fprintf(stderr, "[machip] block enqueue J=%d chunk=%d depth=%d\n", J_enq, chunk, depth);
                bp.emplace_back(J_enq, hi);
                J_enq = hi;
            }
            if (bp.empty()) { fallback = true; break; }
            const std::pair<int, int> p = bp.front();
            bp.pop_front();
            ST_TRY(wait_flag(((unsigned long long)epoch << 32) | (unsigned long long)(unsigned int)p.second));
            {
                unsigned long budget = 5000000ul;
                auto wait_slot = [&](size_t idx) { volatile double* slot = h_brec + idx; while (*slot != *slot && budget) { --budget; __builtin_ia32_pause(); } };
                for (int j = p.first ? p.first + 1 : 0; j <= p.second && budget; ++j) for (int i = 0; i < 16; ++i) wait_slot((size_t)j * kBRec + 16 + i);
                for (int j = p.first; j < p.second && budget; ++j) { for (int i = 0; i < 16; ++i) wait_slot((size_t)j * kBRec + i); for (int i = 0; i < 4; ++i) wait_slot((size_t)j * kBRec + 32 + i); }
                if (!budget) { fallback = true; break; }         // (a genuine NaN: the scalar path reports it)
            }
            J = p.second;
            // ---- breakdown: a vanishing pivot of some B_j ----
            bool broke = false;
            for (int j = 0; j <= J && !broke; ++j) for (int c = 0; c < kBW; ++c) if (!(h_brec[(size_t)j * kBRec + 16 + c * 5] > 0.0)) { broke = true; break; }
            if (broke) { if (debug) fprintf(stderr, "[machip] block J=%d: breakdown\n", J); fallback = true; break; }
            // ---- banded matrix of the J blocks, its smallest pair ----
            const int N = J * kBW;
            bh.assign((size_t)(kBW + 1) * N, 0.0);
            for (int j = 0; j < J; ++j) {
                const double* Aj = h_brec + (size_t)j * kBRec;
                for (int c = 0; c < kBW; ++c) for (int r = c; r < kBW; ++r) bh[(size_t)(r - c) * N + 4 * j + c] = Aj[r * 4 + c];
                if (j + 1 < J) {
                    const double* Bn = h_brec + (size_t)(j + 1) * kBRec + 16;
                    for (int r = 0; r < kBW; ++r) for (int c = r; c < kBW; ++c) bh[(size_t)(kBW + r - c) * N + 4 * j + c] = Bn[r * 4 + c];
                }
            }
            band::smallest_eigpair(bh.data(), N, kBW, bguess.data(), (int)bguess.size(), theta_prev, bsm, bfac, bwk,
                                   /*rough=*/est_latest > 1e4 * tol * lnorm);
            bguess = bsm.s; theta_prev = bsm.theta;
            const double* BJ = h_brec + (size_t)J * kBRec + 16;
            const double* l1 = h_brec + (size_t)(J - 1) * kBRec + 32;
            double est = 0.0;
            for (int r = 0; r < kBW; ++r) {
                double rho = 0.0;
                for (int c = r; c < kBW; ++c) rho += BJ[r * 4 + c] * bsm.s[(size_t)N - kBW + c];
                est += std::fabs(rho) * (l1[r] > 0 ? l1[r] : std::sqrt((double)n));
            }
            est_latest = est;
            if (est > 0.0) {
                hist.emplace_back(J, std::log(est));
                while (hist.size() > 2 && hist[1].first <= J - 32) hist.pop_front();
                to_go = 1e18;
                if (hist.front().first < J) {
                    const double slope = (hist.front().second - hist.back().second) / (double)(J - hist.front().first);
                    if (slope > 1e-7) to_go = std::max(0.0, (hist.back().second - ltarget) / slope);
                }
            }
            if (debug) fprintf(stderr, "[machip] block J=%d theta=%.15g est=%.3e to_go=%.0f pend=%zu fact=%d\n", J, bsm.theta, est / tiny_l, std::min(to_go, 1e9), bp.size(), bsm.factorisations);
            const bool at_cap = J >= cap_steps || J >= max_steps;
            const bool trig = est < trigger_slack * tol * lnorm;
            if ((trig && est < 0.5 * last_check_est) || at_cap) {
                double rq = 0.0, r1 = 0.0;
                HIP_TRY(hipEventRecord(evs1, stream));
                spec_likely = est < 0.01 * OPT(spec_slack_pct, 105) * tol * lnorm;
                ST_TRY(explicit_check(A, pl, N, bsm.s.data(), &rq, &r1, false));      // syncs the stream
                spec_likely = true;
                ++checks;
                last_check_est = std::max(est, 1e-300);
                lam = rq; res = lnorm > 0 ? r1 / lnorm : r1;
                if (debug) fprintf(stderr, "[machip]    block check J=%d rq=%.15g res=%.3e (tol %.1e)\n", J, rq, res, tol);
                if (res < tol) { converged = true; final_check_seq = check_seq; break; }
                if (at_cap) { fallback = true; break; }
            }
        }
        if (!bp.empty()) HIP_TRY(hipStreamSynchronize(stream));
        bp.clear();
        if (bclk) {
            std::vector<long long> hc(16 * 256);
            HIP_TRY(hipMemcpy(hc.data(), bclk, sizeof(long long) * hc.size(), hipMemcpyDeviceToHost));
            for (int b : {0, grid / 2, grid - 1}) {
                const long long* c = hc.data() + 16 * b; const long long t0 = std::min(c[0], c[8]);
                fprintf(stderr, "[machip] blk clocks wg %d (x10 ns from kernel start): prologue loads %lld tot %lld G %lld chol %lld Ri %lld published %lld wave0 at barrier %lld | rows start %lld gathered %lld past barrier %lld finished %lld end %lld\n", b, c[1] - t0, c[2] - t0, c[5] - t0, c[6] - t0, c[3] - t0, c[7] - t0, c[4] - t0, c[8] - t0, c[9] - t0, c[10] - t0, c[11] - t0, c[12] - t0);
            }
        }
        if (fallback || !converged) {
            if (debug) fprintf(stderr, "[machip] block mode gives up at J=%d: the scalar recurrence takes over\n", J);
            hist_blk_steps = 1l << 40;          // (never again for this handle unless a caller forces it)
            return kBlockFallback;
        }
        // ---- start block of the next solve: the Ritz vectors behind the next three Ritz values (good start vectors is all they
        // have to be: a few sweeps of subspace iteration) ----
        {
            const int N = J * kBW;
            std::vector<double> th, S;
            band::lowest_block(bh.data(), N, kBW, kBW, bsm.theta, bsm.s.data(), th, S, bfac, 3);
            const int KS = std::max(1, std::min(ks_max, N / 8));
            for (int c = 1; c < kBW; ++c) memcpy(h_bs + (size_t)(c - 1) * (vcap + 2), S.data() + (size_t)c * N, sizeof(double) * (size_t)N);
            HIP_TRY(hipMemcpyAsync(bsdev, h_bs, sizeof(double) * (size_t)(kBW - 1) * (vcap + 2), hipMemcpyHostToDevice, stream));
            for (int c = 1; c < kBW; ++c) {
                k_ritz_partial<<<dim3(g2, KS), kBlock, 0, stream>>>(V, n, N, bsdev + (size_t)(c - 1) * (vcap + 2), ypart);
                k_ritz_combine<<<g2, kBlock, 0, stream>>>(ypart, n, KS, bwarm + (size_t)(c - 1) * n, part_c);
            }
            HIP_TRY(hipGetLastError());
            blk_have_warm = true;
        }
        have_prev = true; last_was_lob = true; J_last = 0; last_seq_sharded = false; last_seq_f32 = false;
        last_steps = J_enq; last_steps_lowp = 0; hist_blk_steps = J_enq;
        last_mode = 9;
        float ms = 0.f, sms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        HIP_TRY(hipEventElapsedTime(&sms, evs0, evs1));
        *lambda2 = lam;
        if (stats) {
            stats->lanczos_steps = J_enq; stats->spmv_total = (long)J_enq * kBW + checks; stats->vec_passes = (long)J_enq * 7 * kBW;
            stats->restarts = 0; stats->nnz = nnz; stats->residual = res; stats->lnorm = lnorm; stats->gpu_ms = ms;
            stats->step_ms = sms; stats->steps_timed = J_enq; stats->steps_lowp = 0;
        }
        if (lam < 1e-12 * tiny_l) return fail(MACHIP_DISCONNECTED, "lambda_2 ~ 0: the graph is not connected");
        return MACHIP_OK;
    }

#endif

    // start_mode: 0 = stored cold-start vector (or device pseudo-random if none), 1 = previous
    // Fiedler vector (warm start).
    int solve(const CsrView& A, long nnz, double lnorm, double tol, int max_steps, int start_mode,
              int forced_variant, double* lambda2, machip_solve_stats* stats) {
        int mode = opt.is_set(kOpt_solver) ? OPT(solver, 0) : solver_mode;      // (option "solver" overrides machip_set_solver: 0 auto, 1 Lanczos, 2 preconditioned)
#ifndef MACHIP_EXPERIMENTS
        if (mode == 3) mode = 0;      // (3 = diagonally preconditioned LOBPCG: experiments build only)
#endif
        // auto: chain-dominated graphs with few active closures per node (measured cross-over)
        const bool eligible = n > 256 && n <= kTriBigMaxN;
        // One preconditioned iteration costs about `ratio` Lanczos steps (three launches, one of them a
        // single workgroup).  Sparse closures: preconditioned.  Denser closures: only when the last
        // Lanczos solve of this problem was long (a stiff x) and the preconditioned mode has not been seen
        // to need more than 1/ratio of those steps.  Counts only -- never timings -- so the choice, and
        // with it every rounding of the trajectory, is reproducible run to run.
        // (where the single-workgroup Lanczos applies a step costs ~2 us instead of ~4.5: higher bar there --
        // measured: intel at 9 % closures/node 3.5 ms Lanczos vs 4.7 ms preconditioned, kitti_05 at 0.5 %: 8.6 vs 1.1)
        const bool small = n <= kPersistThreads * kPersistMaxRows;
        const long ratio = small ? 9 : 6;
        const bool sparse = (double)support_hint <= OPT(lob_density_pct, small ? 3 : 12) * 0.01 * (double)n;
        // The preconditioned mode is a block-size-1 LOBPCG preconditioned by the odometry chain: it is built for, and
        // has only been validated on (fuzz runs, every pose graph), chain-DOMINATED graphs.  On a dense random graph
        // the chain is no preconditioner at all (config 2, selected there by a mis-learnt step count: 44 us x 60-220
        // iterations against 1.1 ms of Lanczos), and a single-vector iteration has no guard against settling on a
        // higher eigenpair of a clustered spectrum, which the reference's residual test cannot tell apart.  So it is
        // confined to at most two active closures per node -- also when a caller forces it; denser graphs always take
        // the Lanczos path, which sees the whole Krylov space.
        const bool chain_dominated = support_hint < 0 || support_hint <= 2 * (long)n;
        const bool stiff = hist_lan_steps > 2500 && (hist_lob_iters < 0 || hist_lob_iters * ratio < hist_lan_steps);
        const bool slow_lob = hist_lan_steps > 0 && hist_lob_iters > 0 && hist_lob_iters * ratio > 2 * hist_lan_steps;
        // Round 4: with the hand-written inverse (woodbury.h: 0.08 ms at 157 closures, 0.28 ms at 645, where rocSOLVER took
        // 0.3-1.25 ms) and the parallel n x s product, the EXACT chain + closures preconditioner beats the single-workgroup Lanczos
        // on small pose graphs up to several hundred active closures (intel, 157-645 closures: 0.90 against 1.60 ms per solve,
        // 12 iterations against 830 steps).  Cost model in launch equivalents (~4.7 us), counts only: Lanczos = steps x 0.4
        // (single-workgroup step 1.9 us); exact = 60 (set-up, checks) + 2.7 per 32 closures (one inverse launch) + 7.5 per
        // iteration.  Without a Lanczos history the closure count alone decides (MACHIP_LOB_SMALL_S).
        const long lob_it_guess = hist_lob_iters > 0 ? hist_lob_iters : 14;
        const double cost_lob = 60.0 + 2.7 * ((double)support_hint / 32.0) + 7.5 * (double)lob_it_guess;
        // (Not on evaluation lanes: there 16 single-CU Lanczos solves run side by side -- intel sweep 4 670 it/s aggregate against
        // 2 460 when every lane launches the exact mode's chip-wide kernels; profiles/r4_exact_small.txt.)
        const bool exact_small = small && !throughput_lane && precision == 0 && support_hint >= 0 && support_hint <= OPT(lob_small_s, 700) &&
                                 support_hint <= wb_soft() && OPT(woodbury, 1) != 0 &&
                                 (hist_lan_steps <= 0 || cost_lob < 0.4 * (double)hist_lan_steps);
        // Beyond the single-workgroup sizes (n <= 16 384: the one-workgroup tridiagonal kernels the figures below were measured with)
        // the same choice needs a Lanczos history: with the look-ahead inverse (woodbury.h: 1.7 ms at 2 137 closures) the exact mode
        // solves city10000's second iterate (2 137 closures, lambda_2 = 0.0012) in 2.5 ms where 3 488 Lanczos steps take 14.3 ms.
        // Cost model in us, from counts only: set-up 120 + 4e-6 n s + 1.1e-5 s^2 (column solves, capacitance), inverse
        // (ld / 32) x (13 + 3.5e-6 ld^2), per iteration 45 + 1.2e-6 n s + 2.5e-6 s^2 (n x s and s x s products); a Lanczos step
        // 4.2 + 2e-6 nnz (tools/city_exact_probe.py, tools/ubench_gj.hip; profiles/r4_exact_big.md).
        bool exact_big = false;
        if (!small && n <= kTriMaxN && !throughput_lane && precision == 0 && support_hint > 0 && support_hint <= wb_soft() && hist_lan_steps > 0 &&
            OPT(woodbury, 1) != 0 && OPT(exact_big, 1) != 0) {
            const double sd = (double)support_hint, nd = (double)n, ldw = (double)((support_hint + kGjT - 1) / kGjT * kGjT);
            const double its = hist_exact_iters > 0 ? (double)hist_exact_iters : 14.0;
            const double est_exact = 120.0 + 4e-6 * nd * sd + 1.1e-5 * sd * sd + (ldw / kGjB) * (13.0 + 3.5e-6 * ldw * ldw) +
                                     its * (45.0 + 1.2e-6 * nd * sd + 2.5e-6 * sd * sd);
            exact_big = est_exact < 0.8 * (double)hist_lan_steps * (4.2 + 2e-6 * (double)nnz);
        }
        const bool want = chain_dominated &&
                          (mode == 2 || (mode == 0 && chain_like && support_hint >= 0 && ((sparse && !slow_lob) || stiff || exact_small || exact_big)));
        last_was_lob = false;
        final_check_seq = -2;
        const bool want_jac = mode == 3 && n > 256;
        // No history (or one that spoke for Lanczos): start with Lanczos, but let it hand over to the exact mode once ITS OWN forecast
        // says the rest of the solve costs more than that (city10000's first iterate: forecast > 1 000 steps from step 128 on -> 0.7 ms
        // of Lanczos + 3.0 ms exact instead of 7.3 ms).  Forecasts come from device results that are bit-reproducible: so is the choice.
        bool after_switch = false;
        long pre_steps = 0;
        if (!((eligible && want) || want_jac) && mode == 0 && eligible && chain_dominated && chain_like && !small && n <= kTriMaxN && !throughput_lane &&
            precision == 0 && forced_variant == 0 && support_hint > 0 && support_hint <= wb_soft() &&      // (sharded / inter-process solves too: every rank
            // holds the same tridiagonal records, takes the same decision at the same step, and runs the exact mode replicated)
            OPT(woodbury, 1) != 0 && OPT(exact_big, 1) != 0 && OPT(exact_switch, 1) != 0) {
            const double sd = (double)support_hint, nd = (double)n, ldw = (double)((support_hint + kGjT - 1) / kGjT * kGjT);
            const double its = hist_exact_iters > 0 ? (double)hist_exact_iters : 14.0;
            switch_est_us = 120.0 + 4e-6 * nd * sd + 1.1e-5 * sd * sd + (ldw / kGjB) * (13.0 + 3.5e-6 * ldw * ldw) +
                            its * (45.0 + 1.2e-6 * nd * sd + 2.5e-6 * sd * sd);
            const int st = solve_lanczos(A, nnz, lnorm, tol, max_steps, start_mode, forced_variant, lambda2, stats);
            switch_est_us = 0.0;
            if (st != kSwitchToExact) {
                if (st == MACHIP_OK) hist_lan_steps = last_steps - (long)(0.35 * (double)last_steps_lowp);
                return st;
            }
            after_switch = true;
            pre_steps = last_steps;
            hist_lan_steps = last_steps + (long)std::min(switch_to_go, 1e6);     // (what the solve would have taken, by its own forecast)
        }
        if ((eligible && want) || want_jac || after_switch) {
            if (!after_switch) HIP_TRY(hipEventRecord(ev0, stream));      // (after a hand-over ev0 still marks the start of the Lanczos part)
            double lam = 0.0, res = 0.0;
            long iters = 0, spmvs = 0, rst = 0;
            // (evaluation lanes run many solves side by side: the exact mode's chip-wide kernels -- s column solves, the s x s inverse,
            // an n x s product per application -- are kept to the cheap cases there, the tridiagonal preconditioner serves the rest and
            // a crawling solve still escalates; city10000, 9 budgets on 4 lanes: 422 -> 664 it/s, profiles/r4_exact_small.md)
            wb_limit_now = throughput_lane ? std::min(wb_soft(), OPT(wb_lane_max, 256)) : wb_soft();
            int st = solve_lob(A, nnz, lnorm, tol, max_steps, start_mode, &lam, &res, &iters, &spmvs, &rst, want_jac);
            if (st == MACHIP_NOT_CONVERGED && lob_escalate) {
                long it1 = iters, sp1 = spmvs;
                wb_limit_now = wb_hard();
                st = solve_lob(A, nnz, lnorm, tol, max_steps, start_mode, &lam, &res, &iters, &spmvs, &rst);
                iters += it1; spmvs += sp1;
            }
            if (st == MACHIP_OK) {
                HIP_TRY(hipEventRecord(ev1, stream));
                HIP_TRY(hipEventSynchronize(ev1));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
                last_mode = wb_active.s > 0 ? 7 : 6; last_wb_s = wb_active.s;
                have_prev = true; last_was_lob = true; last_steps = iters; J_last = 0;
                last_seq_sharded = false;        // (the final solve ran replicated, whatever a Lanczos prefix did: machip_comm_mode reports it)
                hist_lob_iters = iters;
                if (wb_active.s > 0) hist_exact_iters = iters;
                *lambda2 = lam;
                if (stats) {
                    stats->lanczos_steps = iters + pre_steps; stats->spmv_total = spmvs + pre_steps; stats->vec_passes = iters * 16 + pre_steps * 7;
                    stats->restarts = rst; stats->nnz = nnz; stats->residual = res; stats->lnorm = lnorm; stats->gpu_ms = ms;
                    stats->step_ms = ms; stats->steps_timed = iters + pre_steps; stats->steps_lowp = 0;
                }
                if (lam < 1e-12 * (lnorm > 0 ? lnorm : 1.0))
                    return fail(MACHIP_DISCONNECTED, "lambda_2 ~ 0: the graph is not connected");
                return MACHIP_OK;
            }
            if (st != MACHIP_NOT_CONVERGED) return st;
            // stagnated / T not positive definite: the Lanczos path takes over from its own start
        }
#ifdef MACHIP_EXPERIMENTS
        // Block Lanczos (blocklan.h) where the scalar recurrence has been seen to need many steps on a matrix small enough for the
        // one-row-per-lane-group step: counts only (hist_lan_steps = steps of the last scalar solve, hist_blk_steps = block steps of
        // the last block solve; a block step is priced at 1.5 scalar steps), so the choice is reproducible.  option blocklan: 0 never, 1 always.
        {
            const int bo = OPT(blocklan, 0);      // (experiments build: opt-in only)
            const bool pmode_fits = OPT(persist, 1) != 0 && chain_like && persist_fits(n, nnz - n - 2 * chain_edges);
            const bool auto_ok = bo < 0 && mode == 0 && !pmode_fits && forced_variant == 0 && hist_lan_steps > OPT(blocklan_min_steps, 400) &&
                                 (hist_blk_steps < 0 || 3 * hist_blk_steps < 2 * hist_lan_steps);
            if ((bo == 1 || auto_ok) && blk_fits(nnz) && n > 1024) {
                const int sb = solve_block(A, nnz, lnorm, tol, max_steps, start_mode, lambda2, stats);
                if (sb != kBlockFallback) return sb;
            }
        }
#endif
        const int st = solve_lanczos(A, nnz, lnorm, tol, max_steps, start_mode, forced_variant, lambda2, stats);
        // (mixed precision: the fp32 pre-phase inflates the count by roughly a third of its length; the learning rule
        // above is calibrated on fp64 step counts)
        if (st == MACHIP_OK) hist_lan_steps = last_steps - (long)(0.35 * (double)last_steps_lowp);
        return st;
    }

    int solve_lanczos(const CsrView& A, long nnz, double lnorm, double tol, int max_steps, int start_mode,
                      int forced_variant, double* lambda2, machip_solve_stats* stats) {
        // explicit-check kernels: run once or twice per solve, nobody re-reduces their partials per step -- at large n they may
        // use the whole chip (config 4: 84 us per check with 256 workgroups of 8 rows each)
        const SpmvPlan pl = plan_spmv(opt, n, nnz, forced_variant, n > 32768 ? kMaxGrid : 0);
        SpmvPlan pp = plan_pipe(opt, n, nnz, maxlen_hint);            // fused Lanczos-step kernel
        const int g2 = vgrid();
        HIP_TRY(hipEventRecord(ev0, stream));
        ev1_at_check = false;
        if (max_steps <= 0) max_steps = 200000;
        long steps_total = 0, spmv_total = 0, restarts = 0;
        long steps_used = 0;        // steps up to the analysis point each sequence ended at (a function of the records alone; steps_total counts launches)
        double step_ms_acc = 0.0;   // stream time of the Krylov chunks alone (step kernels + one tail kernel per chunk)
        long steps_timed_acc = 0;
        int status = MACHIP_NOT_CONVERGED;
        bool switch_out = false;           // handed over to the exact mode (solve(): switch_est_us)
        double lam = 0.0, res = 0.0;
        const double tiny_l = (lnorm > 0 ? lnorm : 1.0);
        if (n < 2) { *lambda2 = 0.0; return fail(MACHIP_BAD_ARG, "graph with a single node has no Fiedler pair"); }

        // ---- start vector ----
        // (single-workgroup form: k_persist_begin of the first sequence copies it -- one launch less per solve)
        const bool pmode_early = OPT(persist, 1) != 0 && chain_like && persist_fits(n, nnz - n - 2 * chain_edges);
        const double* begin_src = nullptr;
        // The landscape weighting moves the start to where low eigenvectors LOCALISE; it pays where they do (the Erdos-Renyi iterates:
        // participation ratio 1-5 vertices, -15 % steps) and costs its sweeps where the Fiedler vector is global (city10000: thousands of
        // vertices, 297 against 300 it/s in round 5).  Gate, round 6: what the LAST pair this handle checked looked like -- skipped when it lived on
        // more than start_land_pr_permille of the vertices (default 2 %); a handle's first solve has nothing to go by and weights.
        // A handle's FIRST solve is gated on the matrix alone: the weighting is for random-graph-like Laplacians (from ~6 entries per row), not for
        // pose graphs (3-5 entries per row: chain + a few closures), where it never paid (+-1 %) and where a floored, half-localised start flattens
        // the residual estimate's decay enough to fool the step forecast (city10000: a hand-over to the tridiagonally preconditioned mode, 26 ms
        // per iteration instead of 3.3 -- tools/floor_probe.py).
        const bool land_local = last_pr <= 0.0 || last_pr <= 0.001 * (double)std::max(1, OPT(start_land_pr_permille, 20)) * (double)n;
        const bool land_dense = 10.0 * (double)nnz >= (double)std::max(0, OPT(start_land_min_mean10, 60)) * (double)n;
        const bool land_ok = !start_guess && !pmode_early && n > OPT(classic_n, 256) && OPT(start_land, 3) > 0 && land_local && land_dense;
        bool warm = start_mode == 1 && have_prev;
        if (warm && land_ok && warm_skip > 0) { --warm_skip; warm = false; }      // (the last warm start was no better than a random vector)
        const bool warm_probe = warm && land_ok;      // this solve measures what its warm start was worth
        double start_overlap = -1.0;
        if (warm) {
            if (pmode_early) begin_src = yvec;
            else HIP_TRY(hipMemcpyAsync(u, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        } else if (have_start) {
            if (pmode_early) begin_src = start;
            else HIP_TRY(hipMemcpyAsync(u, start, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
        } else {
            k_fill_start<<<g2, kBlock, 0, stream>>>(u, n, 0x1234567ull);
        }

        // ---- landscape weighting of a cold start (kernels.h, k_land_*): the multi-workgroup recurrence only (a single-workgroup solve
        // costs less than the sweeps would), never a caller's own guess or a warm start ----
        pan_live = false;       // (the landscape sweeps run further down, once the panel form -- if this solve takes it -- is built: they can use it)

        const int chunk0 = std::min(kMaxChunk, std::max(2, OPT(chunk, 32) & ~1));   // even: Z parity = jrel & 1
        const int chunk_near = std::min(chunk0, std::max(2, OPT(chunk_near, 8) & ~1));   // once the residual estimate is within 1e3 of the target
        if (max_steps & 1) ++max_steps;
        // An explicit check runs when the recurrence's own estimate of the residual is within this factor of the tolerance.  The estimate
        // predicts the measured residual to +/- 5 % (profiles/r5_c4_checks.txt), so round 1-4's factor of 1.5 bought nothing: every check
        // started between 1.1 and 1.5 x the tolerance failed (0.3 per solve at configs[3], ~100 us each: Ritz vector, product, wait) and
        // the solve went on to the same final step anyway.
        const double trigger_slack = 0.01 * OPT(trigger_pct, 110);
        const double near_factor = 0.1 * OPT(near_x10, 20);   // end game (no speculation, short chunks) from this many chunks of predicted steps to go
        std::deque<Pending> pend;
        bool done = false;

        bool classic = n <= OPT(classic_n, 256);
        // LDS-resident single-workgroup form when the matrix fits (classic recurrence: also fine after restarts)
        const bool pmode = pmode_early;
        // column-panel step (panel.h) where the gather operand is too large for the caches (fp64 sequences only)
        // (shifted recurrence, panel_u.h: only where the host follows the records step by step -- its drift monitor lives there)
        pan = plan_panel(opt, n, nnz, maxlen_hint, pan_allowed && !pmode && !classic && pp.variant == kVec && !shard, (long)csr_cap,
                         false, OPT(stream, 1) != 0);   // (the in-process row-partitioned solve shards the gather step)
        pan_u = false; last_amp = 0.0; pan32 = false; pan32_from = INT_MAX;
        // between processes (round 6): the shifted recurrence partitions by the row kernel's workgroups (launch_chunk_ipc_pan) where one wave of
        // them covers every row; the record form and the multi-cell shapes are not partitioned
        if (ipc && !(pan.on && pan.u && precision == 0 && (long)pan.grid2 * pan.block2 >= (long)n && pan.grid2 >= ipc->nranks && OPT(ipc_panel, 1) != 0)) pan.on = false;
        // mixed mode: only the shifted recurrence in a shape whose fp32-tile kernel exists, launched eagerly (the step index decides which tiles
        // a launch reads); anything else keeps round 2's fp32 gather sequences
        if (precision == 1 && !(pan.on && pan.u && pan.TWT == 3 && OPT(pan32, 1) != 0)) pan.on = false;
        // padded fixed-width form for short rows (pose graphs beyond the single-workgroup kernel): no row-pointer round trip
        if (!pan.on && !pmode && !classic && !shard && !ipc && precision == 0 && pp.variant == kVec && pp.width == 4 && pp.defer < 3 &&
            maxlen_hint >= 1 && maxlen_hint <= 16 && OPT(ell, 1) != 0) {
            const int W = maxlen_hint <= 8 ? 8 : 16;
            if (!ell_col) { ST_TRY(dev_alloc(&ell_col, (size_t)n * 16)); ST_TRY(dev_alloc(&ell_val, (size_t)n * 16)); }
            k_ell_build<<<(int)std::min<long>(kMaxGrid, ((long)n * W + kBlock - 1) / kBlock), kBlock, 0, stream>>>(A, W, ell_col, ell_val);
            HIP_TRY(hipGetLastError());
            ell_view = CsrView{n, nullptr, ell_col, ell_val};
            pp.variant = kEll; pp.unroll = W / 4;
        }
        if (pan.on) {
            const bool rows_ready = pan_rows_ready && !pan.verify && pan_rows_NP == pan.NP && pan_rows_C == pan.C && pan_rows_band == pan.band;
            ST_TRY(ensure_panel(A, nnz, pan, pan.band, rows_ready));
            if (pan.verify) {        // hub rows: is every (row, panel) count within what the tiles describe?  (one sync, hub matrices only)
                int over = 0;
                HIP_TRY(hipMemcpyAsync(&over, panv.ovf, sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (over) pan.on = false;            // a hub row concentrated in one panel: the gather step serves this matrix
            }
        }
        if (pan.on) {
            pp.variant = kPanel; pp.grid = pan.fused ? pan.NB * pan.NP : pan.grid2; pp.block = pan.fused ? kBlock : pan.block2;   // (grid = partial sums per quantity)
            pp.width = pan.NP * 100 + pan.RPT; pp.unroll = pan.TWW; pp.defer = pan.NB + (pan.fused ? 1000 : 0) + 10000 * pan.cells + (pan.u ? 1000000 : 0);
            if (pan.u) {
                if (!pu_U0) {
                    ST_TRY(dev_alloc(&pu_U0, (size_t)n + 2)); ST_TRY(dev_alloc(&pu_U1, (size_t)n + 2)); ST_TRY(dev_alloc(&pu.W, (size_t)n));
                    ST_TRY(dev_alloc(&pu_sig, 2));
                    pu.sig = pu_sig;
                }
                // (inter-process communicator: the operand lives in the record buffers -- 16 n bytes each, idle in this form -- which the peers
                // have mapped already: the row kernel writes its rows of the next operand into every rank's copy)
                pu.U0 = ipc ? reinterpret_cast<double*>(Z0) : pu_U0;
                pu.U1 = ipc ? reinterpret_cast<double*>(Z1) : pu_U1;
                pan_u = true;
                if (precision == 1 && !use_graph(launch_us_hist[kPanel] > 0.0 ? launch_us_hist[kPanel] : 8.0)) {
                    if (pan_bv32_cap < pan_cap) {
                        HIP_TRY(hipStreamSynchronize(stream));
                        if (pan_bv32) (void)hipFree(pan_bv32);
                        pan_bv32 = nullptr; pan_bv32_cap = 0;
                        ST_TRY(dev_alloc(&pan_bv32, pan_cap));
                        pan_bv32_cap = pan_cap;
                    }
                    k_to_f32<<<(int)std::min<long>(kMaxGrid, ((long)pan_cap + kBlock - 1) / kBlock), kBlock, 0, stream>>>(panv.bval, pan_bv32, (long)pan_cap);
                    pan32 = true;
                }
            }
        }
        pan_live = pan.on && pan_u;
        if (!warm && land_ok) ST_TRY(landscape_start(A, landscape_plan(nnz)));
        {   // eager launches or captured chunks for this solve's fused steps (use_graph)
            const int vk = std::max(0, std::min(7, (int)pp.variant));
            // (model before the first measurement: the gather step's; a panel launch never runs under ~8 us -- n >= 65 536.  A first solve must not
            // capture by mistake: on a handle with evaluation lanes, hipGraph executables take hardware queues away from the lanes'
            // streams for good -- c4s with 4 lanes 294 -> 231 it/s after ONE captured solve per lane, tools/r5_eager3.sh)
            cur_launch_us = launch_us_hist[vk] > 0.0 ? launch_us_hist[vk] : (pp.variant == kPanel ? 8.0 : 4.2 + 2e-6 * (double)nnz);
        }
        if (OPT(debug, 0)) fprintf(stderr, "[machip] fused steps: variant %d, %.2f us per launch (%s) -> %s\n", (int)pp.variant, cur_launch_us, launch_us_hist[std::max(0, std::min(7, (int)pp.variant))] > 0.0 ? "measured" : "model", use_graph(cur_launch_us) ? "captured chunks" : "eager launches");
        const PipeView L = pview(pp);
        const int pchunk0 = std::min(kPersistMaxSteps, std::max(2, OPT(pchunk, 64)));
        const bool debug = OPT(debug, 0) != 0;
        // ---- Chebyshev-filtered recurrence for the single-workgroup kernel (persist.h, CHEB): after a short plain
        // sequence has produced a Ritz vector -- its Rayleigh quotient rq is a RIGOROUS upper bound of lambda_2 (unit
        // vector orthogonal to 1) -- the solve restarts from that vector on C = -T_d(M), M mapping [a, b] onto [-1, 1] with
        // a = 1.25 rq > lambda_2 and b = ||L||_inf >= lambda_max.  T_d(M(lambda)) decreases monotonically on [0, a] and
        // stays within [-1, 1] beyond, so C's smallest eigenvalue on 1-perp belongs to lambda_2's eigenvector and to no
        // other; the converged pair is checked on L itself by the same explicit test as always.
        PersistCheb cheb;                        // deg = 0: plain recurrence
        bool cheb_started = false;
        double cheb_a = 0.0, cheb_b = 0.0;
        // Measured on MI355X (tools/cheb_probe.py, bench c3 / c5a, round 3): correct on every pose-graph test, but SLOWER than
        // the plain recurrence -- a filtered step costs 3.2 us + 0.55 us per product against 2.35 us for a plain step, and
        // with a = 1.25 rq ~ 20 lambda_2 after 32 plain steps the filter needs 1.4-4x the products (intel 420-435 against
        // 488 it/s, sphere2500 900-1 013 against 1 049 for degrees 8-24).  Off by default (MACHIP_CHEB_DEG=8 turns it on).
#ifdef MACHIP_EXPERIMENTS
        const int cheb_deg_max = OPT(cheb_deg, 0) & ~1;
#else
        const int cheb_deg_max = 0;
#endif
        const bool cheb_ok = pmode && precision == 0 && cheb_deg_max >= 2 && persist_fits_cheb(n, nnz - n - 2 * chain_edges);
        const int cheb_after = std::max(8, OPT(cheb_after, 32));      // plain steps before the hand-over
        const int cheb_chunk = std::max(4, OPT(cheb_chunk, 16) & ~1);  // filtered steps per launch
        const int cheb_depth = std::max(1, OPT(cheb_depth, 2));
        // T_d(x) and its derivative for x >= 1, and the inverse on that branch
        auto cheb_T = [](int d, double x, double* dT) {
            const double th = std::acosh(std::max(1.0, x));
            const double sh = std::sqrt(std::max(0.0, x * x - 1.0));
            if (dT) *dT = sh > 1e-8 ? d * std::sinh(d * th) / sh : (double)d * d;
            return std::cosh(d * th);
        };
        // ---- mixed precision (machip_set_precision(1)): the FIRST Krylov sequence stores matrix values, records and
        // basis in fp32 (inner products accumulated in fp64).  An fp32 recurrence cannot resolve lambda_2 beyond
        // ~eps_32 ||L||, so it only runs until its residual estimate reaches f32_switch ||L||_inf (or stalls); its Ritz
        // vector is then formed with fp64 accumulation, checked in fp64 (Rayleigh quotient + the reference's residual
        // test on the fp64 matrix) and, when the test does not pass yet, fp64 sequences continue from it. ----
        bool f32_seq = precision == 1 && !classic && pp.variant == kVec;
        // (measured floor of the fp32 recurrence's TRUE residual: 8e-7 at config 2 (||L|| = 50, n = 1e4), 1e-6 on
        // sphere2500, 2e-4 on city10000 (||L|| = 1600, stiff) while its own estimate keeps falling: beyond
        // ~eps_32 sqrt(n) x 10 the fp32 steps buy nothing)
        const double f32_switch = std::max(std::max(tol, 1e-9 * (double)OPT(f32_switch_e9, 2000)),
                                           3e-7 * std::sqrt((double)n));
        long steps_lowp = 0;
        if (f32_seq && !pmode) {
            if (valf_cap < (size_t)nnz) {
                // cached chunk graphs carry the old pointer: drop them together with the buffer
                HIP_TRY(hipStreamSynchronize(stream));
                for (auto& kv : graphs) for (hipGraphExec_t ge : kv.second) if (ge) (void)hipGraphExecDestroy(ge);
                graphs.clear();
                if (valf) (void)hipFree(valf);
                valf = nullptr; valf_cap = 0;
                const size_t want = std::max<size_t>((size_t)nnz + (size_t)nnz / 2 + 1024, csr_cap);   // csr_cap: the most L(x) can ever hold
                ST_TRY(dev_alloc(&valf, want));
                valf_cap = want;
            }
            k_to_f32<<<(int)std::min<long>(kMaxGrid, (nnz + kBlock - 1) / kBlock), kBlock, 0, stream>>>(A.val, valf, nnz);
        }
        if (pmode) ST_TRY(pack_persist(A));
        while (!done && steps_total < max_steps) {
            // ---- (re)start a Krylov sequence from u ----
            seq_sharded = false; seq_ipc = false;
            pan32_from = INT_MAX;
            if (pmode) {
                ++epoch;
                k_persist_begin<<<g2, kBlock, 0, stream>>>(persist_view(), (int)epoch, begin_src);
                begin_src = nullptr;            // (restarts continue from u)
            } else if (classic) {
                k_vec_sums<<<g2, kBlock, 0, stream>>>(u, n, part_u);
                k_set_state<<<1, 64, 0, stream>>>(stc, 0);
            } else if (f32_seq) {
                ++epoch;
                k_pipe_init<<<pp.grid, kBlock, 0, stream>>>(pview<float>(pp), u, (int)epoch);
            } else {
                ++epoch;
                if (pan_u) k_pipe_init_u<<<pp.grid, kBlock, 0, stream>>>(L, pu, u, (int)epoch);
                else k_pipe_init<<<pp.grid, kBlock, 0, stream>>>(L, u, (int)epoch);
                if (pan.on && pan.fused) {      // (experiments build) arrival tickets / slice claims of k_pan_step count from the sequence's step 0
                    HIP_TRY(hipMemsetAsync(panv.tick, 0, sizeof(unsigned int) * 256, stream));
                    HIP_TRY(hipMemsetAsync(panv.claim, 0, sizeof(unsigned int) * 4096, stream));
                }
                if (shard && pp.variant == kVec) {       // row-partitioned sequence: every rank starts from the same records
                    seq_sharded = true; seq_plan = pp;
                    ST_TRY(shard_broadcast_init());
                } else if (ipc && precision == 0 && ((pp.variant == kVec && pp.grid >= ipc->nranks) || (pp.variant == kPanel && pan_u))) {
                    // (precision 1: the fp32 sequences run replicated on the very record / partial-sum buffers the peers of a
                    // partitioned step write into, and nothing orders a rank that is still in its fp32 phase against a peer that has
                    // entered the fp64 one -- the mixed mode therefore never partitions; round-4 advisor finding)
                    // between processes every rank has just run k_pipe_init itself on its own copy of u: identical records
                    seq_sharded = true; seq_ipc = true; seq_plan = pp;
                }
            }
            const double seq_tol = f32_seq ? f32_switch : tol;     // what this sequence's residual estimate aims for
            int J_enq = 0;        // steps enqueued in this sequence
            int prev_tailless = 0;   // step count of the last enqueued chunk if its records still wait for a successor's first step
            const bool tail_ok = !pmode && !classic && !seq_sharded && cheb.deg == 0 && !(pan.on && pan.fused) && OPT(tailless, 1) != 0;
            int J_timed = 0;      // ... of which already accounted in step_ms
            HIP_TRY(hipEventRecord(evs0, stream));
            ha.clear(); hb.assign(1, 0.0); hl1.assign(1, 0.0);
            guess.clear();
            double theta_prev = 0.0, last_check_est = 1e300, est_latest = 1e300;
            bool converged = false, need_restart = false;
            // convergence-rate tracking: (J, ln est) over a >= 64-step window predicts the steps still
            // needed, which sizes the chunks and the speculation depth (stiff chain-like graphs spend
            // thousands of steps within 1e3 of the target; random graphs a few dozen)
            std::deque<std::pair<int, double>> hist;
            double to_go = 1e18;
            int switch_votes = 0;
            const bool sched = OPT(sched, 1) != 0;
            const double ltarget = std::log(std::max(seq_tol * tiny_l, 1e-300));
            const int jcap = (int)std::min<size_t>(vcap - 2, (size_t)std::max(2, n - 1) + 8) & ~1;
            const size_t cs = vcap + 2;   // stride of the classic alpha / beta / l1 arrays

            // ---- streamed records (round 5): where the solve ends is decided step by step, not chunk by chunk ----
            // Every step hands its (alpha_{j-1}, l1_{j-1}, beta_j) to the host as it derives them (PipeView::pubstep), so the host
            // (a) analyses T_a at a DETERMINISTIC sequence of points a_0 < a_1 < ... -- each next point a function of the analyses
            //     before it only (a third of the forecast distance to the target, at most 32 steps) -- and ends the sequence at the FIRST
            //     point whose residual estimate is below the trigger: the Ritz pair, and with it every later result, is a function
            //     of the records alone, never of timing;
            // (b) feeds the queue from the forecast: far from the end 32-step chunks one chunk ahead, then chunks of about half
            //     the predicted remainder, each enqueued only when the GPU is within `look` steps of running dry, none beyond the
            //     predicted crossing (+ margin), a one-wave tail kernel behind the last.  How many steps run BEYOND the final
            //     analysis point depends on timing; nothing reads them.
            // The chunk-granular loop below overshoots the crossing by 8.7 steps per solve at configs[3] (3.6 % of 242:
            // tools/sched_probe.sh) and idles the GPU for a host round trip at each of its 2-3 end-game hops.
            // Row-partitioned sequences (every rank has to launch the same steps) and option stream = 2 feed the queue from the analyses
            // alone instead: one chunk with a tail kernel at a time, sized by the forecast once every record of its predecessor has been
            // analysed.  Same points, same rule, same final point: a partitioned solve still reproduces the single-rank one bit for bit.
            const bool stream_mode = !pmode && !classic && cheb.deg == 0 && !(pan.on && pan.fused) && !f32_seq && OPT(stream, 1) != 0;
            const bool feed_chunks = seq_sharded || OPT(stream, 1) == 2 || OPT(tailless, 1) == 0;
            if (stream_mode) {
                const double trig_s = 0.01 * OPT(stream_trigger_pct, 95);      // (the estimate predicts the measured residual to +/- 5 %: profiles/r5_c4_checks.txt)
                const double retry_s = 0.01 * OPT(stream_retry_pct, 80);       // after a failed check: the next one when the estimate has fallen to this fraction
                const int margin = std::max(0, OPT(stream_margin, 0));
                // time of one step: measured in-solve on this handle's last solve in the same step form, the hand-over forecast's model before
                const bool eager = !use_graph(cur_launch_us);
                const double step_us = std::max(1.0, cur_launch_us * (pp.variant == kPanel ? 2.0 : 1.0));
                const int look_opt = OPT(stream_look, 0);                      // steps of queued work below which the queue is fed (0: from the step-time model)
                const int far_rem = std::max(34, OPT(stream_far, 80));         // whole chunks one ahead while at least this many steps are predicted to remain
                const int win_max = std::max(8, OPT(stream_window, 64));
                FollowCfg fc;
                fc.n = n; fc.jcap = jcap; fc.chunk0 = chunk0; fc.win_max = win_max; fc.margin = margin; fc.tiny_l = tiny_l;
                Follower F(fc, trig_s * seq_tol * tiny_l, ha, hb, hl1, guess, sm, wk);      // (follow.h: the analysis points, the estimate, the forecast)
                int& next_a = F.next_a;      // the next analysis point
                int& T = F.T;                // forecast: no step >= T is wanted (INT_MAX: no forecast yet)
                int prog = -1;               // records up to beta_prog (alpha, l1 up to prog - 1) have landed
                int amp_at = 0;              // shifted recurrence: drift factor accumulated over the steps < amp_at (a function of the records)
                double amp = 1.0;
                const double amp_max = (double)std::max(2, OPT(panel_u_amp, 100000));
                int tail_at = -1;            // a tail kernel has been enqueued behind step tail_at - 1 (delivers beta_tail_at)
                int last_chunk = 0;          // steps of the last enqueued chunk (what its tail kernel advances by)
                unsigned long spins = 0;
                bool stalled = false;
                auto landed = [&](int j) {
                    const volatile double* t3 = h_tri;
                    if (t3[3 * (size_t)j + 1] != t3[3 * (size_t)j + 1]) return false;
                    if (j > 0 && (t3[3 * (size_t)(j - 1)] != t3[3 * (size_t)(j - 1)] || t3[3 * (size_t)(j - 1) + 2] != t3[3 * (size_t)(j - 1) + 2])) return false;
                    return true;
                };
                for (;;) {
                    // (1) how far has the GPU got?
                    const int limit = std::min(tail_at == J_enq ? J_enq : J_enq - 1, jcap);
                    while (prog < limit && (stalled || landed(prog + 1))) ++prog;
                    // (2) analysis, as soon as the records of the next point are there -- BEFORE the queue is fed: a host that cannot launch as
                    // fast as the GPU runs (eager launches under a profiler) would otherwise feed for ever and never look
                    const int a = std::min(next_a, std::min(J_enq, jcap));       // (J_enq < next_a only at the caps)
                    // An analysis at a point SHORT of next_a is for the caps only (nothing more will be enqueued).  While the feeder still has
                    // steps to enqueue it must come first: whether the records up to J_enq have landed before or after the loop passes here is
                    // timing, and an analysis at J_enq would change every later point -- two ranks of a row-partitioned solve then disagreed
                    // about where the sequence ends (2 of 30 two-rank configs[3] runs of round 5's library, 3 of 14 with the panel step:
                    // tools/archive/ipc_pan_debug.py prints the first diverging trace line).
                    const bool more_coming = J_enq < jcap && steps_total < max_steps && J_enq < std::max(T, next_a);
                    bool analysed = false;
                    if (prog >= a && a > 0 && (a == next_a || !more_coming)) {
                        analysed = true;
                        ST_TRY(ipc_check_err("Lanczos steps"));
                        const int J = a;
                        if (!F.analyse(h_tri, a)) return fail(MACHIP_BAD_ARG, "start vector is constant, zero or not finite");
                        const int Jeff = F.Jeff;
                        const bool broke = F.broke;
                        const double est = F.est;
                        est_latest = est; to_go = F.to_go;
                        // Mixed mode of the panel step: products may be INEXACT late in a Krylov sequence -- an error E_j in step j's product enters the
                        // final residual weighted by the Ritz vector's coefficient of v_j, which is ~ the residual at that step (Simoncini 2005 /
                        // Bouras & Fraysse: relaxed accuracy of matrix-vector products in projection methods) -- not early.  So the sequence starts on
                        // the fp64 tiles and switches to the fp32 copy (6 bytes per entry instead of 10, relative error 6e-8) once the estimate is below
                        // pan32_switch_e9 1e-9 ||L||.  Measured on the 20 configs[3] iterates (tools/pan32_probe.py, profiles/r6_pan32.md): 1e-3 passes the
                        // explicit check on 19 -- residuals unchanged in the second digit -- and FAILS on the near-degenerate iterate 6 (353 steps:
                        // the gap grows with 1 / (1 - convergence rate)), whose restart in the two-kernel form then costs 5 000 steps; 3e-4 (default)
                        // passes on all twenty.  The switch step is a function of the records alone: 32 steps behind the analysis point that saw
                        // the threshold -- as far as the feeder lets the queue run while undecided.
                        if (pan32 && pan32_from == INT_MAX && est < 1e-9 * (double)std::max(1, OPT(pan32_switch_e9, 300000)) * tiny_l) {
                            pan32_from = (J + 32 + 1) & ~1;       // (the feeder has held the queue at next_a + 32 = J + 32 until now)
                            if (debug) fprintf(stderr, "[machip]    J=%d: estimate %.2e -- fp32 tile values from step %d on\n", J, est / tiny_l, pan32_from);
                        }
                        bool amp_trip = false;
                        if (pan_u && pp.variant == kPanel) {
                            // w_{k+1} = (L u_k - (alpha_k - alpha_{k-1}) w_k) / beta_{k+1}: what step k multiplies the difference w_k - L v_k by
                            for (; amp_at < J; ++amp_at) {
                                const double ak = h_tri[3 * (size_t)amp_at], akm = amp_at ? h_tri[3 * (size_t)(amp_at - 1)] : 0.0, bk1 = h_tri[3 * (size_t)(amp_at + 1) + 1];
                                amp = (bk1 > 0.0 ? std::fabs(ak - akm) / bk1 : 0.0) * amp + 1.0;
                                last_amp = std::max(last_amp, amp);
                            }
                            amp_trip = !(last_amp <= amp_max);
                            if (amp_trip && debug) fprintf(stderr, "[machip]    J=%d: drift factor of the shifted recurrence %.3g > %.3g -- this sequence ends here, the solve continues in the accurate form\n", J, last_amp, amp_max);
                        }
                        const bool at_cap = (J >= jcap) || (steps_total >= max_steps && J >= J_enq) || amp_trip;
                        const bool trig = F.triggered();
                        if (switch_est_us > 0.0 && restarts == 0 && !broke && !trig && !at_cap && J >= 128 && to_go < 1e17 &&
                            to_go * (4.2 + 2e-6 * (double)nnz) > 1.3 * switch_est_us) {
                            if (++switch_votes >= 2) {
                                if (debug) fprintf(stderr, "[machip]    J=%d: forecast %.0f steps to go -- handing over to the exact chain + closures mode (estimate %.0f us)\n", J, to_go, switch_est_us);
                                switch_to_go = to_go; switch_out = true; steps_used += J; break;
                            }
                        } else switch_votes = 0;
                        if (debug) fprintf(stderr, "[machip] stream J=%d Jeff=%d theta=%.15g est=%.3e to_go=%.1f T=%d next_a=%d enq=%d prog=%d broke=%d passes=%d\n", J, Jeff, sm.theta, lnorm > 0 ? est / lnorm : est, std::min(to_go, 1e9), T == INT_MAX ? -1 : T, next_a, J_enq, prog, (int)broke, sm.passes);
                        if (trig || at_cap) {
                            double rq = 0.0, r1 = 0.0;
                            if (tail_at != J_enq && J_enq > 0 && last_chunk > 0) { flush_tail(pp, last_chunk, false); tail_at = J_enq; }    // (a tail-less chunk never ends a sequence: the counters move with its successor)
                            HIP_TRY(hipEventRecord(evs1, stream));
                            spec_likely = est < 0.01 * OPT(spec_slack_pct, 105) * seq_tol * lnorm;
                            ST_TRY(explicit_check(A, pl, Jeff, sm.s.data(), &rq, &r1, false));   // syncs the stream: every enqueued step has run
                            spec_likely = true;
                            {
                                float sms = 0.f;
                                HIP_TRY(hipEventElapsedTime(&sms, evs0, evs1));
                                step_ms_acc += sms; steps_timed_acc += J_enq - J_timed;
                                J_timed = J_enq;
                                HIP_TRY(hipEventRecord(evs0, stream));
                            }
                            spmv_total += 1;
                            J_last = Jeff;
                            last_check_est = std::max(est, 1e-300);
                            lam = rq;
                            res = lnorm > 0 ? r1 / lnorm : r1;
                            if (debug) fprintf(stderr, "[machip]    check J=%d rq=%.15g res=%.3e (tol %.1e) ran=%d\n", Jeff, rq, res, tol, J_enq);
                            if (res < tol) {
                                converged = true; status = MACHIP_OK; final_check_seq = check_seq; steps_used += Jeff;
                                if (pan32 && pan32_from < Jeff) steps_lowp += Jeff - pan32_from;      // (steps that read fp32 tile values)
                                if (restarts == 0 && !sm.s.empty()) start_overlap = std::fabs(sm.s[0]);     // <v_0, y>: what the start vector was worth
                                break;
                            }
                            if (broke || at_cap) { need_restart = true; steps_used += Jeff; if (pan32 && pan32_from < Jeff) steps_lowp += Jeff - pan32_from; break; }
                            F.lower_target(retry_s * last_check_est);      // the estimate flattered the residual: further down before the next check
                            T = std::max(T, J_enq + 2);
                        }
                    }
                    // (3) feed the queue
                    const bool room = J_enq < jcap && steps_total < max_steps;
                    const int want = std::max(T, next_a);      // beta_{next_a} comes from step next_a -- or from a tail kernel when the queue ends exactly there
                    if (feed_chunks) {
                        // (decided in a state that is a function of the records: all of the queue delivered, every point up to there analysed)
                        if (room && J_enq < want && prog >= J_enq - (J_enq == 0) && next_a > prog) {
                            const long remaining = (long)want - J_enq;
                            // (far: long chunks, a host round trip each; near: a little short of the predicted crossing -- the forecast errs on the long side)
                            int chunk = remaining >= 4 * chunk0 ? std::min(kMaxChunk, 2 * chunk0) : remaining >= 2 * chunk0 ? chunk0
                                        : remaining >= 12 ? (int)std::min<long>(chunk0, ((long)(0.7 * (double)remaining) + 1) & ~1L) : (int)((remaining + 1) & ~1L);
                            chunk = std::min(std::max(chunk, 2), jcap - J_enq);
                            chunk = (int)std::min<long>(chunk, max_steps - steps_total) & ~1;
                            if (pan32 && pan32_from == INT_MAX) chunk = std::min(chunk, std::max(0, next_a + 32 - J_enq)) & ~1;
                            if (chunk >= 2) {
                                const int hi = J_enq + chunk;
                                const double qnan = std::numeric_limits<double>::quiet_NaN();
                                for (int j = J_enq ? J_enq + 1 : 0; j <= hi; ++j) h_tri[3 * (size_t)j + 1] = qnan;
                                for (int j = J_enq; j < hi; ++j) { h_tri[3 * (size_t)j] = qnan; h_tri[3 * (size_t)j + 2] = qnan; }
                                ST_TRY(enqueue_chunk(A, pp, chunk, false, false, 0, J_enq));
                                if (debug) fprintf(stderr, "[machip] chunk enqueue J=%d chunk=%d T=%d next_a=%d\n", J_enq, chunk, T == INT_MAX ? -1 : T, next_a);
                                J_enq = hi; last_chunk = chunk; tail_at = hi;
                                steps_total += chunk; spmv_total += chunk;
                                continue;
                            }
                        }
                    } else if (room && J_enq < want) {
                        const long remaining = (long)want - J_enq;
                        const int ahead = J_enq - std::max(prog, 0);
                        // (the host must be back before the queue runs dry: a graph launch + one O(J) analysis of the tridiagonal)
                        const int look = look_opt > 0 ? look_opt : std::max(2, std::min(24, (int)(((eager ? 28.0 : 36.0) + 0.07 * (double)J_enq) / step_us) + 1));
                        int chunk = 0;
                        if (remaining >= far_rem) { if (ahead <= 16 + look) chunk = J_enq >= 4096 ? std::min(kMaxChunk, 2 * chunk0) : chunk0; }
                        else if (ahead <= look) {
                            chunk = 2;
                            // captured chunks: about 0.6 of the remainder, a power of two (few shapes to capture); eager launches have no
                            // shapes -- two steps at a time, every decision on the latest forecast
                            if (!eager) while (2 * chunk <= chunk0 && 5 * (long)(2 * chunk) <= 3 * remaining + 4) chunk *= 2;
                        }
                        // (mixed panel mode, switch step not decided yet: nothing is enqueued beyond 32 steps behind the next analysis point, so that
                        // the point that sees the threshold can still name a switch step nobody has launched -- a function of the records alone)
                        if (pan32 && pan32_from == INT_MAX) chunk = std::min(chunk, std::max(0, next_a + 32 - J_enq));
                        if (chunk > 0) {
                            chunk = std::min(chunk, jcap - J_enq);
                            chunk = (int)std::min<long>(chunk, max_steps - steps_total) & ~1;
                            if (chunk >= 2) {
                                const int hi = J_enq + chunk;
                                const double qnan = std::numeric_limits<double>::quiet_NaN();
                                for (int j = J_enq ? J_enq + 1 : 0; j <= hi; ++j) h_tri[3 * (size_t)j + 1] = qnan;
                                for (int j = J_enq; j < hi; ++j) { h_tri[3 * (size_t)j] = qnan; h_tri[3 * (size_t)j + 2] = qnan; }
                                ST_TRY(enqueue_chunk(A, pp, chunk, false, true, -1, J_enq));
                                if (debug) fprintf(stderr, "[machip] stream enqueue J=%d chunk=%d prog=%d T=%d next_a=%d\n", J_enq, chunk, prog, T == INT_MAX ? -1 : T, next_a);
                                J_enq = hi; last_chunk = chunk;
                                steps_total += chunk; spmv_total += chunk;
                                continue;
                            }
                        }
                    }
                    // nothing more is wanted (or allowed) and the last record needs a tail kernel: beta_{J_enq} comes from the step that follows, or from a tail
                    if (!feed_chunks && tail_at != J_enq && J_enq > 0 && (J_enq >= want || !room) && next_a >= J_enq) {
                        flush_tail(pp, last_chunk, false);
                        tail_at = J_enq;
                        if (debug) fprintf(stderr, "[machip] stream tail at J=%d (prog=%d T=%d next_a=%d)\n", J_enq, prog, T == INT_MAX ? -1 : T, next_a);
                        continue;
                    }
                    // (4) nothing to do: wait for records
                    if (!analysed) {
                        if ((++spins & 0xfffff) == 0) {     // a device fault or a genuine NaN (non-finite input) must not hang the host
                            const hipError_t q = hipStreamQuery(stream);
                            if (q != hipSuccess && q != hipErrorNotReady)
                                return fail(MACHIP_HIP_ERROR, std::string("stream error while following the Lanczos steps: ") + hipGetErrorString(q));
                            if (q == hipSuccess && prog < limit && !landed(prog + 1)) stalled = true;        // everything enqueued has run: the slot holds a NaN of the recurrence's own
                            ST_TRY(ipc_check_err("Lanczos steps"));
                        }
                        __builtin_ia32_pause();
                    }
                }
            } else
            while (!converged && !need_restart) {
                // keep one chunk in flight beyond the one being analysed
                // one chunk runs ahead of the host -- except in the end game (same threshold as the
                // short chunks), where the chunk in flight is likely the last one and a speculative
                // successor would only delay the explicit residual check queued behind it
                const bool near = sched ? (to_go < near_factor * chunk0 || (to_go >= 1e17 && est_latest < 1e3 * seq_tol * lnorm))
                                        : est_latest < 1e3 * seq_tol * lnorm;
                const bool use_classic = classic && !pmode;
                int depth = (use_classic || near) ? 1 : ((sched && to_go > 8.0 * chunk0) ? 3 : 2);
                if (cheb.deg) depth = (to_go < 3.0 * cheb_chunk * cheb.deg) ? 1 : std::min(2, cheb_depth);     // (chunks are ~80 us: little to hide, much to overshoot)
                while ((int)pend.size() < depth && J_enq < jcap && steps_total < max_steps) {
                    // (the O(J) host analysis must keep up with the GPU: longer chunks once J is large -- a function of
                    // J only, so the step count at which convergence is noticed stays reproducible)
                    int chunk = near ? chunk_near : (J_enq >= 4096 ? std::min(kMaxChunk, 2 * chunk0) : chunk0);
                    if (near && sched && to_go < 1e17)   // aim a little short of the predicted crossing
                        chunk = std::min(chunk0, std::max(chunk_near, ((int)(0.75 * to_go) + 1) & ~1));
                    if (pmode) {   // steps cost ~0.5 us here: long chunks, so the O(J) host analysis keeps up
                        chunk = pchunk0;
                        if (to_go < 1e17) chunk = std::min(pchunk0, std::max(16, ((int)(0.9 * to_go) + 1) & ~1));
                        if (cheb_ok && cheb.deg == 0 && restarts == 0 && J_enq < cheb_after) chunk = std::min(chunk, cheb_after - J_enq);   // decide after a short plain sequence
                        if (cheb.deg) {      // a filtered step is cheb.deg products (~5 us): short chunks, to_go counts products there
                            chunk = std::min(chunk, cheb_chunk);
                            if (to_go < 1e17) chunk = std::max(4, std::min(chunk, (int)(0.9 * to_go / cheb.deg) + 2) & ~1);
                        }
                    }
                    if (use_classic) chunk = std::min(chunk, 16);
                    chunk = std::min(chunk, jcap - J_enq);
                    chunk = (int)std::min<long>(chunk, max_steps - steps_total);
                    if (chunk <= 0) break;
                    const int lo = std::max(0, J_enq - 1);
                    const int hi = J_enq + chunk;           // records lo..hi inclusive
                    if (pmode || !classic) {   // zero-copy records: poison every slot this chunk delivers for the first time
                        const double qnan = std::numeric_limits<double>::quiet_NaN();
                        for (int j = J_enq ? J_enq + 1 : 0; j <= hi; ++j) h_tri[3 * (size_t)j + 1] = qnan;        // beta_j
                        for (int j = J_enq; j < hi; ++j) { h_tri[3 * (size_t)j] = qnan; h_tri[3 * (size_t)j + 2] = qnan; }   // alpha_j, l1_j
                    }
                    if (pmode) {
                        launch_persist(A, chunk, f32_seq, cheb);
                        HIP_TRY(hipGetLastError());   // (157 KB of static LDS: a refused launch must surface, not time out)
                    } else if (classic) {
                        enqueue_classic(A, pl, chunk);
                        double* hp = h_pin + (vcap + 2) + 2 * kMaxGrid + 32;   // staging: 3 x (chunk+1)
                        for (int q = 0; q < 3; ++q)
                            HIP_TRY(hipMemcpyAsync(hp + q * (size_t)(kMaxChunk + 2), ctri + q * cs + (size_t)J_enq,
                                                   sizeof(double) * (size_t)(chunk + 1), hipMemcpyDeviceToHost, stream));
                    } else {
                        // far from the end (a successor will follow): no tail kernel, the successor's first step publishes
                        const bool tailless = tail_ok && depth >= 2 && chunk >= 2;
                        ST_TRY(enqueue_chunk(A, pp, chunk, f32_seq, tailless, prev_tailless, J_enq));
                        if (debug) fprintf(stderr, "[machip] enqueue J=%d chunk=%d tailless=%d pub=%d depth=%d near=%d\n", J_enq, chunk, (int)tailless, prev_tailless, depth, (int)near);
                        prev_tailless = tailless ? chunk : 0;
                    }
                    (void)lo;
                    Pending p;
                    p.tailless = (pmode || classic) ? 0 : prev_tailless;
                    p.jstart = J_enq;
                    p.classic = use_classic;
                    p.jend = hi;
                    p.ev = nullptr;
                    if (use_classic) {
                        ST_TRY(get_event(&p.ev));
                        HIP_TRY(hipEventRecord(p.ev, stream));
                    }
                    pend.push_back(p);
                    J_enq = hi;
                    steps_total += chunk; spmv_total += cheb.deg ? (long)chunk * cheb.deg : chunk;
                    steps_used += chunk;
                    if (f32_seq) steps_lowp += chunk;
                }
                if (pend.empty()) { need_restart = true; break; }
                // a tail-less chunk that gets no successor now (end game, caps): its tail kernel after all
                if (prev_tailless && pend.back().tailless && (depth == 1 || J_enq >= jcap || steps_total >= max_steps)) {
                    flush_tail(pp, prev_tailless, f32_seq);
                    pend.back().tailless = 0; prev_tailless = 0;
                }
                Pending p = pend.front();
                pend.pop_front();
                if (p.tailless && pend.empty()) {     // (cannot happen after the test above; a lost record would stall the host)
                    flush_tail(pp, p.tailless, f32_seq);
                    prev_tailless = 0;
                }
                if (p.classic) {
                    HIP_TRY(hipEventSynchronize(p.ev));
                    ev_pool.push_back(p.ev);
                } else {
                    ST_TRY(wait_flag(((unsigned long long)epoch << 32) | (unsigned long long)(unsigned int)p.jend));
                    ST_TRY(ipc_check_err("Lanczos chunk"));
                    // the flag can overtake the records on their way to host memory: the beta slots of this chunk
                    // were poisoned before it was enqueued, wait until every one has landed
                    unsigned long budget = 5000000ul;   // ~20 ms in total: a genuine NaN (non-finite input) must not stall the solve
                    auto wait_slot = [&](size_t idx) {
                        volatile double* slot = h_tri + idx;
                        while (*slot != *slot && budget) { --budget; __builtin_ia32_pause(); }
                    };
                    for (int j = p.jstart ? p.jstart + 1 : 0; j <= p.jend && budget; ++j) wait_slot(3 * (size_t)j + 1);
                    for (int j = p.jstart; j < p.jend && budget; ++j) { wait_slot(3 * (size_t)j); wait_slot(3 * (size_t)j + 2); }
                }
                if (p.classic) {   // scatter the staged (alpha, beta, l1) into the interleaved mirror
                    const double* hp = h_pin + (vcap + 2) + 2 * kMaxGrid + 32;
                    for (int i = 0; i <= p.jend - p.jstart; ++i) {
                        const size_t j = (size_t)(p.jstart + i);
                        h_tri[3 * j] = hp[i];
                        h_tri[3 * j + 1] = hp[(kMaxChunk + 2) + i];
                        h_tri[3 * j + 2] = hp[2 * (kMaxChunk + 2) + i];
                    }
                }
                const int Jold = (int)ha.size();
                const int J = p.jend;
                ha.resize((size_t)J); hb.resize((size_t)J + 1); hl1.resize((size_t)J + 1);
                for (int j = std::max(0, Jold - 1); j < J; ++j) {
                    ha[(size_t)j] = h_tri[3 * (size_t)j];
                    hl1[(size_t)j] = h_tri[3 * (size_t)j + 2];
                }
                for (int j = Jold; j <= J; ++j) hb[(size_t)j] = h_tri[3 * (size_t)j + 1];
                if (Jold == 0 && (hb[0] <= 0.0 || !(hb[0] == hb[0])))
                    return fail(MACHIP_BAD_ARG, "start vector is constant, zero or not finite");
                // ---- breakdown: beta_j ~ 0 means span(v_0..v_{j-1}) is invariant ----
                int Jeff = J;
                bool broke = false;
                const double bscale = cheb.deg ? 1.0 : tiny_l;         // (the filtered operator's spectrum is O(1))
                for (int j = std::max(1, Jold); j <= J; ++j) {
                    if (!(hb[(size_t)j] > 1e-13 * bscale)) { Jeff = j; broke = true; break; }
                }
                // ---- host: smallest Ritz pair of T_Jeff ----
                tri::smallest_eigpair(ha.data(), hb.data(), Jeff, guess.data(), (int)guess.size(), theta_prev, sm, wk);
                guess = sm.s;
                theta_prev = sm.theta;
                const double rho = broke ? 0.0 : std::fabs(hb[(size_t)Jeff] * sm.s[(size_t)Jeff - 1]);
                const double l1v = hl1[(size_t)Jeff - 1] > 0 ? hl1[(size_t)Jeff - 1] : std::sqrt((double)n);
                double est = rho * l1v;   // predicted ||r||_1 (r = rho v_J; ||v_J||_1 ~ ||v_{J-1}||_1)
                if (cheb.deg) {
                    // residual of C = -p(L) -> residual of L: (C + p(lambda)) y ~ -p'(lambda) (L - lambda) y near an eigenpair,
                    // lambda from the Ritz value through the inverse of p on [0, a]
                    const double pv = -sm.theta;                                   // p(lambda-hat)
                    double x = 1.0, dT = (double)cheb.deg * cheb.deg;
                    if (pv > 1.0) { x = std::cosh(std::acosh(pv) / cheb.deg); (void)cheb_T(cheb.deg, x, &dT); }
                    const double dp = dT * 2.0 / (cheb_b - cheb_a);                 // |p'(lambda-hat)| (>= its value at a)
                    est /= dp;
                }
                est_latest = est;
                if (!broke && est > 0.0) {
                    hist.emplace_back(J, std::log(est));
                    while (hist.size() > 2 && hist[1].first <= J - 64) hist.pop_front();
                    to_go = 1e18;
                    if (hist.front().first < J) {
                        const double slope = (hist.front().second - hist.back().second) / (double)(J - hist.front().first);
                        if (slope > 1e-7) to_go = std::max(0.0, (hist.back().second - ltarget) / slope) * (cheb.deg ? cheb.deg : 1);
                    }
                }
                const bool at_cap = (J >= jcap) || (steps_total >= max_steps && pend.empty());
                const bool trig = broke || est < trigger_slack * seq_tol * lnorm;
                if (switch_est_us > 0.0 && restarts == 0 && !f32_seq && cheb.deg == 0 && !broke && !trig && !at_cap && J >= 128 && to_go < 1e17 &&
                    to_go * (4.2 + 2e-6 * (double)nnz) > 1.3 * switch_est_us) {
                    if (++switch_votes >= 2) {
                        if (debug) fprintf(stderr, "[machip]    J=%d: forecast %.0f steps to go -- handing over to the exact chain + closures mode (estimate %.0f us)\n", J, to_go, switch_est_us);
                        switch_to_go = to_go; switch_out = true; break;
                    }
                } else switch_votes = 0;

                if (debug) fprintf(stderr, "[machip] %s J=%d Jeff=%d theta=%.15g est=%.3e to_go=%.0f broke=%d pend=%zu passes=%d\n", pmode ? "persist" : (classic ? "classic" : "pipe"), J, Jeff, sm.theta, lnorm > 0 ? est / lnorm : est, std::min(to_go, 1e9), (int)broke, pend.size(), sm.passes);
                bool handover = false;
                if (cheb_ok && cheb.deg == 0 && restarts == 0 && !trig && !broke && !at_cap && J >= cheb_after &&
                    (to_go > 4.0 * cheb_after || to_go >= 1e17)) handover = true;     // far from done: the filtered recurrence pays
                if ((trig && est < 0.5 * last_check_est) || broke || at_cap || handover) {
                    double rq = 0.0, r1 = 0.0;
                    HIP_TRY(hipEventRecord(evs1, stream));   // everything enqueued so far = steps [J_timed, J_enq)
                    spec_likely = !f32_seq && !handover && est < 0.01 * OPT(spec_slack_pct, 105) * seq_tol * lnorm;
                    ST_TRY(explicit_check(A, pl, Jeff, sm.s.data(), &rq, &r1, f32_seq));   // syncs the stream
                    spec_likely = true;
                    {
                        float sms = 0.f;
                        HIP_TRY(hipEventElapsedTime(&sms, evs0, evs1));
                        step_ms_acc += sms; steps_timed_acc += J_enq - J_timed;
                        J_timed = J_enq;
                        HIP_TRY(hipEventRecord(evs0, stream));
                    }
                    spmv_total += 1;
                    J_last = Jeff;
                    last_check_est = std::max(est, 1e-300);
                    lam = rq;
                    res = lnorm > 0 ? r1 / lnorm : r1;
                    if (debug) fprintf(stderr, "[machip]    check J=%d rq=%.15g res=%.3e (tol %.1e)\n", Jeff, rq, res, tol);
                    if (res < tol) { converged = true; status = MACHIP_OK; final_check_seq = check_seq; break; }
                    if (handover) {
                        // rq = Rayleigh quotient of the unit Ritz vector (orthogonal to 1) >= lambda_2
                        cheb_a = 1.25 * rq; cheb_b = 1.0001 * tiny_l;
                        int d = cheb_deg_max;
                        if (cheb_a > 0.0 && cheb_b > 4.0 * cheb_a) d = std::min(d, 2 * (int)(0.25 * std::sqrt(cheb_b / cheb_a)));
                        else d = 0;
                        if (d >= 2) {
                            cheb.deg = d; cheb.c1 = 2.0 / (cheb_b - cheb_a); cheb.c0 = (cheb_b + cheb_a) / (cheb_b - cheb_a);
                            if (debug) fprintf(stderr, "[machip]    hand-over to the filtered recurrence: a=%.6g b=%.6g deg=%d\n", cheb_a, cheb_b, d);
                            need_restart = true; break;
                        }
                    }
                    if (broke || at_cap || f32_seq) { need_restart = true; break; }   // fp32 sequence: one check, then fp64
                }
            }
            // streamed records: steps still in flight keep writing record slots the next sequence / mode would poison and await
            if (stream_mode && switch_out) HIP_TRY(hipStreamSynchronize(stream));
            // drain what is still in flight (its results are not needed)
            for (const Pending& p : pend) {
                if (p.ev) { HIP_TRY(hipEventSynchronize(p.ev)); ev_pool.push_back(p.ev); }
            }
            if (!pend.empty()) HIP_TRY(hipStreamSynchronize(stream));
            pend.clear();
            last_seq_f32 = f32_seq;
            last_seq_sharded = seq_sharded;
            if (switch_out) break;
            if (converged) { done = true; break; }
            if (steps_total >= max_steps) break;
            // restart from the best Ritz vector found so far (it sits normalised in yvec)
            HIP_TRY(hipMemcpyAsync(u, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
            if (cheb.deg && !cheb_started) { cheb_started = true; continue; }   // the planned hand-over to the filtered recurrence
            if (f32_seq) {
                f32_seq = false;   // the planned hand-over to fp64, not counted as a restart
                // a start vector this good makes the pipelined beta a difference of nearly equal terms (it would report
                // a breakdown at once, seen at config 2): continue with the classic two-kernel recurrence there
                if (res * tiny_l < 1e-4 * std::fabs(lam)) classic = true;
                continue;
            }
            ++restarts;
            classic = true;   // refine with the accurate form (see enqueue_classic)
            if (restarts > 64) break;
        }
        // (what most steps of this solve were: a restart's classic tail does not change that)
        last_mode = steps_lowp > 0 ? 8 : pmode ? 4 : n <= OPT(classic_n, 256) ? 5 : pp.variant == kPanel ? 2 : pp.variant == kEll ? 3 : 1;
        if (switch_out) {        // (no Ritz vector was formed: have_prev keeps its value; the caller continues with the exact mode)
            last_steps = steps_used; last_steps_lowp = steps_lowp;
            return kSwitchToExact;
        }
        have_prev = true;
        last_steps = steps_used;
        last_steps_lowp = steps_lowp;
        if (warm_probe && start_overlap >= 0.0) {
            if (start_overlap < 2.0 / std::sqrt((double)n)) { warm_skip = std::min(63, kWarmSkip << std::min(warm_fails, 4)); ++warm_fails; }    // (a random unit vector's overlap is ~1/sqrt(n))
            else warm_fails = 0;
        }
        if (OPT(debug, 0) && warm_probe) fprintf(stderr, "[machip] warm start: overlap of the previous vector with the result %.3f%s\n", start_overlap, warm_skip ? " -> the next warm requests start cold (landscape-weighted)" : "");
        if (!(ev1_at_check && status == MACHIP_OK)) {     // (a converged Lanczos solve ends with its explicit check: ev1 is there)
            HIP_TRY(hipEventRecord(ev1, stream));
            HIP_TRY(hipEventSynchronize(ev1));
        }
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        *lambda2 = lam;
        if (steps_timed_acc >= 16 && last_mode >= 1 && last_mode <= 3) {     // (fused forms: gather, panel, padded)
            const int vk = std::max(0, std::min(7, (int)pp.variant));
            launch_us_hist[vk] = 1e3 * step_ms_acc / (double)steps_timed_acc / (pp.variant == kPanel ? 2.0 : 1.0);
        }
        if (stats) {
            stats->lanczos_steps = steps_used;      // (streamed records: up to the analysis point the solve ended at; steps_timed counts what was launched)
            stats->spmv_total = spmv_total;
            stats->vec_passes = steps_total * 7;   // per step and row: Z gathered (2) + Z own-row read (2) + Z written (2) + V written (1)
            stats->restarts = restarts;
            stats->nnz = nnz;
            stats->residual = res;
            stats->lnorm = lnorm;
            stats->gpu_ms = ms;
            stats->step_ms = step_ms_acc;
            stats->steps_timed = steps_timed_acc;
            stats->steps_lowp = steps_lowp;
            stats->drift = last_amp;
        }
        if (status == MACHIP_OK && lam < 1e-12 * tiny_l)
            return fail(MACHIP_DISCONNECTED, "lambda_2 ~ 0: the graph is not connected");
        if (status != MACHIP_OK)
            return fail(MACHIP_NOT_CONVERGED, "Lanczos hit the step cap before the residual test passed");
        return MACHIP_OK;
    }

    // q Ritz vectors (column-major n x q) of the last Krylov sequence -> host buffer.  Column 0 is
    // the converged Fiedler vector in yvec; the others are the next Ritz vectors, orthonormalised
    // on the host; copies produced by Lanczos "ghost" Ritz values are skipped.
    int ritz_block(int q, double* X_host) {
        const int Jeff = std::max(1, std::min(J_last, (int)ha.size()));
        const int ncand = std::min(Jeff, q + 6);
        std::vector<double> th, S;
        if (!last_was_lob && !last_seq_f32 && !last_seq_sharded) tri::smallest_block(ha.data(), hb.data(), Jeff, ncand, th, S, wk);   // (no Krylov basis after the preconditioned mode: X is completed with orthonormal filler)
        const int g2 = vgrid();
        HIP_TRY(hipMemcpyAsync(X_host, yvec, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int acc = 1;
        std::vector<double> col((size_t)n);
        auto orth_accept = [&](double n0) -> bool {
            for (int pass = 0; pass < 2; ++pass)
                for (int p = 0; p < acc; ++p) {
                    const double* xp = X_host + (size_t)p * n;
                    double dot = 0.0;
                    for (int i = 0; i < n; ++i) dot += xp[i] * col[(size_t)i];
                    for (int i = 0; i < n; ++i) col[(size_t)i] -= dot * xp[i];
                }
            double n2 = 0.0;
            for (int i = 0; i < n; ++i) n2 += col[(size_t)i] * col[(size_t)i];
            if (!(n2 > 1e-6 * std::max(n0, 1e-300))) return false;   // ghost copy / dependent
            const double inv = 1.0 / std::sqrt(n2);
            double* dst = X_host + (size_t)acc * n;
            for (int i = 0; i < n; ++i) dst[i] = col[(size_t)i] * inv;
            ++acc;
            return true;
        };
        for (int c = 1; c < (int)th.size() && acc < q; ++c) {
            memcpy(h_pin, S.data() + (size_t)c * Jeff, sizeof(double) * (size_t)Jeff);
            HIP_TRY(hipMemcpyAsync(sdev, h_pin, sizeof(double) * (size_t)Jeff, hipMemcpyHostToDevice, stream));
            const int KS = std::max(1, std::min(ks_max, Jeff / 8));
            k_ritz_partial<<<dim3(g2, KS), kBlock, 0, stream>>>(V, n, Jeff, sdev, ypart);
            k_ritz_combine<<<g2, kBlock, 0, stream>>>(ypart, n, KS, y_raw, part_c);
            HIP_TRY(hipMemcpyAsync(col.data(), y_raw, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            double mean = 0.0, n0 = 0.0;
            for (int i = 0; i < n; ++i) mean += col[(size_t)i];
            mean /= n;
            for (int i = 0; i < n; ++i) { col[(size_t)i] -= mean; n0 += col[(size_t)i] * col[(size_t)i]; }
            (void)orth_accept(n0);
        }
        // Krylov space exhausted (tiny or highly symmetric graph): complete the block with
        // deterministic vectors orthogonal to 1 and to the accepted columns, so X is always an
        // orthonormal n x q block like the reference's (nx:238).
        for (unsigned long long seed = 1; acc < q && seed < 64; ++seed) {
            double mean = 0.0, n0 = 0.0;
            for (int i = 0; i < n; ++i) {
                unsigned long long z = seed * 0xD1B54A32D192ED03ull + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z = z ^ (z >> 31);
                col[(size_t)i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
                mean += col[(size_t)i];
            }
            mean /= n;
            for (int i = 0; i < n; ++i) { col[(size_t)i] -= mean; n0 += col[(size_t)i] * col[(size_t)i]; }
            (void)orth_accept(n0);
        }
        for (; acc < q; ++acc) {
            double* dst = X_host + (size_t)acc * n;
            for (int i = 0; i < n; ++i) dst[i] = 0.0;
        }
        return MACHIP_OK;
    }
};

}  // namespace machip
