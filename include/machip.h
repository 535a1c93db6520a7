/*
 * machip.h -- C ABI of libmachip.so, the MI355X (gfx950) implementation of the
 * Frank-Wolfe / Fiedler hot path of MarineRoboticsGroup/mac.
 *
 * Plain C, no torch / numpy types: pointers + sizes only.  Host pointers are
 * borrowed for the duration of a call; all device memory is owned by the
 * handle.  Every function returns a machip_status (0 = OK) and never throws;
 * machip_last_error() gives the text of the last failure on this thread.
 * A handle is not thread-safe; distinct handles are independent (one HIP
 * stream each).
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the reference repository; "nx:" = networkx 3.4.2
 * networkx/linalg/algebraicconnectivity.py, the third-party module the
 * reference delegates the eigen-solve to).  The reference-side binding is
 * shown in INTEGRATION.md.
 */
#ifndef MACHIP_H
#define MACHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MACHIP_ABI_VERSION 6   /* 6: machip_solve_stats.drift, machip_panel_plan fills 12 entries; 5: per-handle option table (machip_set_option), machip_comm_drop_ipc; 4: inter-process communicator */

typedef enum machip_status {
    MACHIP_OK = 0,
    MACHIP_NOT_CONVERGED = 1, /* iteration cap hit (the reference would spin, nx:236)     */
    MACHIP_DISCONNECTED = 2,  /* lambda_2 ~ 0: graph not connected (reference: SuperLU
                                 "Factor is exactly singular" RuntimeError)               */
    MACHIP_BAD_ARG = 3,       /* reference: AssertionError (mac.py:47,52,183; fiedler.py:35) */
    MACHIP_HIP_ERROR = 4,
    MACHIP_RCCL_ERROR = 5,
    MACHIP_NO_DEVICE = 6
} machip_status;

typedef struct machip_problem machip_problem; /* one MAC instance (mac/solvers/mac.py:16)  */

/* Statistics of the last eigen-solve (n_mv, n_vec counters). */
typedef struct machip_solve_stats {
    int64_t lanczos_steps; /* Lanczos steps the solve used: up to the analysis point it ended at
                            * (a function of the records alone, round 5); steps_timed counts the
                            * launches, a few more when the queue ran past that point              */
    int64_t spmv_total;    /* + SpMVs of the explicit residual checks                      */
    int64_t vec_passes;    /* length-n vector reads+writes outside the SpMV                */
    int64_t restarts;
    int64_t nnz;           /* nnz of the assembled L(x) (diagonal included)                */
    int64_t support;       /* |{k : x_k > min_selection_weight_tol}|                       */
    double residual;       /* ||L v - lambda v||_1 / ||L||_inf  (the reference's stop rule,
                              nx:232,246)                                                  */
    double lnorm;          /* ||L||_inf                                                    */
    double gpu_ms;         /* device time of the solve (hipEvents on the handle's stream)  */
    double step_ms;        /* ... of which the Krylov chunks alone (events bracketing the
                              step kernels; explicit checks / Ritz vector excluded)        */
    int64_t steps_timed;   /* steps covered by step_ms: step_ms / steps_timed = in-solve
                              duration of one fused step launch                            */
    int64_t steps_lowp;    /* of lanczos_steps, those run with fp32 storage
                              (machip_set_precision(1)); the rest are fp64                 */
    double drift;          /* column-panel steps with an 8-byte operand (mac_amd/csrc/panel_u.h):
                              largest accumulated factor by which a rounding error of the
                              recurred product L v_j was carried forward, from the tridiagonal
                              records (0: the solve took no such step); beyond option
                              "panel_u_amp" (1e5) the sequence ends and the solve goes on
                              in the two-kernel form                                        */
} machip_solve_stats;

int machip_version(void);
int machip_sizeof_stats(void);             /* sizeof(machip_solve_stats): binding layout check */
int machip_device_count(void);            /* 0 when no GPU is visible                      */
const char* machip_last_error(void);

/* MAC.__init__ (mac/solvers/mac.py:22-72): fixed + candidate edge lists (SoA,
 * int32 node ids in [0,n)), n nodes.  Builds the union sparsity pattern once and
 * keeps weights / edge lists resident in HBM.  Duplicate pairs are summed exactly
 * as coo->csr does (mac/utils/graphs.py:48,98). */
int machip_create(int device, int64_t n,
                  int64_t n_fixed, const int32_t* fi, const int32_t* fj, const double* fw,
                  int64_t m, const int32_t* ci, const int32_t* cj, const double* cw,
                  double min_selection_weight_tol, machip_problem** out);
void machip_destroy(machip_problem* p);

/* x <- host vector (m doubles); x -> host.  (The reference passes x by value
 * into MAC.laplacian / problem / evaluate_objective, mac.py:74,91,104.) */
int machip_set_x(machip_problem* p, const double* x);
int machip_get_x(machip_problem* p, double* x);

/* MAC.laplacian(x) (mac.py:74-89): assemble L(x) = L_fixed + sum_{x_k>tol} x_k w_k L_k
 * on the device as CSR.  nnz_out may be NULL. */
int machip_assemble(machip_problem* p, int64_t* nnz_out);
/* Copy the assembled CSR to the host (indptr n+1, indices nnz, data nnz; the
 * diagonal is the first entry of each row, the rest sorted by column). */
int machip_get_laplacian(machip_problem* p, int32_t* indptr, int32_t* indices, double* data);

/* find_fiedler_pair on the assembled L(x) (mac/utils/fiedler.py:9-44 ->
 * nx:151-256).  Stop rule identical to the reference: ||L v - lambda v||_1 /
 * ||L||_inf < tol.  x0: optional start vector (n doubles, host) -- the drop-in
 * passes column 0 of the reference's RandomState(7) block; NULL + warm_start!=0
 * reuses the previous Fiedler vector resident on the device (the working form of
 * MAC.Cache, mac.py:17-20); NULL + warm_start==0 uses a fixed device-side
 * pseudo-random vector.  v_out (n) and X_out (n*q, column-major) may be NULL;
 * q in [1,4] columns are produced when X_out != NULL.  ONLY COLUMN 0 OBEYS THE STOP RULE
 * (it is v_out).  Columns 1..q-1 are what the reference's block X carries along (fiedler.py:44,
 * nx:238) -- there q vectors converge together; here they are the next Ritz vectors of the LAST
 * Krylov sequence of the solve, orthonormalised against column 0 and against 1: guaranteed
 * orthonormal, orthogonal to the constant vector, with Rayleigh quotients >= lambda_2; how close
 * they are to v_3, v_4, ... depends on that sequence alone -- after a few hundred steps they are
 * accurate to 1e-3 .. 1e-7 (tests: er2000_x0), after a short sequence (a restart from an almost
 * converged vector, 33 steps on er300_x0) or after the preconditioned / exact modes, which keep
 * no Krylov basis, they are an arbitrary orthonormal completion.  A caller that needs more than
 * the Fiedler pair must not take them for eigenvectors. */
int machip_fiedler(machip_problem* p, double tol, int max_steps, const double* x0, int warm_start,
                   double* lambda2, double* v_out, double* X_out, int q,
                   machip_solve_stats* stats);

/* Store the cold-start vector (n doubles) used by every later machip_fiedler /
 * machip_fw_step call that passes x0 = NULL and warm_start = 0: the drop-in sends the
 * first column of the reference's RandomState(7) block (fiedler.py:27-32) once. */
int machip_set_start(machip_problem* p, const double* x0);

/* Supergradient g_k = (w_k (v_i - v_j)) (v_i - v_j) for all m candidates from the
 * device-resident Fiedler vector (mac.py:117-124).  g_out (m) may be NULL. */
int machip_gradient(machip_problem* p, double* g_out);

/* solve_subset_box_lp(g, k) (mac/optimization/constraints.py:12-22 ->
 * mac/utils/rounding.py:21-28): s = indicator of the k largest g.  Ties at the
 * k-th value go to the lowest indices (the reference's argpartition leaves them
 * unspecified).  s_out (m doubles) may be NULL. */
int machip_lp_topk(machip_problem* p, int64_t k, double* s_out);

/* One Frank-Wolfe iteration, device-resident (mac/optimization/frankwolfe.py:53-76
 * with problem = MAC.problem, solve_lp = solve_subset_box_lp):
 *   assemble L(x); (f, v) = Fiedler pair; g = supergradient; s = top-k(g);
 *   dual = f + g.(s - x); gnorm = ||g||_2; x_next = x + 2/(iter+2) (s - x).
 * x_next is staged; machip_fw_commit() makes it current (the caller applies the
 * reference's stop tests first: on a stop the reference returns the pre-update
 * x, frankwolfe.py:65-74).  Only scalars cross PCIe. */
int machip_fw_step(machip_problem* p, int64_t k, int iter, double tol, int max_steps, int warm_start,
                   double* f, double* dual, double* gnorm, machip_solve_stats* stats);
int machip_fw_commit(machip_problem* p);
/* The loop itself (mac/optimization/frankwolfe.py:53-76 as mac/solvers/mac.py:196-200 calls it) without returning to the caller
 * between iterations: for i = first_iter .. first_iter + max_iters - 1: machip_fw_step; upper = min(upper, dual); stop when
 * ||g|| < grad_tol (the current x stays) or (upper - f) < gap_tol |f|; else machip_fw_commit.  *upper_inout carries the dual
 * bound in and out (start: +inf).  Optional per-iteration outputs (max_iters entries each, NULL to skip): f_traj, dual_traj,
 * gnorm_traj, stats, modes (2 ints per iteration: machip_solve_mode and its closure count).  *iters_done = iterations run.
 * Between two iterations a Python caller of machip_fw_step / machip_fw_commit leaves the GPU idle for ~50 us (measured,
 * profiles/r5_c4_gaps.txt); this entry point is what MAC.solve and bench.py drive. */
int machip_fw_run(machip_problem* p, int64_t k, int first_iter, int max_iters, double gap_tol, double grad_tol, double tol,
                  int max_steps, int warm_start, double* upper_inout, double* f_traj, double* dual_traj, double* gnorm_traj,
                  machip_solve_stats* stats, int* modes, int* iters_done);

/* round_nearest(w, k, weights, break_ties_decimal_tol) on the device-resident x
 * (mac/utils/rounding.py:7-42, called at mac/solvers/mac.py:209): indicator of the k largest
 * entries under the lexicographic key (round(x, decimals), candidate weight); decimals < 0 =
 * plain top-k of x (rounding.py:21-28).  Full ties (equal rounded x and equal weight) go to the
 * highest indices (the reference's argpartition leaves them unspecified).  rounded_out: m doubles. */
int machip_round_nearest(machip_problem* p, int64_t k, int decimals, double* rounded_out);

/* find_fiedler_pair(L) for an arbitrary scipy CSR Laplacian
 * (mac/utils/fiedler.py:9, called that way by tests/utils/test_fiedler.py:32). */
int machip_fiedler_csr(int device, int64_t n, const int32_t* indptr, const int32_t* indices,
                       const double* data, double tol, int max_steps, const double* x0,
                       double* lambda2, double* v_out, double* X_out, int q,
                       machip_solve_stats* stats);

/* y = L(x) v on the device CSR (host vectors, n doubles): parity probe for the
 * SpMV kernel. variant: 0 = auto, 1 = LDS row-tile ("stream"), 2 = sub-wave vector. */
int machip_spmv(machip_problem* p, const double* v, double* y, int variant);

/* Landscape of L(x) behind the cold start of a Lanczos solve (late round 5): u after `sweeps` Jacobi sweeps
 * u <- u + D^-1 (1 - L u) from u = D^-1 (n doubles to the host).  A cold start multiplies the start vector
 * (machip_set_start: the reference's RandomState(7) column, mac/utils/fiedler.py:27-32) entry by entry by
 * (u / max u)^p -- options start_land (sweeps, default 3, 0 = off) and start_pow (p, default 128): the Fiedler
 * vector of a sparse random graph is localised on the peaks of u, and the reference's start block is only a
 * first guess (nx:151-256 iterates it to the same stop rule).  A caller's own start vector (x0 of
 * machip_fiedler / machip_fiedler_csr: the X argument of find_fiedler_pair) and a warm start are never
 * touched.  Parity probe for the k_land_* kernels and a diagnostic. */
int machip_landscape(machip_problem* p, int sweeps, double* u_out);

/* Average duration (microseconds, hipEvents on the handle's stream) of `reps`
 * back-to-back launches of the fused Lanczos SpMV kernel on the assembled L(x),
 * and its algorithmic bytes per launch (SURVEY section 8(d) B_spmv + the fused vector
 * traffic).  Used by bench.py for the roofline object. */
int machip_profile_spmv(machip_problem* p, int reps, double* avg_us, double* bytes_per_launch);

/* Multi-GPU: candidates are sharded in contiguous ranges over
 * nranks processes (one GPU each); each rank evaluates the supergradient of its
 * range and one RCCL all-gather over xGMI rebuilds the full m-vector on every
 * rank.  unique_id: the 128-byte ncclUniqueId from machip_comm_unique_id() on
 * rank 0, distributed by the caller. */
int machip_comm_unique_id(void* id128);
int machip_comm_init(machip_problem* p, int rank, int nranks, const void* id128);
/* The same split inside ONE process (the ncclCommInitAll style of SURVEY section 8(e)): `nranks` handles of the
 * same problem -- one per GPU, or several on one GPU -- become ranks 0..nranks-1 of an in-process communicator;
 * each handle is then driven by its own host thread and machip_fw_step / machip_gradient on every handle is
 * collective (all handles must call it).  The all-gather is peer-to-peer device copies between the handles'
 * gradient buffers bracketed by a host barrier; shard arithmetic and call site are those of the RCCL path.
 * Destroying one handle releases peers blocked in the collective with MACHIP_RCCL_ERROR. */
int machip_comm_init_local(machip_problem** handles, int nranks);
/* Communicator between PROCESSES (one per GPU: what `bench.py --gpus N` / torch.distributed.run launch) whose EIGEN-SOLVE is
 * row-partitioned (SURVEY section 8(e) "Expected scaling ... unless the eigen-solve is also parallelised"; the reference is a
 * single process, there is no reference line to match).  Every rank exports IPC handles of its Lanczos record buffers,
 * partial sums, Ritz staging vector, gradient and flag words (machip_ipc_export -> a blob of machip_ipc_blob_bytes() bytes), the
 * caller distributes the blobs (mac_amd.dist.FileGroup), every rank maps the peers' buffers (machip_comm_init_ipc: blobs =
 * nranks blobs in rank order; hipIpcOpenMemHandle, peer access across devices).  A rank then launches its share of every
 * fused Lanczos step's workgroups and writes records / partial sums into every rank's copy; steps are ordered ON THE
 * DEVICE by flag words (no host, no cross-stream edge; bounded waits: a stalled or dead peer makes the call return
 * MACHIP_RCCL_ERROR after timeout_s, default 10); every rank runs the same deterministic host logic, so results are bit-identical
 * to a single rank's.  Composes with machip_comm_init: with an RCCL communicator the gradient shards travel by ncclAllGather,
 * without one (ranks sharing a GPU) by peer writes through the mapped buffers.  machip_comm_close_ipc before machip_destroy
 * marks an orderly exit (a handle destroyed without it raises abort on its peers). */
/* First-contact helpers (round 4).  machip_comm_init and the communicator's FIRST ncclAllGather run under a watchdog
 * (MACHIP_RCCL_TIMEOUT_S, default 600 s): a peer that never arrives / a fabric that cannot carry the collective returns
 * MACHIP_RCCL_ERROR naming the rank instead of hanging the job.  machip_selftest_watchdog exercises that watchdog with a
 * sleeping stand-in (no GPU needed: MACHIP_OK when work_ms < limit_ms, MACHIP_RCCL_ERROR otherwise); machip_peer_access =
 * hipDeviceCanAccessPeer (1 / 0, -1 on error), what `bench.py --gpus N --dry` prints as a matrix. */
int machip_selftest_watchdog(int work_ms, int limit_ms);
int machip_peer_access(int device_a, int device_b);
int machip_ipc_blob_bytes(void);
int machip_ipc_export(machip_problem* p, void* blob, int blob_bytes);
int machip_comm_init_ipc(machip_problem* p, int rank, int nranks, const void* blobs, double timeout_s);
int machip_comm_close_ipc(machip_problem* p);
/* Leave the inter-process communicator again (mappings closed, flag words freed, rank 0 of 1): the way back to a replicated
 * solve when the attach succeeded here but failed on a peer (mac_amd.dist.attach_ipc).  Collective in spirit: call it on
 * every rank, behind a barrier, before anybody launches another step. */
int machip_comm_drop_ipc(machip_problem* p);
/* Leave whatever inter-process communicator the handle belongs to (RCCL: ncclCommAbort; IPC: as above): the handle is a
 * single-rank handle again.  For a first contact that failed on SOME rank: every rank drops, behind an agreement exchange. */
int machip_comm_drop(machip_problem* p);
/* Device time of the last sharded gradient of this handle (hipEvents on its stream): the gradient kernel over this rank's
 * candidate range, and the exchange behind it (ncclAllGather, or the IPC publish / wait pair) -- SURVEY section 8(e)'s two
 * terms, measured per Frank-Wolfe iteration by bench.py --gpus N. */
int machip_comm_timing(machip_problem* p, double* grad_us, double* exchange_us);
/* Which launch group the steps of the last eigen-solve were: 1 fused gather step (k_pipe_vec), 2 column-panel step
 * (k_pan_mul + k_pan_fin), 3 padded fixed-width step, 4 single-workgroup kernel (k_lan_persist), 5 classic two-kernel step,
 * 6 LOBPCG preconditioned by the chain, 7 exact chain + closures mode (*closures = size of its capacitance matrix),
 * 8 fp32 sequence + fp64 refinement.  bench.py prices the roofline of THAT group. */
int machip_solve_mode(machip_problem* p, int64_t* closures);
/* With 2..8 handles the in-process communicator also ROW-PARTITIONS THE EIGEN-SOLVE (MACHIP_SHARD_EIG=0 turns that off):
 * inside machip_fw_step rank 0's solver drives every rank's stream; per Lanczos step each rank launches its share of
 * the step's workgroups on its own copy of L(x) and of the gather operand and writes the records / partial sums it
 * produces into every rank's copy (peer-mapped device pointers across xGMI; plain device memory when ranks share a
 * GPU), steps ordered by HIP events; the Krylov basis is sharded by rows.  Workgroup w of the partitioned step does
 * exactly what workgroup w of a single rank's launch does, so lambda_2, the Fiedler vector and everything after them
 * are bit-identical to a single-rank run.
 * machip_comm_mode: 0 = no communicator, 1 = RCCL (candidate shard, eigen-solve replicated), 2 = in-process with a
 * replicated eigen-solve, 3 = in-process with the row-partitioned eigen-solve, 4 = inter-process (machip_comm_init_ipc) with
 * the row-partitioned eigen-solve enabled, 5 = ... and the LAST eigen-solve really ran row-partitioned (the fused Lanczos step;
 * other solver modes run replicated on every rank), 6 = inter-process, gradient exchange only (MACHIP_SHARD_EIG=0). */
int machip_comm_mode(machip_problem* p);
/* The candidate range [lo, hi) of `rank` and the padded shard length (host arithmetic only, no GPU needed):
 * shard = ceil(m / nranks), lo = min(m, rank shard), hi = min(m, lo + shard). */
int machip_shard_plan(int64_t m, int nranks, int rank, int64_t* lo, int64_t* hi, int64_t* shard);

/* MAC.evaluate_objective for B selection vectors at once (mac/solvers/mac.py:91-102 as called in a loop by
 * round_madow(value_fn=evaluate_objective, max_iters > 1), mac/utils/rounding.py:63-75, and by the budget sweep of
 * examples/g2o_experiment.py:347-376).  X: B x m row-major on the host; lambda2: B doubles; status (may be NULL): B
 * machip_status values (OK / NOT_CONVERGED / DISCONNECTED per entry).  The handle keeps up to MACHIP_LANES (default 16 for
 * single-workgroup solves, 4 otherwise) evaluation lanes -- own x, CSR buffers, eigen-solver state and stream, sharing the pattern and the candidate
 * arrays -- driven by one host thread each, so the small latency-bound solves of a pose graph overlap on the GPU.
 * Cold starts from the handle's start vector (machip_set_start), solver mode / precision of the handle; the
 * handle's own x, gradient and Fiedler vector are not touched.  Returns the first hard error, else MACHIP_OK. */
int machip_eval_batch(machip_problem* p, int B, const double* X, double tol, int max_steps, double* lambda2,
                      int* status);

/* Frank-Wolfe on B problems of the same graph AT ONCE -- the reference's real workload is a sweep of budgets over one
 * pose graph (examples/g2o_experiment.py:306-336: 10 budgets x MAC.solve(max_iters = 20)), each a chain of small,
 * latency-bound launches that leaves most of the chip idle.  Problem b runs the loop of
 * mac/optimization/frankwolfe.py:53-76 as mac/solvers/mac.py:196-200 calls it -- problem = MAC.problem, solve_lp =
 * top-ks[b], gamma_i = 2/(i+2), dual bound from the pre-update x, stop when ||g|| < grad_tol or
 * (upper - f) < gap_tol |f|, at most max_iters iterations -- from X0[b] (B x m row-major, host), on one of the handle's
 * evaluation lanes (own x / gradient / CSR / eigen-solver state / stream, sharing pattern and candidate arrays; up to
 * MACHIP_LANES run concurrently, one host thread each, each stream on a hardware queue of its own).  Every problem starts
 * from a clean solver state and the handle's start vector: its results are bit-identical to a fresh handle running
 * machip_fw_step / machip_fw_commit in a loop IN THE SAME SOLVER MODE (machip_set_solver(1), the Lanczos path, on both: under
 * the automatic mode a standalone handle may take the exact chain + closures mode -- small pose graphs up to 700 closures, larger
 * ones on a long forecast -- which a lane keeps to 256 closures so as not to starve the other lanes; the two then agree to the
 * solver tolerance, not bit for bit), whatever lane takes it -- as long as no single Krylov sequence outgrows the
 * lane's basis (a lane holds MACHIP_LANE_VBUDGET_MB = 1/8 of the handle's basis budget: 6 710 columns against 10 010 at
 * n = 1e4, 671 against 5 368 at n = 1e5); a longer sequence restarts earlier on a lane than on the handle and then agrees
 * with it to the solver tolerance, not bit for bit.
 * Outputs (host): X_out B x m final relaxed x (what MAC.solve returns as `unrounded`); R_out (may be NULL) B x m
 * round_nearest(x, ks[b], weights, round_decimals) as machip_round_nearest computes it; upper[B] dual upper bounds;
 * f_traj (may be NULL) B x max_iters lambda_2 per iteration; iters[B] iterations done; status[B] per problem
 * (OK / NOT_CONVERGED / DISCONNECTED).  Returns the first hard error, else MACHIP_OK. */
int machip_fw_sweep(machip_problem* p, int B, const int64_t* ks, const double* X0, int max_iters, double gap_tol,
                    double grad_tol, double tol, int max_steps, int warm_start, int round_decimals, double* X_out,
                    double* R_out, double* upper, double* f_traj, int* iters, int* status);

/* Eigen-solver selection -- the reference's `fiedler_method` string (mac/solvers/mac.py:23,68;
 * mac/utils/fiedler.py:38-42 dispatches 'tracemin_pcg' | 'tracemin_lu' | 'tracemin_cholesky', all
 * computing the same pair).  mode 0 = automatic (default), 1 = Lanczos on L restricted to 1-perp,
 * 2 = preconditioned (LOBPCG with a tridiagonal odometry-chain solve; the drop-in maps
 * 'tracemin_pcg' here).  Mode 2 falls back to mode 1 when it does not apply (n <= 256) or
 * stagnates; results obey the same stop rule either way. */
int machip_set_solver(machip_problem* p, int mode);

/* Arithmetic of the Krylov iterate (BASELINE.json configs[4], SURVEY section 8(b) `precision`):
 * 0 = fp64 throughout (default, what the reference computes in, mac/utils/fiedler.py:27-44);
 * 1 = fp32 Lanczos iterate (fp32 matrix values, 8-byte {t, v} float2 gather records, fp64
 *     accumulation of every inner product) followed by fp64 refinement: the Ritz vector is
 *     re-orthogonalised, its Rayleigh quotient and the reference's residual test are evaluated in
 *     fp64 on the fp64 L(x), and fp64 steps continue from it until the test passes.  The returned
 *     pair therefore obeys the same stop rule and the same 1e-8 parity bound as mode 0. */
int machip_set_precision(machip_problem* p, int precision);

/* Per-handle option table (round 5; replaces the MACHIP_* environment knobs of rounds 1-4).  The reference's only selector is
 * one string, `fiedler_method` (mac/utils/fiedler.py:38-42 -> machip_set_solver above); everything else this library can
 * vary -- launch shapes, thresholds of the automatic mode, test hooks that force a code path onto a small graph -- is an entry
 * of the handle's table, named as listed by machip_option_name(i) (mac_amd/csrc/options.h): e.g. "panel" (-1 automatic, 0 off,
 * 1 forced), "chunk", "lanes", "woodbury".  value = MACHIP_OPTION_AUTO restores the measured default.
 * p == NULL edits the PROCESS defaults: what handles created afterwards (and the handle machip_fiedler_csr keeps) start
 * from.  Those defaults are initialised from the environment once, when the library is first used (MACHIP_<NAME>, developer
 * use: sweeps under tools/); nothing reads the environment after that.  Options consumed when a handle is created
 * ("asm_g", "vbudget_mb", "vcap", ...) must be set as process defaults before machip_create.  An evaluation lane runs on its
 * owner's table.  Unknown name -> MACHIP_BAD_ARG.  Not thread-safe against a running call on the same handle. */
#define MACHIP_OPTION_AUTO INT64_MIN
int machip_set_option(machip_problem* p, const char* name, int64_t value);
int machip_get_option(machip_problem* p, const char* name, int64_t* value);   /* MACHIP_OPTION_AUTO when not set */
const char* machip_option_name(int i);    /* i = 0, 1, ...; NULL past the last one */

int machip_synchronize(machip_problem* p);

/* Measurement helpers (no counterpart in the reference; bench.py and the CPU tests use them).
 * machip_membench: achieved HBM bandwidth on `device` in GB/s -- a read-only pass and the STREAM triad
 * a = b + s c (two reads + one write per element) over arrays of `bytes` bytes each, `reps` back-to-back
 * launches timed with HIP events.  SURVEY section 8(d): the roofline object of bench.py reports this measured
 * peak next to the nominal 8 TB/s.
 * machip_host_tridiag_smallest: host only, no GPU -- smallest eigenpair (theta, s[J]) of the J x J symmetric
 * tridiagonal with diagonal a[0..J) and off-diagonal b[1..J) (b[0] unused): the O(J) analysis the Lanczos driver
 * runs on every chunk of steps; exported so that `-m "not gpu"` tests can check it against LAPACK. */
int machip_membench(int device, int64_t bytes, int reps, double* read_gbs, double* triad_gbs);
/* Host only, no GPU: the shape the column-panel Lanczos step (mac_amd/csrc/panel.h) would use for a matrix of n rows, nnz
 * entries and longest row maxlen -- out12 = {on, NP panels, C columns per panel, NB row blocks, NTB 64-row tiles per block,
 * TWW tiles per worker wave, RPT records per worker thread (record form), workgroups of the row kernel, u (1: the shifted
 * recurrence with an 8-byte operand, mac_amd/csrc/panel_u.h), LPT 16-byte operand loads per worker thread, TWT tiles per worker
 * wave its kernel instantiation holds, row blocks per workgroup} -- under the process-default options ("panel", "panel_u" etc.).
 * For CPU tests of the shape arithmetic (coverage of all rows / columns, LDS and register limits). */
int machip_panel_plan(int64_t n, int64_t nnz, int maxlen, int* out12);
/* machip_fiedler_csr keeps one CSR-only handle (stream, device buffers, chunk graphs) between calls and reuses it when
 * device and n match and the matrix fits -- every solve still starts from a clean solver state.  This frees it (the
 * Python layer calls it at interpreter exit). */
void machip_release_cache(void);
int machip_host_tridiag_smallest(const double* a, const double* b, int J, double* theta, double* s);
/* Host only, no GPU: the rule a Lanczos solve ends by (mac_amd/csrc/follow.h; round 5, streamed records) applied to a finished
 * record array.  tri3 = interleaved (alpha_j, beta_j, ||v_j||_1) triples valid through beta_J; e_target = the residual estimate
 * (||r||_1, not yet divided by ||L||_inf) a check needs; tiny_l = ||L||_inf; jcap = basis capacity (0: none).  Out: the analysis
 * points visited (a function of the records alone), the order of the tridiagonal the sequence ends on (-1: not within J) and
 * the estimate there.  The reference's counterpart is the per-iteration test of nx:246. */
int machip_host_follow_records(const double* tri3, int J, int n, double e_target, double tiny_l, int jcap, int* points, int cap,
                               int* npoints, int* jeff, double* est);

#ifdef __cplusplus
}
#endif
#endif /* MACHIP_H */
