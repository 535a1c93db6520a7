#!/usr/bin/env python3
"""bench.py -- Frank-Wolfe iterations/second (each including the full Fiedler solve) of the
MAC hot path on MI355X, with the CPU paths timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4|c2|c3|c5a|c5b|c5|c5s|c4s|c2s] [--mode all|shard|replicas|ipc_eig]

One "step" = one Frank-Wolfe iteration (mac/optimization/frankwolfe.py:53-76 with
problem = MAC.problem): assemble L(x) -> Fiedler pair to the reference's stop rule at tol 1e-8 ->
supergradient of all m candidates -> top-K LP -> dual bound / norms -> x update.  Inputs (edge
lists, x0, start vector) are resident in HBM before the timed region starts.

Default workload (N = 1): BASELINE.json configs[3], the configuration the north star quotes its target on
(ER N = 100 000, ~2M candidate edges, chain fixed, K = 10 %); it fits one GPU.  The timed region is a PASS of
exactly K iterations from x0; passes are repeated until >= --min-seconds of wall time have been measured and
the MEDIAN pass is reported (`repeats`, `pass_ms`), so the figure is reproducible and visible from outside.

Every cold eigen-solve starts from the reference's start column (fiedler.py:27-32) multiplied by the landscape weighting of DESIGN
section 4.2 (inside the timed region: five launches per solve); the object `cold_start` of the line holds the same pass with the
weighting off and -- configs[1] / configs[3] -- both settings on the reference's own 20 iterates (tests/golden).  `warm_start`: the
same pass with use_cache=True made real.

--gpus N > 1 (default --mode all): ONE invocation runs, in one process group and in this order,
  * the SHARD pass (headline `value`, "scaling": "strong"; the north star's split): the candidates
    sharded over the ranks, every rank evaluates the supergradient of its contiguous range, one ncclAllGather rebuilds the
    m-vector on every rank, everything else is replicated; hipEvent-timed gradient / exchange per iteration (`shard`);
  * the REPLICAS pass ("scaling": "weak"): every rank runs an independent problem of the same graph with its own budget K_r
    (the reference's budget sweep, examples/g2o_experiment.py:306-336), no collective; value = all ranks' iterations /
    max-over-ranks time (`replicas`);
  * the ROW-PARTITIONED eigen-solve pass: every rank launches its share of each fused Lanczos step, buffers IPC-mapped, steps
    ordered by device-side flags (`ipc_eig`: us per step);
  a leg whose first contact fails on any rank is dropped on every rank and reported under `errors`; the others still print.
  DESIGN section 7 states what each can and cannot scale.  --mode shard|replicas|ipc_eig runs one leg; --config c5: one pose
  graph per rank (replicas).
  Launched by `torch.distributed.run` (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the environment) or, when those
  are absent, by bench.py itself: it spawns N rank processes, one GPU each, and refuses to run when fewer
  than N devices are visible.

Prints ONE JSON line (rank 0).  No torch anywhere: ranks find each other through mac_amd.dist.FileGroup
(single node), the data path is libmachip.so (HIP + RCCL) via ctypes.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
import uuid

os.environ.setdefault("OMP_NUM_THREADS", "1")          # the CPU baselines are 1-core paths (SuperLU, ARPACK)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
KERNEL = ("one Lanczos step = k_pipe_vec (one kernel: CSR SpMV with gathered operand + all vector work) on sparse iterates, "
          "k_pan_mul8 + k_pan_finu (column-panel form, 8-byte operand of the shifted recurrence in LDS; mac_amd/csrc/panel_u.h) from ~11 entries per row at N >= 65536")
STEP_KERNELS = ("k_pipe_vec", "k_pipe_stream", "k_pan_mul", "k_pan_fin")     # launches that make up Lanczos steps
STEP_HEADS = ("k_pipe_vec", "k_pipe_stream", "k_pan_mul")                    # one of these per step
# reference-equivalent CPU path at sizes where it does finish (measured in the build
# container, 1 core): seconds per Fiedler solve of the same ER family -- the extrapolation points for configs[3]
# NOT measured on the node this bench runs on: constants from the 8-vCPU build container, reported under a key that says so;
# the same-node point is measured by every default run (cpu_same_node_point below)
REF_EXTRAPOLATION = {"N=20000 (m=200k cands)": 88.0, "N=40000 (m=800k cands)": 448.0, "N=100000": "> 3000 (did not finish in 50 min)"}


# ---------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------
def er_workload(n, p, seed, name):
    cache = os.path.join(tempfile.gettempdir(), f"machip_wl_{n}_{seed}_{os.getuid()}.npz")
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            ci, cj = z["ci"], z["cj"]
        except Exception:
            ci = None
    else:
        ci = None
    if ci is None:
        import networkx as nx
        G = nx.fast_gnp_random_graph(n, p, seed=seed)
        e = np.array([(min(a, b), max(a, b)) for a, b in G.edges() if abs(a - b) != 1], dtype=np.int32)
        ci, cj = e[:, 0].copy(), e[:, 1].copy()
        try:        # generation takes ~20 s at N = 100k; the PMC / rank children of one run reuse it
            tmp = cache + f".{os.getpid()}.tmp.npz"
            np.savez(tmp, ci=ci, cj=cj)
            os.replace(tmp, cache)
        except OSError:
            pass
    m = len(ci)
    k = m // 10
    x0 = np.zeros(m)
    x0[np.random.default_rng(0).choice(m, k, replace=False)] = 1.0
    fi = np.arange(n - 1, dtype=np.int32)
    return dict(name=name, n=n, fi=fi, fj=fi + 1, fw=np.ones(n - 1), ci=ci, cj=cj, cw=np.ones(m), k=k, x0=x0)


def g2o_workload(fname, name):
    from mac_amd.utils.g2o import read_g2o_edges, split_chain
    path = os.path.join(ROOT, "tests", "golden", "data", fname)
    i, j, kap, n = read_g2o_edges(path)
    fixed = split_chain(i, j)
    ci, cj, cw = i[~fixed].astype(np.int32), j[~fixed].astype(np.int32), kap[~fixed]
    m = len(cw)
    k = int(0.2 * m)
    x0 = np.zeros(m)
    x0[np.argpartition(cw, -k)[-k:]] = 1.0        # NaiveGreedy init (mac/solvers/baseline.py:10-13)
    return dict(name=name, n=n, fi=i[fixed].astype(np.int32), fj=j[fixed].astype(np.int32), fw=kap[fixed],
                ci=ci, cj=cj, cw=cw, k=k, x0=x0)


def make_workload(cfg):
    if cfg == "c2":
        return er_workload(10000, 0.01, 0, "configs[1]: ER N=10000 p=0.01 (nx.fast_gnp_random_graph seed 0), chain fixed, K=10%")
    if cfg == "c4":
        n = 100000
        return er_workload(n, 2.0e6 / (n * (n - 1) / 2), 0, "configs[3]: ER N=100000 ~2M candidates, chain fixed, K=10%")
    if cfg == "c3":
        return g2o_workload("intel.g2o", "configs[2]: intel.g2o odometry fixed, K=20% loop closures")
    if cfg == "c5a":
        return g2o_workload("sphere2500.g2o", "configs[4]a: sphere2500.g2o, K=20%")
    if cfg == "c5b":
        return g2o_workload("city10000.g2o", "configs[4]b: city10000.g2o, K=20%")
    raise SystemExit(f"unknown --config {cfg}")


def step_bytes(n, nnz):
    """Algorithmic bytes of ONE fused Lanczos step (SURVEY 8(d) B_spmv + the fused vector work):
    12 nnz (val f64 + col i32) + 4 (n+1) (rowptr) + 56 n (gather record counted once per row 16 B, own-row
    record 16 B, next record written 16 B, basis column written 8 B).  In the mixed-precision mode the
    records and values are 4-byte floats: 8 nnz + 4 (n+1) + 32 n."""
    return 12.0 * nnz + 4.0 * (n + 1) + 56.0 * n


# ---------------------------------------------------------------------------------------------
def run_pass(P, k, iters, x0, tol=1e-8):
    """iters Frank-Wolfe iterations from x0 with the stop tests disabled (device-resident loop)."""
    P.set_x(x0)
    if iters <= 0:
        return []
    r = P.fw_run(k, iters, tol=tol)               # the loop runs on the C side (machip_fw_run): no Python between iterations
    rec = []
    for it in range(r["iters"]):
        st = r["stats"][it]
        rec.append(dict(f=float(r["f"][it]), dual=float(r["dual"][it]), gnorm=float(r["gnorm"][it]), steps=int(st.lanczos_steps), nnz=int(st.nnz),
                        support=int(st.support), gpu_ms=float(st.gpu_ms), step_ms=float(st.step_ms), steps_timed=int(st.steps_timed),
                        steps_lowp=int(st.steps_lowp), residual=float(st.residual), mode=r["modes"][it]))
    return rec


# ---------------------------------------------------------------------------------------------
# CPU baselines (oracle/ is imported here and only here, as the thing timed BESIDE the product)
# ---------------------------------------------------------------------------------------------
def cpu_baseline(w, fiedler, budget_s, max_iters=20, min_s=3.0):
    """Frank-Wolfe iterations of the same workload from x0 on the host CPU, 1 thread, until `budget_s` is spent.
    fiedler = "tracemin": the reference-equivalent path (oracle/: TraceMIN + SuperLU exactly as networkx runs it
    for the reference, NumPy assembly / gradient / LP).  fiedler = "eigsh": the same loop with SciPy's ARPACK
    Lanczos -- NOT what the reference runs, the strong CPU baseline SURVEY section 8(d) asks for."""
    import oracle
    mo = oracle.MacOracle(w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"], w["n"], fiedler=fiedler)
    t0 = time.perf_counter()
    done, passes = 0, 0
    fs = []
    while True:
        x = w["x0"].copy()
        for it in range(max_iters):
            f, g = mo.problem(x)
            s = oracle.solve_subset_box_lp(g, w["k"])
            _ = f + g @ (s - x), np.linalg.norm(g)
            x = x + oracle.naive_stepsize(it) * (s - x)
            done += 1
            if passes == 0:
                fs.append(float(f))
            if time.perf_counter() - t0 > budget_s:
                break
        passes += 1
        if time.perf_counter() - t0 > min_s:
            break
    el = time.perf_counter() - t0
    what = f"first {done} Frank-Wolfe iterations" if passes == 1 else f"{passes} passes of the first {max_iters} Frank-Wolfe iterations"
    how = ("oracle/ (TraceMIN-Fiedler with SuperLU LU, tol 1e-8, RandomState(7) start = the reference's arithmetic)"
           if fiedler == "tracemin" else
           "oracle/ loop with scipy.sparse.linalg.eigsh (ARPACK Lanczos, which='SA', tol 1e-10) instead of TraceMIN -- not the reference's solver")
    return dict(value=done / el, unit="iter/s", cores=1, host_cores=os.cpu_count(), kind="port",
                sample=f"{what} of the same workload ({el:.1f} s), {how} on the host CPU, 1 thread", f_traj=fs)


def _same_node_child(n, q):
    """ONE reference-equivalent Fiedler solve (oracle/: TraceMIN + SuperLU, tol 1e-8, RandomState(7) start) of the configs[3] ER
    family at N = n, m ~ n^2 / 2000 candidates, 10 % support -- the size at which the reference's sparse LU still finishes."""
    import oracle
    w = er_workload(n, 2.0 * (n * n / 2000.0) / (n * (n - 1)), 0, f"ER N={n}")
    Lf = oracle.laplacian_from_edges(w["fi"], w["fj"], w["fw"], n)
    L = oracle.mac_laplacian(Lf, w["ci"].astype(np.int64), w["cj"].astype(np.int64), w["cw"], w["x0"], n)
    t0 = time.perf_counter()
    lam = oracle.find_fiedler_pair(L)[0]
    q.put(dict(n=n, m_candidates=int(len(w["cw"])), nnz=int(L.nnz), seconds_per_solve=time.perf_counter() - t0, lambda2=float(lam)))


def cpu_same_node_point(n=20000, hard_s=240.0):
    """the reference's sparse LU does not finish at configs[3]'s N = 100 000, so the same-node CPU figure is one
    Fiedler solve at N = 20 000 of the same family, timed in THIS run on THIS host (1 thread), next to the GPU's solve of
    the same matrix.  Returns None when it does not finish within hard_s."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_same_node_child, args=(n, q))
    t0 = time.perf_counter()
    p.start()
    res = None
    while time.perf_counter() - t0 < hard_s:
        try:
            res = q.get(timeout=1.0)
            break
        except Exception:      # queue.Empty
            if not p.is_alive():
                break
    if p.is_alive():
        p.terminate()
    p.join()
    return res


def gpu_same_node_point(n, device):
    """The GPU's eigen-solve of the matrix cpu_same_node_point() times (median of 5 cold solves)."""
    from mac_amd import _lib
    from mac_amd.utils.fiedler import reference_start_block
    w = er_workload(n, 2.0 * (n * n / 2000.0) / (n * (n - 1)), 0, f"ER N={n}")
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"], device=device)
    P.set_start(reference_start_block(n)[:, 0].copy())
    P.set_x(w["x0"])
    ts, lam = [], 0.0
    for _ in range(6):
        P.synchronize()
        t0 = time.perf_counter()
        lam, _, _ = P.fiedler(tol=1e-8, want_vec=False)
        P.synchronize()
        ts.append(time.perf_counter() - t0)
    P.close()
    return dict(seconds_per_solve=sorted(ts[1:])[2], lambda2=float(lam))


def _cpu_child(cfg, fiedler, budget_s, q):
    q.put(cpu_baseline(make_workload(cfg), fiedler, budget_s))


def cpu_baseline_bounded(cfg, fiedler, budget_s, hard_s):
    """cpu_baseline in a child process with a hard wall-clock limit: one SuperLU factorisation cannot be
    interrupted from Python, and at config 4 (N = 100k) it does not finish (fill-in, SURVEY 6.2: > 50 min)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_cpu_child, args=(cfg, fiedler, budget_s, q))
    t0 = time.perf_counter()
    p.start()
    res = None
    while p.is_alive() and time.perf_counter() - t0 < hard_s:
        try:
            res = q.get(timeout=1.0)
            break
        except Exception:      # queue.Empty
            pass
    if res is None and not p.is_alive():
        try:
            res = q.get(timeout=1.0)
        except Exception:
            res = None
    if p.is_alive():
        p.terminate()
    p.join()
    if res is None and p.exitcode not in (None, 0, -15):
        raise RuntimeError(f"cpu_baseline child exited with code {p.exitcode}")
    if res is None:
        el = time.perf_counter() - t0
        return dict(value=1.0 / el, unit="iter/s", cores=1, host_cores=os.cpu_count(), kind="port", f_traj=[],
                    upper_bound=True,
                    sample=f"the first Frank-Wolfe iteration of the same workload did not finish within {el:.0f} s "
                           "(workload load included; the reference's sparse LU needs > 50 min here, SURVEY 6.2); value is that "
                           f"UPPER BOUND; oracle/ ({fiedler}) on the host CPU, 1 thread")
    return res


# ---------------------------------------------------------------------------------------------
# PMC traffic of the dominant kernel: two short rocprofv3 passes of this same script
# ---------------------------------------------------------------------------------------------
def pmc_traffic(cfg, steps, precision=0, timeout_s=150):
    """HBM bytes per launch of the fused step kernel = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes), each counter from
    its own `rocprofv3 --pmc` pass (MI355X_MICROARCH.md: FETCH_SIZE on gfx950 tallies 128-byte requests at 64 B)
    of `bench.py --config cfg --steps K --warmup 0 --pmc-child`: the SAME K iterations from x0 as a timed pass, so the
    launches counted are the launches timed.  Returns (bytes or None, note)."""
    import csv
    import glob
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 not on PATH"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="machip_pmc_")
        env = dict(os.environ, TMPDIR=tempfile.gettempdir())
        cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", str(steps), "--warmup", "0",
               "--precision", str(precision), "--pmc-child"]
        try:
            subprocess.run(cmd, cwd=tempfile.gettempdir(), env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            tot, cnt = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r["Counter_Name"] != counter:
                            continue
                        if any(kn in r["Kernel_Name"] for kn in STEP_KERNELS):
                            tot += float(r["Counter_Value"])
                        if any(kn in r["Kernel_Name"] for kn in STEP_HEADS):
                            cnt += 1
            if cnt == 0:
                return None, f"rocprofv3 --pmc {counter} produced no rows for the step kernels"
            vals[counter] = (tot / cnt, cnt)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} pass timed out"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    by = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return by, (f"traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB per Lanczos step (all launches of a step) over {vals['FETCH_SIZE'][1]} steps, two separate "
                f"rocprofv3 --pmc passes of this script (the same {steps} iterations from x0 as a timed pass) run by this bench invocation")


# ---------------------------------------------------------------------------------------------
# roofline unit per solver mode: the object describes the launch group that
# actually ran (machip_solve_mode), with its own algorithmic bytes and its own in-solve duration
# ---------------------------------------------------------------------------------------------
MODE_INFO = {
    1: ("fused Lanczos step, gather form: ONE launch of k_pipe_vec (CSR SpMV with gathered 16-byte records + all vector work of the step)", "step"),
    2: ("fused Lanczos step, column-panel form: k_pan_mul8 + k_pan_finu (8-byte operand in LDS; mac_amd/csrc/panel_u.h; record form k_pan_mul + k_pan_fin with option panel_u = 0)", "step"),
    3: ("fused Lanczos step on the padded fixed-width copy of L(x): ONE launch of k_pipe_vec<.., ELLW>", "step"),
    4: ("one Lanczos step INSIDE the single-workgroup kernel k_lan_persist (matrix in registers / LDS; 64 steps per launch; mac_amd/csrc/persist.h)", "step"),
    5: ("classic two-kernel Lanczos step: k_spmv_* <OpLanczos> + k_lan_update", "step"),
    6: ("one LOBPCG iteration preconditioned by the odometry chain: k_tri_solve + product + k_lob_update (mac_amd/csrc/precond.h)", "iter"),
    7: ("one preconditioned iteration of the exact chain + closures mode: k_lob_fused (update + tridiagonal solve) + k_wb_h (s x s) + k_wb_w (n x s) + product; "
        "set-up (s column solves, capacitance matrix, its inverse on the f64 matrix cores) included in the duration (mac_amd/csrc/woodbury.h)", "iter"),
    8: ("fused Lanczos step with fp32 storage (8-byte records, fp32 values and basis), fp64 refinement sequences behind it", "step"),
}


def mode_bytes(mode, n, nnz, closures, precision):
    """Algorithmic bytes of ONE unit of the mode (SURVEY 8(d) B_spmv + the vector passes the unit makes)."""
    if mode in (1, 2, 3, 5):
        return step_bytes(n, nnz)
    if mode == 8:
        return 8.0 * nnz + 4.0 * (n + 1) + 32.0 * n
    if mode == 4:
        return 8.0 * n                                   # the basis column the step stores; L(x) and the operand stay on the CU
    spmv = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n
    if mode == 6:
        return spmv + 16 * 8.0 * n                       # 16 vector passes per iteration (machip_solve_stats.vec_passes)
    return spmv + 16 * 8.0 * n + 8.0 * n * closures + 8.0 * closures * closures     # + Z = T^-1 U (n x s) and C^-1 (s x s) per application


def roofline_by_mode(passes, n, precision):
    """Group every timed iteration by the solver mode that served it; the object describes the group with the most device time."""
    groups = {}
    for _, rec in passes:
        for r in rec:
            mode, clos = (r[7], r[8]) if len(r) > 8 else (1, 0)
            unit = MODE_INFO.get(mode, MODE_INFO[1])[1]
            units = r[6] if unit == "step" else r[1]
            ms = r[5] if unit == "step" else r[4]         # steps: events around the Krylov chunks; iterations: the whole solve
            if units <= 0 or ms <= 0:
                continue
            g = groups.setdefault(mode, dict(units=0, ms=0.0, bytes=0.0, iters=0, closures=0))
            g["units"] += units; g["ms"] += ms; g["iters"] += 1; g["closures"] = max(g["closures"], clos)
            g["bytes"] += units * mode_bytes(mode, n, r[2], clos, precision)
    if not groups:
        return None, []
    listing = [dict(mode=m_, unit=MODE_INFO.get(m_, MODE_INFO[1])[1], units=g["units"], frac_of_solve_time=g["ms"] / sum(x["ms"] for x in groups.values()),
                    us_per_unit=1e3 * g["ms"] / g["units"], fw_iterations=g["iters"]) for m_, g in sorted(groups.items())]
    top = max(groups, key=lambda m_: groups[m_]["ms"])
    return (top, groups[top]), listing


# ---------------------------------------------------------------------------------------------
def bench_c5_batched(args):
    """BASELINE.json configs[4] on ONE GPU: city10000 + sphere2500 as a batch of independent problems (replicas,
    no collective): one handle + stream + host thread per graph on the same GPU; ctypes releases the GIL, the
    tiny kernels of the two solves overlap on the chip.  value = total FW iterations of both / wall time."""
    import threading
    from mac_amd import _lib
    from mac_amd.utils.fiedler import reference_start_block
    ws = [make_workload("c5b"), make_workload("c5a")]
    Ps = []
    for w in ws:
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
        P.set_precision(args.precision)
        P.set_start(reference_start_block(w["n"])[:, 0].copy())
        run_pass(P, w["k"], args.warmup, w["x0"])
        P.set_x(w["x0"]); P.synchronize()
        Ps.append(P)
    fs = [None, None]

    def work(i):
        fs[i] = [r["f"] for r in run_pass(Ps[i], ws[i]["k"], args.steps, ws[i]["x0"])]
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for P in Ps:
        P.synchronize()
    el = time.perf_counter() - t0
    t1 = time.perf_counter()
    for i in range(2):
        run_pass(Ps[i], ws[i]["k"], args.steps, ws[i]["x0"])
        Ps[i].synchronize()
    seq = time.perf_counter() - t1
    print(json.dumps({"metric": "frank_wolfe_iters_per_sec", "value": 2 * args.steps / el, "unit": "iter/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / (2 * args.steps),
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f64" if args.precision == 0 else "f32 iterate + f64 Rayleigh/residual refinement",
                      "data": "dataset (tests/golden/data)",
                      "config": {"workload": "configs[4]: city10000.g2o + sphere2500.g2o batched (2 concurrent handles, 1 GPU), K=20%"
                                             + ("" if args.precision == 0 else " -- precision 1 (fp32 iterate + fp64 refinement) as BASELINE.json words it; "
                                                "measured SLOWER than the fp64 default on this hardware (DESIGN 4.2, docs/log_r1_r4.md 4.2d)"),
                                 "precision_note": "default --precision 0 (fp64) is the faster mode; --precision 1 is the fp32 + fp64-refinement arithmetic configs[4] names",
                                 "fw_iters_each": args.steps, "parallelism": "replicas: one stream + host thread per graph"},
                      "sequential_value": 2 * args.steps / seq, "lambda2_last": [fs[0][-1], fs[1][-1]]}))
    for P in Ps:
        P.close()


# ---------------------------------------------------------------------------------------------
def bench_c5_sweep(args):
    """The reference's real experiment on BASELINE.json configs[4]'s two pose graphs (examples/g2o_experiment.py:306-336: a
    sweep of budgets, MAC.solve(max_iters = 20) each) on ONE GPU: per graph one handle whose evaluation lanes run the
    budgets 10 % .. 90 % concurrently (machip_fw_sweep), the two graphs in two host threads.  A step = one
    Frank-Wolfe iteration of one budget; value = all iterations of all budgets of both graphs / wall time.  Stop tests
    disabled, cold eigen-solves, exactly as the other configs are timed."""
    import threading
    from mac_amd import _lib
    from mac_amd.utils.fiedler import reference_start_block
    ws = [make_workload("c5b"), make_workload("c5a")]
    pcts = (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)
    Ps, ks, X0 = [], [], []
    for w in ws:
        P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
        P.set_precision(args.precision)
        P.set_start(reference_start_block(w["n"])[:, 0].copy())
        m = len(w["cw"])
        kk = [int(p_ * m) for p_ in pcts]
        x0 = np.zeros((len(kk), m))
        for j, k in enumerate(kk):
            x0[j, np.argpartition(w["cw"], -k)[-k:]] = 1.0       # NaiveGreedy init per budget (mac/solvers/baseline.py:10-13)
        P.fw_sweep(kk, x0, max_iters=max(1, args.warmup), gap_tol=0.0, grad_tol=0.0, want_rounded=False)   # lanes, graphs, buffers
        Ps.append(P); ks.append(kk); X0.append(x0)
    res = [None, None]

    def work(i):
        res[i] = Ps[i].fw_sweep(ks[i], X0[i], max_iters=args.steps, gap_tol=0.0, grad_tol=0.0, want_rounded=False)
    passes = []
    total = 0.0
    while True:
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        passes.append(el); total += el
        if total >= args.min_seconds or len(passes) >= args.max_repeats:
            break
    el = sorted(passes)[(len(passes) - 1) // 2]
    its = sum(int(r["iters"].sum()) for r in res)
    t1 = time.perf_counter()
    for i in range(2):                    # the same budgets one after the other, one at a time (what a loop over MAC.solve does)
        for j, k in enumerate(ks[i]):
            run_pass(Ps[i], k, args.steps, X0[i][j])
    seq = time.perf_counter() - t1
    print(json.dumps({"metric": "frank_wolfe_iters_per_sec", "value": its / el, "unit": "iter/s", "n_gpus": 1,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / its, "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == 0 else "f32 iterate + f64 Rayleigh/residual refinement",
                      "data": "dataset (tests/golden/data)",
                      "config": {"workload": "configs[4] graphs as the reference's budget sweep: city10000.g2o + sphere2500.g2o, 9 budgets (10..90 percent of the loop "
                                             f"closures) x {args.steps} Frank-Wolfe iterations each, concurrently on one GPU (machip_fw_sweep)",
                                 "budgets_per_graph": len(pcts), "fw_iters_each": args.steps,
                                 "parallelism": "evaluation lanes: up to 12 budgets per graph at a time, one host thread per lane"},
                      "repeats": len(passes), "pass_ms": [round(1e3 * p, 3) for p in passes], "iterations_per_pass": its,
                      "sequential_value": its / seq,
                      "lambda2_last": [[float(r["f_traj"][j, -1]) for j in (0, len(pcts) - 1)] for r in res]}))
    for P in Ps:
        P.close()


# ---------------------------------------------------------------------------------------------
def bench_er_sweep(args, base):
    """The reference's sweep (examples/g2o_experiment.py:306-336: one MAC.solve per budget on the same graph) at the ER sizes of
    BASELINE.json configs[1] / configs[3] (--config c2s / c4s): B budgets K_b = K (0.5 + b/(B-1)) of the same graph through
    machip_fw_sweep -- every budget on an evaluation lane of its own (own x / gradient / CSR / panel form / Krylov basis / stream on
    its own hardware queue), the chip-filling step kernels of the lanes interleaving on the GPU.  A step = one Frank-Wolfe iteration of one
    budget; value = all iterations of all budgets / wall time of the concurrent sweep; `sequential_value` = the same budgets one at
    a time on one handle.  Stop tests disabled, cold eigen-solves, as every other config is timed."""
    from mac_amd import _lib
    from mac_amd.utils.fiedler import reference_start_block
    w = make_workload(base)
    n, m, k = w["n"], len(w["cw"]), w["k"]
    B = 4
    ks = [max(1, int(round(k * (0.5 + b / (B - 1))))) for b in range(B)]
    rng = np.random.default_rng(0)
    X0 = np.zeros((B, m))
    for b, kb in enumerate(ks):
        X0[b, rng.choice(m, kb, replace=False)] = 1.0
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(n)[:, 0].copy())
    scan = {}
    res = None
    for lanes in (2, 4):
        P.set_option("lanes", lanes)
        P.fw_sweep(ks, X0, max_iters=max(1, args.warmup), gap_tol=0.0, grad_tol=0.0, want_rounded=False)        # lanes, graphs, buffers
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = P.fw_sweep(ks, X0, max_iters=args.steps, gap_tol=0.0, grad_tol=0.0, want_rounded=False)
            ts.append(time.perf_counter() - t0)
        its = int(res["iters"].sum())
        scan[lanes] = dict(value=its / sorted(ts)[1], pass_ms=[round(1e3 * t, 3) for t in ts], lambda2_last=[float(res["f_traj"][b, -1]) for b in range(B)])
    for b in range(B):
        run_pass(P, ks[b], 1, X0[b])
    t1 = time.perf_counter()
    seq_f = []
    for b in range(B):
        rec = run_pass(P, ks[b], args.steps, X0[b])
        seq_f.append(rec[-1]["f"])
    P.synchronize()
    seq = B * args.steps / (time.perf_counter() - t1)
    best = max(scan, key=lambda l: scan[l]["value"])
    print(json.dumps({"metric": "frank_wolfe_iters_per_sec", "value": scan[best]["value"], "unit": "iter/s", "n_gpus": 1, "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": 1e3 / scan[best]["value"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "f64", "data": "synthetic",
                      "config": {"workload": w["name"] + f" as the reference's budget sweep: {B} budgets K_b = K (0.5 + b/{B - 1}) x {args.steps} Frank-Wolfe iterations, "
                                             "concurrently on one GPU (machip_fw_sweep)", "N": n, "m_candidates": m, "K": ks, "lanes": best,
                                 "parallelism": f"{best} evaluation lanes, one host thread and one hardware queue each"},
                      "lane_scan": {str(l): v for l, v in scan.items()}, "sequential_value": seq,
                      "aggregate_vs_one_at_a_time": scan[best]["value"] / seq,
                      "lambda2_last_equal_to_sequential": [scan[best]["lambda2_last"][b] == seq_f[b] for b in range(B)]}))
    P.close()


# ---------------------------------------------------------------------------------------------
def self_launch(args):
    """--gpus N without a launcher: spawn N rank processes (one GPU each) of this script."""
    from mac_amd import _lib
    _lib.load()
    have = _lib.device_count()
    if have < args.gpus and os.environ.get("MACHIP_SHARE_GPU") != "1":      # (MACHIP_SHARE_GPU=1: protocol test, several ranks on one GPU)
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible -- refusing to run a mislabelled {have}-GPU job")
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    key = uuid.uuid4().hex
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MACHIP_RDZV_KEY=key, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies (no GPU, RCCL refusing the communicator ...) must not leave its peers waiting in a collective
    import threading
    buf = []
    rd = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    stalled = [r for r, p in enumerate(procs) if p.poll() is None]
    codes = [p.wait() for p in procs]
    rd.join(timeout=5)
    if failed:
        first = [r for r, c in enumerate(codes) if c not in (0, -15)]
        sys.stderr.write(f"bench.py launcher: rank(s) {first} failed first (exit codes {codes}); rank(s) {stalled} were still running and were terminated\n")
    if not failed and not any(codes):
        sys.stdout.write((buf[0] if buf else b"").decode())
        sys.stdout.flush()
    else:
        raise SystemExit(f"rank exit codes {codes}")


def collect_nccl_log(prefix, cap=40):
    """Lines RCCL wrote at NCCL_DEBUG=WARN (one file per rank process), for the JSON line."""
    import glob
    lines = []
    for f in sorted(glob.glob(prefix + "_*.log")):
        try:
            with open(f, errors="replace") as fh:
                lines += [f"{os.path.basename(f)}: {l.rstrip()}" for l in fh if l.strip()]
            os.remove(f)
        except OSError:
            pass
    return lines[:cap]


def dry_run(args, dist, rank, local_rank, world, ndev, nccl_log):
    """`--gpus N --dry`: everything the first multi-GPU run touches for the first time, on a tiny problem, each step reported
    instead of raised: visible devices, peer-access matrix (hipDeviceCanAccessPeer), RCCL communicator + first all-gather
    (under libmachip's watchdog), IPC export / open of the peers' buffers and a row-partitioned eigen-solve, results equal on
    every rank."""
    from mac_amd import _lib
    from mac_amd.dist import detach_ipc
    from mac_amd.utils.fiedler import reference_start_block
    lib = _lib.load()
    dev = local_rank % max(1, ndev)
    rep = {"rank": rank, "device": dev, "visible_devices": ndev, "peer_access_row": [int(lib.machip_peer_access(dev, b)) for b in range(ndev)],
           "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    w = er_workload(6000, 0.002, 1, "dry-run ER N=6000")     # (beyond the single-workgroup kernel: the fused, row-partitionable step runs)
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"], device=dev)
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    steps = {}
    share = os.environ.get("MACHIP_SHARE_GPU") == "1" and ndev < world
    if dist is not None and world > 1:
        # first contact is collective and exception-safe (_attach_leg): a step that fails on ANY rank is dropped on every rank, so
        # nobody walks into a collective its peers never joined
        if not share:
            ok, msg = _attach_leg(P, dist, rank, world, rccl=True, ipc=False, shard_eig=False)
            steps["rccl_comm_init"] = "ok" if ok else f"FAILED: {msg}"
        else:
            steps["rccl_comm_init"] = "skipped (ranks share a GPU: RCCL refuses that)"
        ok, msg = _attach_leg(P, dist, rank, world, rccl=False, ipc=True, shard_eig=True)
        steps["ipc_exchange"] = "ok" if ok else f"FAILED: {msg}"
    P.set_x(w["x0"])
    fs = []
    try:
        for it in range(2):
            f, dual, gn = P.fw_step(w["k"], it)
            fs.append(f.hex()); P.fw_commit()
        steps["two_fw_iterations"] = "ok"
        steps["comm_mode"] = int(lib.machip_comm_mode(P._h))
    except Exception as e:             # noqa: BLE001
        steps["two_fw_iterations"] = f"FAILED: {e}"
    rep["steps"] = steps; rep["lambda2_hex"] = fs
    reps = dist.all_gather_object(rep) if dist is not None else [rep]
    if steps.get("ipc_exchange") == "ok" and all(r["steps"].get("two_fw_iterations") == "ok" for r in reps):
        detach_ipc(P, dist)
    if rank == 0:
        same = len({tuple(r["lambda2_hex"]) for r in reps}) == 1
        ok = same and all(v == "ok" or isinstance(v, int) or str(v).startswith("skipped") for r in reps for v in r["steps"].values())
        print(json.dumps({"dry": True, "n_gpus": world, "ok": bool(ok), "results_equal_on_all_ranks": same, "ranks": reps,
                          "errors": collect_nccl_log(nccl_log) if nccl_log else []}))
    P.close()
    if dist is not None:
        dist.close()


# ---------------------------------------------------------------------------------------------
# --gpus N > 1: ONE invocation yields all the multi-GPU evidence: the candidate-shard pass (headline), the
# replicas pass and the row-partitioned (IPC) eigen-solve pass, in that order, inside one process group; a leg whose first
# contact fails is reported under `errors` and the others still print.
# ---------------------------------------------------------------------------------------------
def _agree(dist, ok):
    return all(dist.all_gather_object(bool(ok)))


def _attach_leg(P, dist, rank, world, rccl, ipc, shard_eig):
    """First contact of one leg, collective and exception-safe: every rank runs the same exchanges whatever happens locally;
    when any rank fails, every rank drops what it had attached.  Returns (ok, message)."""
    from mac_amd.dist import attach, attach_ipc
    msgs = []
    if rccl:
        err = None
        try:
            attach(P, dist, rank, world)            # ncclCommInitRank inside libmachip under its watchdog
        except Exception as e:                      # noqa: BLE001
            err = f"rank {rank}: RCCL communicator: {e}"
        errs = [e for e in dist.all_gather_object(err) if e]
        if errs:
            P.comm_drop()
            return False, "; ".join(errs)
    if ipc:
        P.set_option("shard_eig", 1 if shard_eig else 0)
        err = None
        try:
            attach_ipc(P, dist, rank, world, timeout_s=float(os.environ.get("MACHIP_IPC_TIMEOUT", "20")))
        except Exception as e:                      # noqa: BLE001  (attach_ipc has already agreed and dropped on every rank)
            err = f"rank {rank}: IPC attach: {e}"
        errs = [e for e in dist.all_gather_object(err) if e]
        if errs:
            P.comm_drop()
            return False, "; ".join(errs)
    return True, ""


def _collective(dist, fn):
    """Run fn() on every rank; whatever happens locally, every rank then learns every rank's error (no rank is left behind in a
    collective or dies alone).  Returns the list of error strings (empty = all fine)."""
    err = None
    try:
        fn()
    except Exception as e:                          # noqa: BLE001
        err = f"rank {dist.rank}: {e}"
    return [e for e in dist.all_gather_object(err) if e]


def _timed_passes(P, k, steps, x0, dist, npass, comm_timing=False):
    """npass passes of `steps` Frank-Wolfe iterations from x0, barrier + synchronise on both sides, max over ranks.
    Returns (median seconds, records of the median pass); a failing rank makes every rank return an error string."""
    passes = []
    for _ in range(npass):
        err = None
        rec, el = [], 0.0
        try:
            P.set_x(x0); P.synchronize()
        except Exception as e:                      # noqa: BLE001
            err = str(e)
        dist.barrier()
        t0 = time.perf_counter()
        if err is None:
            try:
                r = P.fw_run(k, steps)               # exactly `steps` iterations on the C side, as at N = 1
                P.synchronize()
                el = time.perf_counter() - t0
                rec = [dict(f=float(r["f"][it]), steps=int(st.lanczos_steps), nnz=int(st.nnz), step_ms=float(st.step_ms), steps_timed=int(st.steps_timed),
                            gpu_ms=float(st.gpu_ms)) for it, st in enumerate(r["stats"])]
            except Exception as e:                  # noqa: BLE001
                err = f"rank {dist.rank}: {e}"
        errs = [e for e in dist.all_gather_object(err) if e]
        if errs:
            return None, "; ".join(errs)
        passes.append((dist.max(el), rec))
    passes.sort(key=lambda t: t[0])
    el, rec = passes[(len(passes) - 1) // 2]
    if comm_timing:
        # per-iteration device time of the gradient kernel and of the exchange behind it (hipEvents, machip_comm_timing): an untimed
        # pass of the same iterations driven one machip_fw_step at a time
        err = None
        try:
            P.set_x(x0)
            for it in range(steps):
                P.fw_step(k, it)
                rec[it]["grad_us"], rec[it]["exchange_us"] = P.comm_timing()
                P.fw_commit()
            P.synchronize()
        except Exception as e:                      # noqa: BLE001
            err = f"rank {dist.rank}: {e}"
        errs = [e for e in dist.all_gather_object(err) if e]
        if errs:
            return None, "; ".join(errs)
    return el, rec


def bench_multi(args, dist, rank, local_rank, world, ndev, nccl_log):
    from mac_amd import _lib
    from mac_amd.dist import detach_ipc
    from mac_amd.utils.fiedler import reference_start_block
    cfg = args.config
    w = make_workload(cfg)
    n, m, k = w["n"], len(w["cw"]), w["k"]
    dev = local_rank % max(1, ndev)
    share = os.environ.get("MACHIP_SHARE_GPU") == "1" and ndev < world       # protocol run: several ranks on one GPU (RCCL refuses that)
    start = reference_start_block(n)[:, 0].copy()
    errors, legs = {}, {}
    want = ("shard", "replicas", "ipc_eig") if args.mode == "all" else (args.mode,)

    def mk():
        P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"], device=dev)
        P.set_precision(args.precision)
        P.set_start(start)
        return P

    # ---- leg 1: candidate shard + one exchange of the gradient per iteration, eigen-solve replicated (the north star's split) ----
    head = None
    if "shard" in want:
        P = mk()
        ok, msg = _attach_leg(P, dist, rank, world, rccl=not share, ipc=share, shard_eig=False)
        if not ok:
            errors["shard"] = ("refused: two ranks on one device -- " if share else "") + msg
        else:
            werr = _collective(dist, lambda: run_pass(P, k, args.warmup, w["x0"]))      # (the communicator's first collective runs in here, under libmachip's watchdog)
            passes, total = [], 0.0
            while True:
                if werr:
                    errors["shard"] = "warm-up: " + "; ".join(werr); break
                r = _timed_passes(P, k, args.steps, w["x0"], dist, 1, comm_timing=True)
                if r[0] is None:
                    errors["shard"] = r[1]; break
                passes.append(r); total += r[0]
                if total >= args.min_seconds or len(passes) >= args.max_repeats:
                    break
            if "shard" not in errors:
                passes.sort(key=lambda t: t[0])
                el, rec = passes[(len(passes) - 1) // 2]
                head = dict(el=el, rec=rec, pass_ms=[round(1e3 * p_[0], 3) for p_ in passes], total=total,
                            mode=int(_lib.load().machip_comm_mode(P._h)))
                legs["shard"] = {"value": args.steps / el, "unit": "iter/s", "scaling": "strong", "ms_per_step": 1e3 * el / args.steps,
                                 "grad_us": float(np.mean([r_["grad_us"] for r_ in rec])), "exchange_us": float(np.mean([r_["exchange_us"] for r_ in rec])),
                                 "exchange": "IPC peer writes (ranks share a GPU)" if share else "ncclAllGather of the padded m-vector (RCCL over xGMI)",
                                 "eig_ms_per_iter": float(np.mean([r_["gpu_ms"] for r_ in rec])), "comm_mode": head["mode"],
                                 "lambda2_first_last": [rec[0]["f"], rec[-1]["f"]], "lanczos_steps_per_iter": float(np.mean([r_["steps"] for r_ in rec]))}
            if share and ok and "shard" not in errors:
                try:
                    detach_ipc(P, dist)
                except Exception:               # noqa: BLE001
                    pass
        if "shard" in errors:
            try:
                P.comm_drop()                   # (a failed leg: abort the communicator rather than wait for peers in its destructor)
            except Exception:                   # noqa: BLE001
                pass
        P.close()
        dist.barrier()
    # ---- leg 2: replicas -- the reference's budget sweep (examples/g2o_experiment.py:306-336), one budget per GPU, no collective ----
    if "replicas" in want:
        kr = max(1, int(round(k * (0.5 + rank / max(1, world - 1)))))
        P = mk()
        werr = _collective(dist, lambda: run_pass(P, kr, min(args.warmup, 2), w["x0"]))
        r = (None, "warm-up: " + "; ".join(werr)) if werr else _timed_passes(P, kr, args.steps, w["x0"], dist, 3)
        P.close()
        if r[0] is None:
            errors["replicas"] = r[1]
        else:
            el, rec = r
            rates = dist.all_gather_object(dict(rank=rank, K=kr, eig_ms_per_iter=float(np.mean([r_["gpu_ms"] for r_ in rec])),
                                                lanczos_steps_per_iter=float(np.mean([r_["steps"] for r_ in rec]))))
            legs["replicas"] = {"value": args.steps * world / el, "unit": "iter/s", "scaling": "weak", "ms_per_pass_max_over_ranks": 1e3 * el,
                                "per_rank": rates, "what": "every rank runs its own budget K_r = K (0.5 + r/(R-1)) of the same graph, no collective; "
                                "value = all ranks' iterations / max-over-ranks time"}
        dist.barrier()
    # ---- leg 3: eigen-solve row-partitioned between the processes (IPC-mapped buffers, device-side flags), gradient by RCCL ----
    if "ipc_eig" in want:
        P = mk()
        ok, msg = _attach_leg(P, dist, rank, world, rccl=not share, ipc=True, shard_eig=True)
        if not ok:
            errors["ipc_eig"] = msg
        else:
            werr = _collective(dist, lambda: run_pass(P, k, min(args.warmup, 2), w["x0"]))
            r = (None, "warm-up: " + "; ".join(werr)) if werr else _timed_passes(P, k, args.steps, w["x0"], dist, 3, comm_timing=True)
            if r[0] is None:
                errors["ipc_eig"] = r[1]
            else:
                el, rec = r
                sm, sc = sum(r_["step_ms"] for r_ in rec), sum(r_["steps_timed"] for r_ in rec)
                legs["ipc_eig"] = {"value": args.steps / el, "unit": "iter/s", "scaling": "strong", "ms_per_step": 1e3 * el / args.steps,
                                   "us_per_lanczos_step": 1e3 * sm / max(1, sc), "steps_timed": sc, "comm_mode": int(_lib.load().machip_comm_mode(P._h)),
                                   "exchange_us": float(np.mean([r_["exchange_us"] for r_ in rec])), "lambda2_first_last": [rec[0]["f"], rec[-1]["f"]],
                                   # (identical bits where the shard leg ran the same gather step; at sizes where a single rank takes the
                                   # column-panel form -- which the partitioned solve does not shard -- the two agree to the last digits)
                                   "lambda2_max_rel_diff_vs_shard_leg": (None if head is None else float(max(abs(a_["f"] - b_["f"]) / abs(b_["f"]) for a_, b_ in zip(rec, head["rec"])))),
                                   "bit_identical_to_shard_leg": (head is not None and [r_["f"] for r_ in rec] == [r_["f"] for r_ in head["rec"]]),
                                   "what": "every rank launches its share of the workgroups of every fused Lanczos step on its own copy of L(x) and writes "
                                           "records / partial sums into every rank's copy (peer-mapped); steps ordered by flag words in device memory"}
            if "ipc_eig" not in errors:
                try:
                    detach_ipc(P, dist)
                except Exception:               # noqa: BLE001
                    pass
        if "ipc_eig" in errors:
            try:
                P.comm_drop()
            except Exception:                   # noqa: BLE001
                pass
        P.close()
        dist.barrier()
    if rank == 0:
        # headline: the shard pass; if its first contact failed, the replicas aggregate (named so)
        if "shard" in legs:
            value, scaling, el = legs["shard"]["value"], "strong", head["el"]
            par = (f"candidate shard x{world} + " + ("IPC peer writes of the gradient (ranks share a GPU)" if share else "ncclAllGather of the gradient (RCCL)") +
                   ", eigen-solve replicated on every rank")
        elif "replicas" in legs:
            value, scaling, el = legs["replicas"]["value"], "weak", 1e-3 * legs["replicas"]["ms_per_pass_max_over_ranks"]
            par = f"replicas x{world} (the shard leg failed first contact: see errors)"
        elif "ipc_eig" in legs:
            value, scaling, el = legs["ipc_eig"]["value"], "strong", args.steps / legs["ipc_eig"]["value"]
            par = f"row-partitioned eigen-solve x{world}"
        else:
            value, scaling, el, par = 0.0, "strong", 0.0, "every leg failed: see errors"
        out = {"metric": "frank_wolfe_iters_per_sec", "value": value, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * el / args.steps if el else None, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
               "dtype": "f64" if args.precision == 0 else "f32 iterate + f64 Rayleigh/residual refinement",
               "data": "synthetic" if cfg in ("c2", "c4") else "dataset (tests/golden/data)",
               "config": {"workload": w["name"], "N": n, "m_candidates": m, "K": k, "fixed_edges": int(len(w["fw"])), "fw_iters": args.steps,
                          "fiedler_tol": 1e-8, "parallelism": par},
               "legs_run": list(want)}
        if head is not None:
            out.update(repeats=len(head["pass_ms"]), pass_ms=head["pass_ms"], timed_wall_s=head["total"])
            # the roofline unit of the headline leg: the replicated Lanczos step, as at N = 1
            sm, sc = sum(r_["step_ms"] for r_ in head["rec"]), sum(r_["steps_timed"] for r_ in head["rec"])
            if sc > 0 and sm > 0 and not args.no_roofline:
                by = sum(r_["steps_timed"] * step_bytes(n, r_["nnz"]) for r_ in head["rec"]) / sc
                us = 1e3 * sm / sc
                out["roofline"] = {"bound": "hbm", "kernel": KERNEL, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": us, "algorithmic_bytes_per_launch": by,
                                   "launches_timed": sc, "note": "rank 0's replicated eigen-solve of the shard leg (median pass); same unit as the N = 1 line"}
        for nm in ("shard", "replicas", "ipc_eig"):
            if nm in legs:
                out[nm] = legs[nm]
        if "shard" in legs and "replicas" in legs:
            out["replicas"]["vs_shard_leg"] = legs["replicas"]["value"] / legs["shard"]["value"]
        out["predicted_vs_1gpu"] = {"shard": {2: 0.99, 4: 0.98, 8: 0.97}.get(world, 1.0 - 0.004 * world), "replicas": 0.8 * world,
                                    "why": "DESIGN section 7: the shard divides only the supergradient (0.5 % of an iteration) and pays one all-gather of 16 MB; "
                                           "replicas are independent problems (the 1.5 K budget has ~1.4x the entries of the 0.5 K one)"}
        errs = dict(errors)
        if share:
            errs["rccl"] = ("refused: two ranks on one device (MACHIP_SHARE_GPU=1 protocol run) -- RCCL was not initialised; the shard leg exchanged its "
                            "gradient through the IPC-mapped buffers instead")
        if nccl_log:
            lines = collect_nccl_log(nccl_log)
            if lines:
                errs["nccl_log"] = lines
        out["errors"] = errs
        print(json.dumps(out))
    dist.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4")
    ap.add_argument("--mode", default="all", choices=["all", "shard", "replicas", "ipc_eig"],
                    help="N > 1: all = candidate shard (headline, strong) + replicas (weak) + row-partitioned eigen-solve, one JSON line; or one leg only")
    ap.add_argument("--precision", type=int, default=0, help="0 = f64 throughout; 1 = f32 Krylov iterate + f64 Rayleigh/residual refinement (configs[4])")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the K-iteration pass until this much wall time is measured")
    ap.add_argument("--max-repeats", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline legs")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic = null)")
    ap.add_argument("--no-warm", action="store_true", help="skip the warm-start passes (use_cache=True: the reference's own micro-benchmark)")
    ap.add_argument("--no-same-node", action="store_true", help="configs[3]: skip the same-node reference-equivalent solve at N = 20 000 (~90 s of CPU)")
    ap.add_argument("--dry", action="store_true", help="N > 1: first-contact check only -- device visibility, peer access matrix, RCCL communicator + "
                    "one all-gather and the IPC buffer exchange on a tiny problem, no workload; prints one JSON line")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    nccl_log = None
    if world > 1:                      # RCCL's own warnings end up in the JSON line (`errors`), not on a terminal nobody reads
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        key = os.environ.get("MACHIP_RDZV_KEY") or os.environ.get("MASTER_PORT", "0")
        nccl_log = os.path.join(tempfile.gettempdir(), f"machip_nccl_{key}")
        os.environ.setdefault("NCCL_DEBUG_FILE", nccl_log + "_%h_%p.log")
    from mac_amd import _lib           # loads libmachip.so (HIP 7.2 runtime) before anything else
    _lib.load()
    _lib.require_device()
    ndev = _lib.device_count()
    if world > 1 and ndev < world and os.environ.get("MACHIP_SHARE_GPU") != "1":
        raise SystemExit(f"--gpus {world} but only {ndev} GPU(s) visible")
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):
        from mac_amd.dist import FileGroup     # rendezvous / barrier / max only; data path is RCCL
        dist = FileGroup(rank, world)

    def barrier():
        if dist is not None:
            dist.barrier()

    if args.dry:
        return dry_run(args, dist, rank, local_rank, world, ndev, nccl_log)
    if world > 1 and args.config not in ("c5", "c5s"):
        return bench_multi(args, dist, rank, local_rank, world, ndev, nccl_log)
    if args.config == "c5" and world == 1:
        return bench_c5_batched(args)
    if args.config == "c5s" and world == 1:
        return bench_c5_sweep(args)
    if args.config in ("c4s", "c2s") and world == 1:
        return bench_er_sweep(args, args.config[:2])
    replicas = world > 1 and (args.mode == "replicas" or args.config == "c5")
    cfg = args.config
    if cfg == "c5":                      # one pose graph per rank (SURVEY 8(e) last row)
        cfg = "c5b" if rank % 2 == 0 else "c5a"
    w = make_workload(cfg)
    n, m, k = w["n"], len(w["cw"]), w["k"]
    if replicas and args.config != "c5":  # budget sweep: rank r solves the same graph with budget K_r
        k = max(1, int(round(k * (0.5 + rank / max(1, world - 1)))))
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"], device=local_rank % max(1, ndev))
    P.set_precision(args.precision)
    from mac_amd.utils.fiedler import reference_start_block
    P.set_start(reference_start_block(n)[:, 0].copy())
    eig_mode = "single rank"
    if dist is not None and not replicas:
        from mac_amd.dist import attach, attach_ipc
        share = os.environ.get("MACHIP_SHARE_GPU") == "1" and ndev < world       # protocol test: several ranks on one GPU
        if not share:
            attach(P, dist, rank, world)  # ncclCommInitRank inside libmachip; the file group only carries the id
        eig_mode = "replicated on every rank"
        # Default: on when the ranks share a GPU (protocol run), OFF on distinct devices -- measured on one GPU the device-ordered
        # step costs +7.4 .. +8.8 us (two more launches per step and the flag round trip; profiles/r4_ipc_one_gpu.txt), which the
        # 1/R share of the gathers does not buy back at configs[3] (predicted 0.7x of one GPU at R = 8, below); MACHIP_IPC_EIG=1
        # turns it on anywhere.
        ipc_default = "1" if share else "0"
        if world > 1 and os.environ.get("MACHIP_IPC_EIG", ipc_default) != "0":
            # row-partitioned eigen-solve between the processes (machip_comm_init_ipc): first contact is guarded -- any failure
            # to map the peers' buffers leaves the replicated solve in place and is reported in the line
            try:
                attach_ipc(P, dist, rank, world, timeout_s=float(os.environ.get("MACHIP_IPC_TIMEOUT", "20")))
                eig_mode = "row-partitioned between the processes, steps ordered by device-side flags (IPC-mapped buffers)"
            except Exception as e:        # noqa: BLE001
                oks = False
                eig_mode = f"replicated on every rank (IPC attach failed on rank {rank}: {e})"
            # all ranks must agree: if anyone failed, nobody may run the partitioned step
            flags = dist.all_gather_object(eig_mode.startswith("row-partitioned"))
            if not all(flags) and any(flags):
                raise SystemExit("bench.py: IPC attach succeeded on some ranks only -- cannot continue")
            if share and not all(flags):
                raise SystemExit("bench.py: MACHIP_SHARE_GPU=1 needs the IPC communicator (RCCL refuses two ranks on one device): " + eig_mode)

    # ---- warmup (untimed) ----
    run_pass(P, k, args.warmup, w["x0"])
    if args.pmc_child:                    # profiled child of pmc_traffic(): a few iterations, nothing printed
        run_pass(P, k, args.steps, w["x0"])
        P.close()
        return
    # ---- timed region: passes of exactly K Frank-Wolfe iterations from x0, median pass reported ----
    passes, total = [], 0.0
    while True:
        P.set_x(w["x0"])
        P.synchronize()
        barrier()
        t0 = time.perf_counter()
        r = P.fw_run(k, args.steps)               # exactly K iterations, stop tests disabled, on the C side (machip_fw_run)
        P.synchronize()
        barrier()
        el = time.perf_counter() - t0
        rec = [(float(r["f"][it]), int(st.lanczos_steps), int(st.nnz), int(st.support), float(st.gpu_ms), float(st.step_ms), int(st.steps_timed)) + r["modes"][it]
               for it, st in enumerate(r["stats"])]
        if dist is not None:
            el = dist.max(el)             # every rank sees the same number, so every rank stops after the same pass
        passes.append((el, rec))
        total += el
        if total >= args.min_seconds or len(passes) >= args.max_repeats:
            break
    order = sorted(range(len(passes)), key=lambda i: passes[i][0])
    el, rec = passes[order[(len(order) - 1) // 2]]       # the median pass (lower median)
    # ---- the reference's own micro-benchmark is cache on / off (tests/benchmarks/test_cache_performance.py:10-48): the same
    #      pass with use_cache=True semantics (every eigen-solve but the first starts from the previous Fiedler vector) ----
    warm = None
    if world == 1 and not args.no_warm and not args.pmc_child:
        wp = []
        for _ in range(3):
            P.set_x(w["x0"])
            P.synchronize()
            t0 = time.perf_counter()
            r = P.fw_run(k, args.steps, warm_start=True)
            wrec = [(float(r["f"][it]), int(st.lanczos_steps)) for it, st in enumerate(r["stats"])]
            P.synchronize()
            wp.append((time.perf_counter() - t0, wrec))
        wel, wrec = sorted(wp, key=lambda t: t[0])[1]
        warm = {"value": args.steps / wel, "unit": "iter/s", "ms_per_step": 1e3 * wel / args.steps,
                "lanczos_steps_per_iter": float(np.mean([r[1] for r in wrec])), "lambda2_last": wrec[-1][0],
                "what": "the same pass with use_cache=True (MAC.Cache made real: every eigen-solve after the first starts from the previous "
                        "Fiedler vector); the headline value is the cold pass, the reference's effective behaviour (its cache write-back is a no-op, mac.py:126-127)"}
    # ---- cold start: the same pass with the landscape weighting of the start vector switched off (option start_land = 0: the start
    #      column as the reference draws it), and -- configs[1] / configs[3] -- both settings on the REFERENCE'S OWN 20 iterates
    #      (tests/golden): free-running trajectories part at the first near-tie of the top-K selection (configs[1]: from iterate 5 on,
    #      the reference's own run included), so only the teacher-forced pair compares the same matrices ----
    cold = None
    if world == 1 and not args.no_warm and not args.pmc_child:
        def _passes(nrep=3):
            cp = []
            for _ in range(nrep):
                P.set_x(w["x0"]); P.synchronize()
                t0 = time.perf_counter()
                r = P.fw_run(k, args.steps)
                P.synchronize()
                cp.append((time.perf_counter() - t0, [int(st.lanczos_steps) for st in r["stats"]], float(r["f"][args.steps - 1])))
            return sorted(cp, key=lambda t: t[0])[(nrep - 1) // 2]

        def _teacher():
            fx = {"c2": ("er10k_vertices.npz", "f_traj"), "c4": ("er100k_arpack.npz", "lam_traj")}.get(cfg)
            path = os.path.join(ROOT, "tests", "golden", fx[0]) if fx else None
            if not path or not os.path.exists(path):
                return None
            gv = np.load(path)
            x = w["x0"].copy(); st_sum, ms_sum, worst = 0, 0.0, 0.0
            for i in range(20):
                P.set_x(x)
                lam, _, _ = P.fiedler(tol=1e-8, want_vec=False)
                st_sum += int(P.stats.lanczos_steps); ms_sum += float(P.stats.gpu_ms)
                worst = max(worst, abs(lam - float(gv[fx[1]][i])) / abs(float(gv[fx[1]][i])))
                x = x + 2.0 / (i + 2) * (np.unpackbits(gv["ref_s_bits"][i])[:m].astype(np.float64) - x)
            return {"lanczos_steps": st_sum, "eig_ms": ms_sum, "worst_rel_lambda2_error": worst}
        try:
            t_on = _teacher()
            P.set_option("start_land", 0)
            cel, csteps, clam = _passes()
            t_off = _teacher()
            P.set_option("start_land", None)
            cold = {"unweighted": {"value": args.steps / cel, "unit": "iter/s", "lanczos_steps_per_iter": float(np.mean(csteps)), "lambda2_last": clam},
                    "what": "option start_land = 0: every cold eigen-solve starts from the reference's start column as drawn (fiedler.py:27-32); the headline "
                            "multiplies it by (landscape / max)^128 (DESIGN 4.2; include/machip.h machip_landscape)"}
            if t_on and t_off:
                cold["teacher_forced"] = {"weighted": t_on, "unweighted": t_off,
                                          "what": "lambda_2 on the reference's own 20 iterates (tests/golden), the same matrices for both settings: Lanczos steps and "
                                                  "device time of the 20 eigen-solves"}
        except Exception as e:        # noqa: BLE001
            cold = {"error": str(e)}
            P.set_option("start_land", None)
    units = args.steps
    if replicas:
        units = args.steps * world        # every rank ran K iterations of its own problem

    out = None
    if rank == 0:
        steps = np.array([r[1] for r in rec], dtype=float)
        if world == 1:
            par = "single GPU" + (" (RCCL communicator of 1 rank)" if dist is not None else "")
        elif replicas:
            par = (f"replicas x{world}: one independent problem per GPU (" +
                   ("city10000 / sphere2500 alternating" if args.config == "c5" else "budget sweep K_r = K (0.5 + r/(R-1))") + "), no collective")
        else:
            par = f"candidate shard x{world} + all-gather of the gradient ({'RCCL' if os.environ.get('MACHIP_SHARE_GPU') != '1' or ndev >= world else 'IPC peer writes: ranks share a GPU'}), eigen-solve {eig_mode}"
        out = {
            "metric": "frank_wolfe_iters_per_sec",
            "value": units / el,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if replicas else "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == 0 else "f32 iterate + f64 Rayleigh/residual refinement",
            "data": "synthetic" if cfg in ("c2", "c4") else "dataset (tests/golden/data)",
            "config": {"workload": w["name"] if args.config != "c5" else "configs[4]: city10000.g2o / sphere2500.g2o, one graph per GPU, K=20%",
                       "N": n, "m_candidates": m, "K": k, "fixed_edges": int(len(w["fw"])),
                       "fw_iters": args.steps, "fiedler_tol": 1e-8, "parallelism": par},
            "repeats": len(passes),
            "pass_ms": [round(1e3 * p[0], 3) for p in passes],
            "timed_wall_s": total,
            "lanczos_steps_per_iter": float(steps.mean()),
            "lambda2_first_last": [rec[0][0], rec[-1][0]],
            "nnz_first_last": [rec[0][2], rec[-1][2]],
            "eig_ms_per_iter": float(np.mean([r[4] for r in rec])),
        }
        if warm is not None:
            warm["steps_vs_cold"] = warm["lanczos_steps_per_iter"] / max(1.0, float(steps.mean()))
            warm["value_vs_cold"] = warm["value"] / (units / el)
            out["warm_start"] = warm
        if cold is not None:
            if "unweighted" in cold:
                cold["unweighted"]["value_vs_headline"] = cold["unweighted"]["value"] / (units / el)
            out["cold_start"] = cold
        if world > 1:
            # what this mode can be expected to deliver, printed next to what it did
            if replicas:
                pred = 1.0 * world if args.config == "c5" else 0.8 * world
                why = ("independent problems, no collective: R x the single-GPU rate x (mean rate / slowest rank's rate); the 1.5 K budget of the "
                       "sweep has ~1.4x the nnz of the 0.5 K one" if args.config != "c5" else "independent pose graphs, one per GPU")
            else:
                pred = {2: 0.99, 4: 0.98, 8: 0.97}.get(world, 1.0 - 0.004 * world)
                why = ("candidate shard: only the supergradient (0.4 % of an iteration) is divided, the all-gather of the 16 MB gradient costs more "
                       "than it saves; the eigen-solve is replicated on every rank (MACHIP_IPC_EIG=1 row-partitions it between the processes)")
                if eig_mode.startswith("row-partitioned"):
                    # DESIGN section 7, from measured components: configs[3] runs the column-panel step row-partitioned (round 6): tile-table
                    # round trip 1.2 + tile stream 5.0 / R + tail 1.5 + boundary 1.8, row kernel 5.5 + boundary 1.8, publish / wait launch 2.5,
                    # peer writes over xGMI + flag round trip >= 3 (unmeasured); one GPU alone: 12.5 us at the end of round 6 (the late work on the step took ~1 us
                    # out of both kernels' fronts: 1.2 -> 0.7 and 5.5 -> 4.5 below).  configs[1]: gather step 5 us fixed
                    # + 4.1 us per million entries / R + 8 us of exchange measured on one GPU + 3.
                    nnz_m = float(np.mean([r[2] for r in rec])) / 1e6
                    t1 = 12.5 if cfg == "c4" else 6.4
                    tR = (0.7 + 5.0 / world + 1.5 + 1.8 + 4.5 + 1.8 + 2.5 + 3.0) if cfg == "c4" else (5.0 + 4.1 * nnz_m / world + 8.0 + 3.0)
                    pred = t1 / tR
                    why = (f"row-partitioned eigen-solve: per step {tR:.1f} us predicted from measured components (DESIGN section 7: the step's fixed part -- two launch "
                           f"floors, one cold round trip each, the exchange -- does not shrink with the rank count) against {t1} us on one GPU: a single "
                           "problem of this size does not get faster on more GPUs; replicas mode is what scales")
            out["predicted_vs_1gpu"] = {"factor": pred, "why": why}
    # ---- roofline of the dominant kernel: in-solve duration from the hipEvents that bracket the Krylov chunks
    #      on the handle's stream (machip_solve_stats.step_ms / steps_timed), summed over EVERY timed pass ----
    if not args.no_roofline and rank == 0:
        top, mode_list = roofline_by_mode(passes, n, args.precision)
        if top is not None and top[0] in (1, 2, 3):
            # the fused step in whichever forms ran (gather on sparse iterates, panel on dense ones): ONE unit, as in rounds 1-4
            sel = (1, 2, 3)
            sm = sum(r[5] for p in passes for r in p[1] if len(r) <= 8 or r[7] in sel)
            sc = sum(r[6] for p in passes for r in p[1] if len(r) <= 8 or r[7] in sel)
            by_num = sum(r[6] * step_bytes(n, r[2]) for p in passes for r in p[1] if len(r) <= 8 or r[7] in sel)
            kernel_name = KERNEL
        elif top is not None:
            sm, sc, by_num = top[1]["ms"], top[1]["units"], top[1]["bytes"]
            kernel_name = MODE_INFO[top[0]][0] + (f" (up to {top[1]['closures']} closures)" if top[0] == 7 else "")
        else:
            sm = sc = by_num = 0
        if sc > 0 and sm > 0:
            us = 1e3 * sm / sc
            by = by_num / sc
            ach = by / (us * 1e-6) / 1e9
            traffic, tnote = None, "PMC passes skipped"
            if world == 1 and not args.no_pmc and top[0] in (1, 2, 3, 8):
                traffic, tnote = pmc_traffic(args.config, args.steps, args.precision)
            elif top[0] not in (1, 2, 3, 8):
                tnote = "traffic: null -- the PMC passes count the fused step kernels only; this mode's unit spans several kernels of a chain (or a slice of one persistent launch)"
            if top[0] in (1, 2, 3):
                note = ("unit = one Lanczos step (one launch of k_pipe_vec, or k_pan_mul8 + k_pan_finu where the column-panel form runs); "
                        "avg_launch_us = step_ms / steps_timed of machip_solve_stats: hipEvents on the handle's stream around the Krylov "
                        "chunks of every solve in the timed passes (step kernels + one 1-wave tail kernel per chunk, so slightly above "
                        "the pure kernel sums rocprofv3 reports); algorithmic bytes = 12 nnz + 4 (n+1) + 56 n per step whichever "
                        "kernels run it, step-weighted over the iterations; the CSR and the operand are Infinity-Cache resident "
                        "(<= 60 MB), peak is the 8 TB/s HBM figure all the same; " + tnote)
            elif MODE_INFO[top[0]][1] == "step":
                note = ("unit = one Lanczos step of the mode named in `kernel` (machip_solve_mode says which launch group served each timed solve: "
                        "`solver_modes`); avg_launch_us = in-solve duration per step (hipEvents around the Krylov chunks); algorithmic bytes per step as "
                        "that mode moves them (single-workgroup kernel: the 8n-byte basis column -- matrix and operand never leave the CU, so an "
                        "HBM fraction near zero is the design, not a defect: the step is a latency chain); " + tnote)
            else:
                note = ("unit = one preconditioned iteration of the mode named in `kernel` (`solver_modes` lists every mode that served a timed solve); "
                        "avg_launch_us = device time of the whole solve / its iterations (hipEvents ev0..ev1 of machip_solve_stats.gpu_ms: set-up -- "
                        "closure list, column solves, capacitance inverse -- and the explicit checks included); algorithmic bytes per iteration = "
                        "12 nnz + 4 (n+1) + 16 n (product) + 16 vector passes of 8n + (exact mode) 8 n s + 8 s^2 for the low-rank correction; " + tnote)
            peak_meas = None
            try:                          # the achievable peak measured on this box, reported next to the nominal one
                rd, tr = _lib.membench(1 << 30, 8, local_rank % max(1, ndev))
                peak_meas = {"read_GBps": rd, "triad_GBps": tr, "frac_of_read": ach / rd,
                             "how": "machip_membench: read-only sum / STREAM triad over 1 GiB arrays (4x the Infinity Cache), 8 launches each"}
            except Exception as e:        # noqa: BLE001
                peak_meas = {"error": str(e)}
            # secondary bound of the gather step (profiles/r2_c4_counters.md): one L1->L2 request per matrix
            # entry + one per 128 contiguous bytes; the L2 takes 128 channels x ~2.1 GHz requests per second
            nnz_mean = sum(r[6] * r[2] for p in passes for r in p[1]) / sc
            req = nnz_mean * 1.1 + (12.0 * nnz_mean + 60.0 * n) / 128.0
            l2_frac = (req / (us * 1e-6)) / (128 * 2.1e9)
            out["roofline"] = {"bound": "hbm", "kernel": kernel_name, "solver_modes": mode_list, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_us": us,
                               "algorithmic_bytes_per_launch": by, "launches_timed": sc, "peak_measured": peak_meas,
                               "l2_request_frac": l2_frac,
                               "l2_request_note": "requests per step if every entry were gathered from L2 (1.1 nnz + streamed bytes / 128) "
                                                  "/ step time / (128 channels x 2.1 GHz): the bound the one-kernel step runs into; the "
                                                  "panel form serves the gathers from LDS and issues ~6x fewer",
                               "note": note}
    if rank == 0 and world == 1 and not args.no_cpu:
        small = n <= 20000 and cfg != "c2"
        cb = cpu_baseline_bounded(cfg, "tracemin", budget_s=12.0, hard_s=20.0 if small else 45.0)
        ft = cb.pop("f_traj")
        if cfg == "c4":
            # measured elsewhere, labelled so: the reference itself (networkx TraceMIN + SuperLU), seconds per Fiedler solve of the
            # same ER family in the 8-vCPU BUILD CONTAINER -- not this node
            cb["reference_extrapolation_build_container"] = REF_EXTRAPOLATION
            if not args.no_same_node:
                pt = cpu_same_node_point(20000)
                if pt is not None:
                    gp = gpu_same_node_point(20000, local_rank % max(1, ndev))
                    pt["gpu_seconds_per_solve"] = gp["seconds_per_solve"]
                    pt["gpu_speedup_measured"] = pt["seconds_per_solve"] / gp["seconds_per_solve"]
                    pt["lambda2_rel_diff"] = abs(gp["lambda2"] - pt["lambda2"]) / abs(pt["lambda2"])
                    pt["what"] = ("ONE reference-equivalent Fiedler solve (oracle/: TraceMIN + SuperLU, tol 1e-8, 1 thread) of the configs[3] ER family at "
                                  "N = 20 000 (the size at which the sparse LU still finishes), timed in THIS run on THIS host, next to the GPU's solve "
                                  "of the same matrix")
                    cb["same_node_points"] = [pt]
                else:
                    cb["same_node_points"] = [{"n": 20000, "seconds_per_solve": None, "what": "did not finish within 240 s on this host"}]
        out["cpu_baseline"] = cb
        if ft:
            out["cpu_parity_lambda2_rel"] = float(max(abs(a - r[0]) / abs(a) for a, r in zip(ft, rec)))
        if cb.get("upper_bound"):
            # the CPU figure is an upper bound (first iteration did not finish): the ratio is a LOWER bound, named so
            out["speedup_vs_cpu_lower_bound"] = out["value"] / cb["value"]
        else:
            out["speedup_vs_cpu"] = out["value"] / cb["value"]
        cs = cpu_baseline_bounded(cfg, "eigsh", budget_s=15.0, hard_s=60.0)
        fts = cs.pop("f_traj")
        out["cpu_baseline_strong"] = cs
        if fts:
            out["cpu_strong_parity_lambda2_rel"] = float(max(abs(a - r[0]) / abs(a) for a, r in zip(fts, rec)))
        out["speedup_vs_cpu_strong"] = out["value"] / cs["value"]
    if rank == 0:
        if nccl_log:
            out["errors"] = collect_nccl_log(nccl_log)
        print(json.dumps(out))
    if dist is not None and eig_mode.startswith("row-partitioned"):
        from mac_amd.dist import detach_ipc
        detach_ipc(P, dist)
    P.close()
    if dist is not None:
        dist.close()


if __name__ == "__main__":
    main()
