"""GPU parity tests: every call goes through the C ABI of libmachip.so (ctypes) and is compared
with the CPU oracle on the same inputs, with the committed golden vectors of the reference, and
-- at BASELINE.json's full sizes -- through size-independent identities.

Tolerances (BASELINE.json north_star: fp64, lambda_2 within 1e-8 relative):
  lambda_2: 1e-8 relative;  Fiedler vector: max |v - v_ref| <= 2e-6 after sign alignment (both
  sides stop at residual 1e-8);  supergradient: 1e-5 relative to max|g| across solvers and
  BIT-EXACT given the same vector;  assembly: structure exact, values exact off-diagonal,
  diagonal to 1e-14 relative;  top-k / x update: bit-exact.
"""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp
import subprocess
import sys

import oracle
from conftest import ROOT, load_golden, sign_align
from mac_amd import _lib
from mac_amd.solvers import MAC, NaiveGreedy
from mac_amd.utils.fiedler import find_fiedler_pair, reference_start_block
from mac_amd.utils.graphs import Edge, weight_graph_lap_from_edge_list

pytestmark = pytest.mark.gpu

LAM_RTOL = 1e-8


def edges_of(g, pre):
    return [Edge(int(a), int(b), float(w)) for a, b, w in zip(g[pre + "i"], g[pre + "j"], g[pre + "w"])]


def problem_of(g):
    return _lib.Problem(int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])


def oracle_of(g):
    return oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], int(g["n"]))


# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nm", ["petersen_x0", "er300_x0", "er300_xfrac", "er2000_x0", "er2000_xfrac"])
def test_assembly_matches_reference(nm):
    g = load_golden(nm)
    P = problem_of(g)
    P.set_x(g["x"])
    indptr, indices, data = P.laplacian_csr()
    n = int(g["n"])
    # diagonal first, then sorted columns
    for r in [0, n // 2, n - 1]:
        assert indices[indptr[r]] == r
        assert np.all(np.diff(indices[indptr[r] + 1:indptr[r + 1]]) > 0)
    L = sp.csr_matrix((data, indices, indptr), shape=(n, n))
    L.sort_indices()
    assert np.array_equal(L.indptr, g["L_indptr"]) and np.array_equal(L.indices, g["L_indices"])
    off = L.indices != np.repeat(np.arange(n), np.diff(L.indptr))
    assert np.array_equal(L.data[off], g["L_data"][off])             # x_k*w_k: bit-exact
    assert np.allclose(L.data[~off], g["L_data"][~off], rtol=1e-14, atol=0)
    assert P.stats is not None
    P.close()


@pytest.mark.parametrize("nm", ["er300_xfrac", "er2000_x0", "er2000_xfrac"])
@pytest.mark.parametrize("variant", [1, 2])
def test_spmv_matches_scipy(nm, variant):
    g = load_golden(nm)
    P = problem_of(g)
    P.set_x(g["x"])
    L = oracle_of(g).laplacian(g["x"])
    rng = np.random.default_rng(0)
    v = rng.normal(size=int(g["n"]))
    y = P.spmv(v, variant=variant)
    ref = L @ v
    assert np.abs(y - ref).max() <= 1e-13 * np.abs(ref).max()
    P.close()


def test_spmv_long_rows_all_candidates():
    g = load_golden("er300_x0")
    P = problem_of(g)
    x = np.ones(len(g["cw"]))
    P.set_x(x)
    L = oracle_of(g).laplacian(x)
    v = np.random.default_rng(1).normal(size=int(g["n"]))
    for variant in (0, 1, 2):
        y = P.spmv(v, variant=variant)
        assert np.abs(y - L @ v).max() <= 1e-13 * np.abs(L @ v).max()
    P.close()


# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nm,exact", [("k5", 5.0), ("p2", 2.0), ("p3", 1.0),
                                      ("p50", 2 - 2 * np.cos(np.pi / 50)),
                                      ("c12", 2 - 2 * np.cos(2 * np.pi / 12)), ("star9", 1.0)])
def test_find_fiedler_pair_closed_forms(nm, exact):
    g = load_golden("fiedler_" + nm)
    n = int(g["n"])
    L = weight_graph_lap_from_edge_list([Edge(int(a), int(b), float(w)) for a, b, w in zip(g["ei"], g["ej"], g["ew"])], n)
    lam, v, X = find_fiedler_pair(L)                     # reference tests/utils/test_fiedler.py:32
    assert np.isclose(lam, exact)                        # the reference's own assertion (K5 -> 5)
    assert abs(lam - exact) <= LAM_RTOL * exact
    assert abs(lam - g["lam"]) <= LAM_RTOL * exact
    q = min(4, n - 1)
    assert X.shape == (n, q) and v.shape == (n,)
    assert abs(np.linalg.norm(v) - 1) < 1e-12 and abs(v.sum()) < 1e-10
    assert np.abs(L @ v - lam * v).sum() / abs(L).sum(axis=1).max() < 1e-8      # nx:246 rule
    assert np.array_equal(X[:, 0], v)
    assert np.abs(X.T @ X - np.eye(q)).max() < 1e-8 and np.abs(X.sum(axis=0)).max() < 1e-8


@pytest.mark.parametrize("nm", ["petersen_x0", "er300_x0", "er300_xfrac", "er2000_x0", "er2000_xfrac"])
def test_fiedler_pair_and_gradient_vs_reference(nm):
    g = load_golden(nm)
    P = problem_of(g)
    P.set_x(g["x"])
    lam, v, X = P.fiedler(tol=1e-8, x0=reference_start_block(int(g["n"]))[:, 0].copy(), q=min(4, int(g["n"]) - 1))
    assert abs(lam - g["lam"]) <= LAM_RTOL * abs(g["lam"])
    assert P.stats.residual < 1e-8
    va = sign_align(v, g["v"])
    assert np.abs(va - g["v"]).max() <= 2e-6
    grad = P.gradient()
    assert np.array_equal(grad, oracle.supergradient(v, g["ci"], g["cj"], g["cw"]))      # bit-exact
    assert np.abs(grad - g["grad"]).max() <= 1e-5 * np.abs(g["grad"]).max()
    # Ritz block: orthonormal, column 0 the Fiedler vector
    assert np.abs(X.T @ X - np.eye(X.shape[1])).max() < 1e-6
    # identity: sum_k x_k g_k + v^T L_fixed v = lambda_2
    Lf = oracle.laplacian_from_edges(g["fi"], g["fj"], g["fw"], int(g["n"]))
    xs = np.where(g["x"] > 1e-10, g["x"], 0.0)
    assert abs(xs @ grad + v @ (Lf @ v) - lam) <= 1e-9 * max(1.0, abs(lam))
    P.close()


def golden_lambda(g, key):
    """lambda_2 to compare with at 1e-8.  kitti_02 / ais2klinik (round 4) are stiff chains, lambda_2 / ||L||_inf down to 1e-8:
    there the reference's own stop rule (nx:246) leaves ITS lambda_2 accurate to 8e-9 / 3e-7 relative only, so those fixtures also
    hold lambda_2 of the reference's own MAC.laplacian(x) from SciPy's shift-invert Lanczos (`<key>_exact`, make_golden.py
    g2o_extra).  The HIP value must hit the exact one to 1e-8; the reference's must lie within its own measured deviation."""
    if key + "_exact" in g:
        exact = float(g[key + "_exact"])
        assert abs(float(g[key]) - exact) <= 5e-7 * exact          # (documents how far the reference itself is off)
        return exact
    return float(g[key])


def golden_grad(g):
    """Supergradient at x_init to compare with: the exact one where the fixture holds it (stiff chains: the reference's 1e-8
    residual leaves its eigenvector -- and with it its gradient -- off by up to 2e-3 of the largest entry; bounded here)."""
    if "grad_init_exact" in g:
        assert np.abs(g["grad_init"] - g["grad_init_exact"]).max() <= 5e-3 * np.abs(g["grad_init_exact"]).max()
        return g["grad_init_exact"]
    return g["grad_init"]


@pytest.mark.parametrize("nm", ["intel", "sphere2500", "city10000", "kitti_05", "kitti_02", "ais2klinik"])
def test_pose_graph_fiedler(nm):
    """lambda_2 to 1e-8 and the supergradient against the EXACT pair of the reference's own MAC.laplacian(x_init)
    (tests/golden/g2o_exact_<name>.npz, make_golden.py g2o_exact: dense eigh / shift-invert Lanczos; round 5 -- rounds 1-4
    compared with the reference's own 1e-8-residual gradient under a blanket 2e-4 of the largest entry).  The tolerance is what
    the stop rule allows and no more: a unit vector whose residual is ||r||_2 lies within sin(theta) <= ||r||_2 / (lambda_3 - rho)
    of the eigenvector, ||r||_2 <= ||r||_1 = residual ||L||_inf as MEASURED by the solve (machip_solve_stats.residual < 1e-8), so
    |dv_i - dv_j| <= delta = sqrt(2) sin(theta) and |dg_k| <= w_k (2 |v_i - v_j| delta + delta^2) entry by entry."""
    g = load_golden("g2o_" + nm)
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    lam = mac.evaluate_objective(g["x_init"])
    assert abs(lam - golden_lambda(g, "lam_init")) <= LAM_RTOL * g["lam_init"]
    f, grad = mac.problem(g["x_init"])
    assert abs(f - golden_lambda(g, "lam_init")) <= LAM_RTOL * g["lam_init"]
    ex_path = os.path.join(ROOT, "tests", "golden", f"g2o_exact_{nm}.npz")
    if os.path.exists(ex_path):
        ex = np.load(ex_path)
        assert abs(f - float(ex["lam_init_exact"])) <= LAM_RTOL * float(ex["lam_init_exact"])
        assert abs(float(g["lam_init"]) - float(ex["lam_init_exact"])) <= 1e-9 * float(ex["lam_init_exact"])      # (how far the reference itself is off)
        res, lnorm = mac.last_stats["residual"], mac.last_stats["lnorm"]
        assert res < 1e-8 and abs(lnorm - float(ex["lnorm_init"])) <= 1e-9 * lnorm
        sin_theta = res * lnorm / (float(ex["lam3_init"]) - f)
        delta = np.sqrt(2.0) * 1.01 * sin_theta
        ve = ex["v_init_exact"]
        d = np.abs(ve[g["ci"]] - ve[g["cj"]])
        bound = g["cw"] * (2.0 * d * delta + delta * delta)
        err = np.abs(grad - ex["grad_init_exact"])
        assert np.all(err <= bound + 1e-300), (nm, float((err / np.maximum(bound, 1e-300)).max()))
        # what that band is, relative to the largest entry (rounds 1-4 allowed 2e-4 everywhere): stated, and the measured error well inside
        assert err.max() <= 2e-4 * np.abs(ex["grad_init_exact"]).max(), (nm, err.max() / np.abs(ex["grad_init_exact"]).max())
        lam_all = mac.evaluate_objective(np.ones(len(g["cw"])))
        assert abs(lam_all - float(ex["lam_all_exact"])) <= LAM_RTOL * float(ex["lam_all_exact"])
    else:
        assert np.abs(grad - golden_grad(g)).max() <= 2e-4 * np.abs(g["grad_init"]).max()
        lam_all = mac.evaluate_objective(np.ones(len(g["cw"])))
        assert abs(lam_all - golden_lambda(g, "lam_all")) <= LAM_RTOL * g["lam_all"]


# ---- preconditioned eigen-solver mode (LOBPCG + tridiagonal chain solve, precond.h) ----------
@pytest.mark.parametrize("nm", ["intel", "sphere2500", "city10000", "kitti_05", "kitti_02", "ais2klinik"])
def test_preconditioned_mode_on_pose_graphs(nm):
    """fiedler_method='tracemin_pcg' (the reference's preconditioned flavour) routes to the
    preconditioned HIP mode; same pair as the reference / the Lanczos mode to the same tolerances."""
    g = load_golden("g2o_" + nm)
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="tracemin_pcg")
    ref = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="hip_lanczos")
    for x, key in [(g["x_init"], "lam_init"), (np.ones(len(g["cw"])), "lam_all")]:
        lam = mac.evaluate_objective(x)
        assert abs(lam - golden_lambda(g, key)) <= LAM_RTOL * g[key]
        assert mac.last_stats["residual"] < 1e-8
    f, grad = mac.problem(g["x_init"])
    f2, grad2 = ref.problem(g["x_init"])
    assert abs(f - f2) <= LAM_RTOL * f2
    assert np.abs(grad - grad2).max() <= 2e-4 * np.abs(grad2).max()   # both eigenvectors carry the 1e-8 residual
    assert np.abs(grad - golden_grad(g)).max() <= 2e-4 * np.abs(g["grad_init"]).max()
    v = mac._dev.fiedler()[1]
    assert abs(np.linalg.norm(v) - 1) < 1e-12 and abs(v.sum()) < 1e-9


def test_preconditioned_mode_full_solve_matches_reference_trajectory():
    g = load_golden("g2o_kitti_05")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="tracemin_pcg")
    rounded, w, u = mac.solve(int(g["k"]), g["x_init"], max_iters=20, use_cache=False)
    ft = np.array([t[0] for t in mac.trace])
    assert np.allclose(ft, g["f_traj"][:len(ft)], rtol=1e-6)
    assert np.array_equal([t[3] for t in mac.trace], g["supp"][:len(ft)])
    assert abs(u - g["upper"]) <= 1e-5 * abs(g["upper"])
    # warm start through the same mode
    mac2 = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="tracemin_pcg")
    mac2.solve(int(g["k"]), g["x_init"], max_iters=20, use_cache=True)
    assert np.allclose([t[0] for t in mac2.trace], ft, rtol=1e-6)


def test_preconditioned_mode_fallbacks_and_errors():
    rng = np.random.default_rng(11)
    # (a) no chain at all (random graph): T = diag + sigma, still a valid preconditioner or a clean fallback
    n = 3000
    ci, cj = make_er(n, 0.004, 3)
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi[::7], fi[::7] + 1, np.ones(len(fi[::7])), ci, cj, rng.random(len(ci)) + 0.5)
    P.set_x(np.ones(len(ci)))
    P.set_solver(1); lam1, v1, _ = P.fiedler()
    P.set_solver(2); lam2, v2, _ = P.fiedler()
    assert abs(lam1 - lam2) <= LAM_RTOL * lam1 and P.stats.residual < 1e-8
    P.close()
    # (b) n above the register-resident solver's limit (16 384): the global-scratch tridiagonal kernels run
    n = 20000
    fi = np.arange(n - 1, dtype=np.int32)
    ci = rng.integers(0, n - 50, 4000).astype(np.int32); cj = (ci + rng.integers(2, 50, 4000)).astype(np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.full(n - 1, 50.0), ci, cj, np.full(4000, 20.0))
    P.set_x(np.ones(4000))
    P.set_solver(2); lam2, v2, _ = P.fiedler(); it2 = int(P.stats.lanczos_steps)
    assert P.stats.residual < 1e-8
    P.set_solver(1); lam1, v1, _ = P.fiedler(); it1 = int(P.stats.lanczos_steps)
    assert abs(lam1 - lam2) <= LAM_RTOL * lam1 and it1 > 0 and it2 > 0
    assert min(np.abs(v1 - v2).max(), np.abs(v1 + v2).max()) < 1e-4
    P.close()
    # a pure path of that size has the closed form 2 - 2 cos(pi / n)
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), np.array([0], np.int32), np.array([2], np.int32), np.ones(1))
    P.set_x(np.zeros(1))
    lam, _, _ = P.fiedler()
    assert abs(lam - (2 - 2 * np.cos(np.pi / n))) <= LAM_RTOL * lam and P.stats.lanczos_steps < 100
    P.close()
    # (c) disconnected graph: same status as the Lanczos mode
    n = 600
    fi = np.concatenate([np.arange(0, 299), np.arange(300, 599)]).astype(np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.ones(len(fi)), np.array([5], np.int32), np.array([50], np.int32), np.ones(1))
    P.set_x(np.ones(1))
    P.set_solver(2)
    with pytest.raises(_lib.Disconnected):
        P.fiedler()
    with pytest.raises(AssertionError):
        P.set_solver(7)
    P.close()


def test_lanczos_hands_over_to_the_exact_mode_on_its_own_forecast():
    """Chain of 9 000 nodes + 1 800 short closures (20 % of n: denser than the static rule for the preconditioned mode admits, no history
    on a fresh handle): the automatic mode starts with Lanczos, whose convergence forecast says thousands of steps to go, and hands
    over to the exact chain + closures mode (solver.h, switch_est_us) -- far fewer steps than the forced Lanczos solve, the same
    lambda_2 (1e-8) and vector, SciPy's shift-invert value; a second solve starts in the exact mode.  A warm start from the converged
    vector stays with Lanczos (its forecast is short)."""
    import scipy.sparse.linalg as spla
    n, s_ = 9000, 1800
    rng = np.random.default_rng(21)
    fi = np.arange(n - 1, dtype=np.int32)
    ci = rng.choice(n - 60, s_, replace=False).astype(np.int32); cj = (ci + rng.integers(2, 50, s_)).astype(np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.full(n - 1, 40.0), ci, cj, np.full(s_, 15.0))
    P.set_start(reference_start_block(n)[:, 0].copy())
    P.set_x(np.ones(s_))
    P.set_solver(1); lam_l, v_l, _ = P.fiedler(); st_l = int(P.stats.lanczos_steps)
    P.close()
    P = _lib.Problem(n, fi, fi + 1, np.full(n - 1, 40.0), ci, cj, np.full(s_, 15.0))
    P.set_start(reference_start_block(n)[:, 0].copy())
    P.set_x(np.ones(s_))
    P.set_solver(0)
    lam_a, v_a, _ = P.fiedler(); st_a = int(P.stats.lanczos_steps)
    lam_b, v_b, _ = P.fiedler(); st_b = int(P.stats.lanczos_steps)
    assert st_l > 1500 and 128 < st_a < st_l // 3 and st_b <= 40, (st_l, st_a, st_b)      # (1 977 forced-Lanczos steps with the landscape-weighted cold start, 2 100 without)
    assert abs(lam_a - lam_l) <= LAM_RTOL * lam_l and abs(lam_b - lam_l) <= LAM_RTOL * lam_l and P.stats.residual < 1e-8
    assert np.abs(sign_align(v_a, v_l) - v_l).max() < 1e-5
    ip, ix, da = P.laplacian_csr()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    w = spla.eigsh(L + 1e-9 * sp.identity(n), k=2, sigma=0, which="LM", return_eigenvectors=False)
    assert abs(np.sort(w)[1] - 1e-9 - lam_a) <= 1e-7 * lam_a
    P.close()


def test_auto_mode_learns_a_stiff_problem_from_step_counts():
    """city10000 with the first 20 % of the closures selected is stiff (~10^4 Lanczos steps) although its closure density is above
    the static threshold.  (a) Round-3 rule (option exact_big = 0): the automatic mode sees that from the first solve's step count and
    runs the preconditioned mode from then on.  (b) Round 4: the first solve itself hands over -- the Lanczos loop's own forecast of
    the steps to go exceeds the exact chain + closures mode's estimated cost (2 137 closures) twice in a row --, later solves start
    in the exact mode.  Deterministic either way: counts and bit-reproducible forecasts, never timings."""
    g = load_golden("g2o_city10000")
    m = len(g["cw"])
    x = np.zeros(m); x[: m // 5] = 1.0
    lams = []
    for big in ("0", "1"):
        P = problem_of(g)
        P.set_option("exact_big", int(big))
        P.set_x(x)
        P.set_solver(0)
        lam1, _, _ = P.fiedler(); s1 = int(P.stats.lanczos_steps)
        lam2, _, _ = P.fiedler(); s2 = int(P.stats.lanczos_steps)
        lam3, _, _ = P.fiedler(); s3 = int(P.stats.lanczos_steps)
        P.close()
        if big == "0":
            assert s1 > 2500 and s2 * 6 < s1 and s3 == s2, (s1, s2, s3)
        else:
            assert 128 < s1 < 1500 and s2 <= 40 and s3 == s2, (s1, s2, s3)      # (Lanczos steps before the hand-over + exact iterations)
        assert abs(lam1 - lam2) <= LAM_RTOL * lam1 and lam2 == lam3
        lams.append(lam1)
    assert abs(lams[0] - lams[1]) <= LAM_RTOL * lams[0]


@pytest.mark.parametrize("seed", range(8))
def test_solver_modes_agree_on_random_chain_graphs(seed):
    """Seeded sweep over pose-graph-like inputs (chain with weights over three decades, random closures,
    sizes on both sides of the single-workgroup / register-resident limits): Lanczos, preconditioned and
    automatic modes against a dense eigen-solve (n <= 2500) and against each other."""
    rng = np.random.default_rng(100 + seed)
    n = int([300, 1024, 1025, 2500, 3073, 6000, 16384, 16400][seed])
    ncl = int(rng.integers(3, max(4, n // 5)))
    fi = np.arange(n - 1, dtype=np.int32)
    fw = 10.0 ** rng.uniform(0, 3, n - 1)
    a = rng.integers(0, n, ncl); b = rng.integers(0, n, ncl)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = 10.0 ** rng.uniform(0, 2.5, len(ci))
    x = rng.random(len(ci)); x[rng.random(len(ci)) < 0.3] = 0.0
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(x)
    lams, vecs = [], []
    for mode in (1, 2, 0):
        P.set_solver(mode)
        lam, v, _ = P.fiedler()
        assert P.stats.residual < 1e-8 and abs(np.linalg.norm(v) - 1) < 1e-12 and abs(v.sum()) < 1e-8
        lams.append(lam); vecs.append(v)
    assert max(lams) - min(lams) <= LAM_RTOL * min(lams)
    if n <= 2500:
        L = oracle.mac_laplacian(oracle.laplacian_from_edges(fi, fi + 1, fw, n), ci.astype(np.int64), cj.astype(np.int64), cw, x, n)
        w = np.linalg.eigvalsh(L.toarray())
        assert abs(lams[0] - w[1]) <= LAM_RTOL * w[1]
        if w[2] - w[1] > 1e-3 * w[1]:      # simple eigenvalue: the vectors agree too
            for v in vecs[1:]:
                assert min(np.abs(v - vecs[0]).max(), np.abs(v + vecs[0]).max()) < 1e-4
    P.close()


@pytest.mark.parametrize("hub_deg", [40, 600, 1500])
def test_single_workgroup_kernel_with_hub_rows(hub_deg):
    """The single-workgroup Lanczos kernel (persist.h) keeps a row's band and two closures in registers; further closures go
    through a flat product list -- held in registers up to 1 024 entries, in LDS beyond.  A hub node with 40 / 600 / 1 500
    closures (plus random ones) exercises both forms and long row segments, against a dense eigen-solve."""
    rng = np.random.default_rng(hub_deg)
    n = 2000
    fi = np.arange(n - 1, dtype=np.int32)
    fw = rng.uniform(50.0, 300.0, n - 1)
    hub = 700
    others = rng.choice(np.setdiff1d(np.arange(n), [hub - 1, hub, hub + 1]), hub_deg, replace=False)
    a = np.r_[np.full(hub_deg, hub), rng.integers(0, n, 200)]; b = np.r_[others, rng.integers(0, n, 200)]
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = rng.uniform(1.0, 50.0, len(ci))
    x = np.ones(len(ci))
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(x)
    P.set_solver(1)
    lam, v, _ = P.fiedler()
    L = oracle.mac_laplacian(oracle.laplacian_from_edges(fi, fi + 1, fw, n), ci.astype(np.int64), cj.astype(np.int64), cw, x, n)
    w = np.linalg.eigvalsh(L.toarray())
    assert abs(lam - w[1]) <= LAM_RTOL * w[1] and P.stats.residual < 1e-8
    lam2, v2, _ = P.fiedler()
    assert lam2 == lam and np.array_equal(v, v2)          # run-to-run identical
    P.close()


def test_auto_mode_picks_preconditioned_solver_on_sparse_chain_graphs():
    """Automatic selection: a chain with few closures runs the preconditioned mode (an order of
    magnitude fewer dependent launches), a dense-closure graph the Lanczos mode; same lambda_2."""
    g = load_golden("g2o_kitti_05")
    P = problem_of(g)
    P.set_x(g["x_init"])
    P.set_solver(1); lam_l, _, _ = P.fiedler(); steps_l = P.stats.lanczos_steps
    P.set_solver(0); lam_a, _, _ = P.fiedler(); steps_a = P.stats.lanczos_steps
    assert abs(lam_l - lam_a) <= LAM_RTOL * lam_l
    assert steps_a * 5 < steps_l
    P.close()


# --------------------------------------------------------------------------------------------
def test_topk_matches_oracle_and_handles_ties():
    g = load_golden("er2000_x0")
    P = problem_of(g)
    P.set_x(g["x"])
    P.fiedler(want_vec=False)
    grad = P.gradient()
    m = len(grad)
    for k in [1, 7, m // 10, m // 2, m - 1, m]:
        s = P.lp_topk(k)
        assert s.sum() == k
        kth = np.sort(grad)[-k]
        assert np.all(s[grad > kth] == 1) and np.all(s[grad < kth] == 0)
        if np.sum(grad == kth) == 1:
            assert np.array_equal(s, oracle.solve_subset_box_lp(grad, k))
    assert P.lp_topk(0).sum() == 0
    P.close()


def test_topk_tie_break_lowest_index():
    # duplicated candidate pairs give exactly equal gradient entries
    n = 12
    fixed = [Edge(i, i + 1, 1.0 + 0.1 * i) for i in range(n - 1)]
    pairs = [(0, 5)] * 4 + [(2, 9)] * 3 + [(1, 3)] * 2 + [(4, 11), (0, 5), (6, 8)]
    cand = [Edge(a, b, 1.0) for a, b in pairs]
    mac = MAC(fixed, cand, n)
    m = len(cand)
    P = mac._dev
    P.set_x(np.zeros(m))
    P.fiedler(want_vec=False)
    grad = P.gradient()
    assert grad[0] == grad[1] == grad[2] == grad[3] == grad[10]
    for k in range(0, m + 1):
        s = P.lp_topk(k)
        assert s.sum() == k
        if k == 0:
            continue
        kth = np.sort(grad)[-k]
        assert np.all(s[grad > kth] == 1) and np.all(s[grad < kth] == 0)
        ties = np.nonzero(grad == kth)[0]
        need = int(k - np.sum(grad > kth))
        assert np.array_equal(np.nonzero(s[ties])[0], np.arange(need))     # lowest indices win


# --------------------------------------------------------------------------------------------
def test_petersen_solve_matches_reference():
    g = load_golden("petersen_solve_k3")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), 10)
    rounded, w, u = mac.solve(3, g["x_init"], max_iters=5)
    assert np.allclose([t[0] for t in mac.trace], g["f_traj"], rtol=1e-8)
    assert np.allclose(w, g["unrounded"], atol=1e-12)
    assert np.array_equal(rounded, g["rounded"])
    assert abs(u - g["upper"]) <= 1e-7 * abs(g["upper"])
    assert abs(mac.evaluate_objective(np.zeros(6)) - g["lam_tree"]) <= LAM_RTOL * g["lam_tree"]
    assert abs(mac.evaluate_objective(np.ones(6)) - 2.0) <= 2 * LAM_RTOL


def test_petersen_sweep_like_reference_test_mac():
    """reference tests/solvers/test_mac.py:35-60 (its own assertion) + golden values."""
    rows = load_golden("petersen_sweep")["rows"]
    g = load_golden("petersen_solve_k3")
    fixed, cand = edges_of(g, "f"), edges_of(g, "c")
    for pct, k, l_init, l_un, l_r, up in rows:
        k = int(k)
        x_init = np.zeros(len(cand))
        x_init[:k] = 1.0
        mac = MAC(fixed, cand, 10)
        result, unrounded, upper = mac.solve(k, x_init, max_iters=100)
        init_l2 = mac.evaluate_objective(x_init)
        un_l2 = mac.evaluate_objective(unrounded)
        assert un_l2 >= init_l2 - 1e-12
        assert abs(init_l2 - l_init) <= LAM_RTOL * l_init
        # 100 FW iterations on this symmetric graph hit near-ties (|dg| ~ 1e-14) in the top-k LP, so
        # the trajectories legitimately fork and FW iterates are not monotone; what
        # must hold across the two runs are the bound relations around the common optimum f*:
        # each run's iterate value <= f* <= the other run's dual upper bound.
        assert un_l2 <= up + 1e-9 and l_un <= upper + 1e-9
        assert un_l2 <= upper + 1e-9


@pytest.mark.parametrize("nm", ["er300_solve", "er2000_solve"])
def test_er_solve_trajectory(nm):
    g = load_golden(nm)
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    rounded, w, u = mac.solve(int(g["k"]), g["x_init"], max_iters=int(g["max_iters"]))
    assert np.allclose([t[0] for t in mac.trace], g["f_traj"], rtol=LAM_RTOL)
    assert np.array_equal([t[3] for t in mac.trace], g["supp"])
    assert np.allclose(w, g["unrounded"], atol=1e-12)
    assert np.array_equal(rounded, g["rounded"])
    assert abs(u - g["upper"]) <= 1e-6 * abs(g["upper"])


@pytest.mark.parametrize("nm", ["intel", "sphere2500", "kitti_05", "city10000", "kitti_02", "ais2klinik"])
def test_pose_graph_solve_trajectory(nm):
    g = load_golden("g2o_" + nm)
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    rounded, w, u = mac.solve(int(g["k"]), g["x_init"], max_iters=20, use_cache=False)
    ft = np.array([t[0] for t in mac.trace])
    if nm == "city10000":
        # all 10 688 closure weights are 100: near-ties at the K-th gradient entry (relative gaps 1e-5..1e-3,
        # golden ref_gap_rel) are decided differently by two 1e-8-accurate eigenvectors from some iteration on;
        # test_city10000_vertices_until_the_fork pins WHERE and shows the HIP vertex is the exact one there.
        # Up to the fork the trajectories are identical, afterwards they stay in the same regime.
        fork = city_fork_iteration()
        assert np.allclose(ft[:fork + 1], g["f_traj"][:fork + 1], rtol=1e-6)
        assert np.array_equal([t[3] for t in mac.trace][:fork + 1], g["supp"][:fork + 1])
        assert np.all(np.abs(ft - g["f_traj"][:len(ft)]) <= 0.05 * g["f_traj"][:len(ft)])
        assert abs(u - g["upper"]) <= 2e-2 * abs(g["upper"])
        assert int(rounded.sum()) == int(g["k"])
        return
    assert np.allclose(ft, g["f_traj"][:len(ft)], rtol=1e-6)
    assert np.array_equal([t[3] for t in mac.trace], g["supp"][:len(ft)])
    assert abs(u - g["upper"]) <= 1e-5 * abs(g["upper"])
    assert abs(mac.evaluate_objective(rounded) - g["lam_rounded"]) <= 1e-6 * g["lam_rounded"] or \
        np.array_equal(rounded, g["rounded"])
    # warm start (the working form of MAC.Cache) reaches the same optimum
    mac2 = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    r2, w2, u2 = mac2.solve(int(g["k"]), g["x_init"], max_iters=20, use_cache=True)
    assert np.allclose([t[0] for t in mac2.trace], ft, rtol=1e-6)


def city_fork_iteration():
    """First iteration whose LP vertex the golden lists as decided by an exact solve (20 = never)."""
    ex = [int(t) for t in load_golden("city10000_vertices")["exact_at"]]
    return min(ex) if ex else 20


def test_city10000_vertices_until_the_fork():
    """BASELINE.json configs[4], city10000: LP vertex by LP vertex against the reference's own run
    (tests/golden/city10000_vertices.npz).  While the vertices agree both runs hold the same x and lambda_2
    must agree to 1e-8.  At the first iteration where they differ the golden holds the EXACT vertex (dense
    numpy eigh of the reference's own MAC.laplacian(x), stable sort) and the HIP vertex must equal it: the
    reference's 1e-8-accurate eigenvector, not the HIP one, mis-ranks the near-tie.  After a fork the two
    runs optimise from different points and are only compared in test_pose_graph_solve_trajectory."""
    g = load_golden("g2o_city10000"); gv = load_golden("city10000_vertices")
    k = int(g["k"])
    P = problem_of(g)
    P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
    P.set_x(g["x_init"])
    forked = False
    for it in range(20):
        f, dual, gn = P.fw_step(k, it)
        assert abs(f - gv["f_traj"][it]) <= LAM_RTOL * abs(f), (it, f, gv["f_traj"][it])
        s = np.nonzero(P.lp_topk(k))[0]
        if not np.array_equal(s, gv["ref_s"][it]):
            key = f"exact_s{it}"
            assert key in gv, f"vertices differ at iteration {it} and the golden holds no exact vertex there"
            assert abs(f - float(gv[f"exact_lam{it}"])) <= 1e-10 * abs(f)
            assert np.array_equal(s, gv[key]), (it, len(np.setdiff1d(s, gv[key])))
            assert not np.array_equal(gv["ref_s"][it], gv[key])        # it is the reference that missed
            forked = True
            break
        P.fw_commit()
    assert forked == (city_fork_iteration() < 20)
    P.close()


# --------------------------------------------------------------------------------------------
# Teacher forcing: the HIP path evaluated ON THE REFERENCE'S OWN ITERATES.  x_i is rebuilt from the LP vertices the
# reference run stored (x_{i+1} = x_i + 2/(i+2) (s_i - x_i), NumPy arithmetic exactly as frankwolfe.py:76 does it), so
# every one of the late, stiff iterates is checked by value -- lambda_2 to 1e-8 -- and not through bounded drift of a
# free-running trajectory.  The LP vertex is compared as a set: entries may only differ where the gradient lies within
# the band a 1e-8-accurate eigenvector cannot resolve around the K-th value T: g = w d^2 with d = v_i - v_j known to
# dv = 2e-6 (the Fiedler-vector tolerance of this file) gives |dg| <= 2 sqrt(w T) dv + w dv^2 (the golden stores the
# reference's own relative gap at the K-th place: 1e-7 .. 1e-4, near-ties are the rule with equal weights).
# --------------------------------------------------------------------------------------------
def _teacher_forced(P, k, x0, ref_vertices, lam_ref, wmax=1.0, dv=2e-6, steps_out=None, stats_out=None):
    x = np.array(x0, dtype=np.float64)
    worst = 0.0
    for i, lam_gold in enumerate(lam_ref):
        P.set_x(x)
        lam, _, _ = P.fiedler(tol=1e-8, want_vec=False)
        if steps_out is not None:
            steps_out.append(int(P.stats.lanczos_steps))
        if stats_out is not None:
            stats_out.append(P.stats.asdict())
        rel = abs(lam - lam_gold) / abs(lam_gold)
        worst = max(worst, rel)
        assert rel <= LAM_RTOL, (i, lam, lam_gold, rel)
        assert P.stats.residual < 1e-8
        g = P.gradient()
        s = P.lp_topk(k)
        s_ref = ref_vertices(i)
        assert int(s.sum()) == k == int(s_ref.sum())
        diff = np.nonzero(s != s_ref)[0]
        if len(diff):
            thr = np.partition(g, len(g) - k)[len(g) - k]          # K-th largest gradient entry
            band = 2.0 * np.sqrt(wmax * thr) * dv + wmax * dv * dv
            assert np.all(np.abs(g[diff] - thr) <= band), (i, len(diff), float(np.abs(g[diff] - thr).max()), band)
        x = x + 2.0 / (i + 2) * (s_ref - x)                          # the reference's iterate, not ours
    return worst


@pytest.mark.parametrize("precision", [0, 1])
def test_teacher_forced_city10000_all_twenty_reference_iterates(precision):
    """BASELINE.json configs[4] (city10000): lambda_2 on x_0 .. x_19 of the reference's own run (stiff from x_1 on:
    lambda_2 = 0.0012, thousands of Lanczos steps or the preconditioned mode), fp64 and mixed precision."""
    g = load_golden("g2o_city10000"); gv = load_golden("city10000_vertices")
    m, k = len(g["cw"]), int(g["k"])
    P = problem_of(g)
    P.set_precision(precision)
    P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())

    def vert(i):
        s = np.zeros(m); s[gv["ref_s"][i]] = 1.0
        return s
    _teacher_forced(P, k, gv["x_init"], vert, gv["f_traj"], wmax=float(np.max(g["cw"])))
    P.close()


@pytest.mark.parametrize("precision", [0, 1])
def test_teacher_forced_config2_all_twenty_reference_iterates(precision):
    """BASELINE.json configs[1] (ER N = 10k): the reference's 20 iterates (tests/golden/er10k_vertices.npz, two CPU
    hours of TraceMIN + SuperLU), lambda_2 to 1e-8 on each."""
    import bench
    w = bench.make_workload("c2")
    gv = load_golden("er10k_vertices")
    m, k = len(w["cw"]), w["k"]
    assert int(gv["m"]) == m and int(gv["k"]) == k
    assert np.array_equal(np.nonzero(w["x0"])[0], load_golden("er10k_x0")["x0_idx"])
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_precision(precision)
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    bits = gv["ref_s_bits"]
    steps = []
    _teacher_forced(P, k, w["x0"], lambda i: np.unpackbits(bits[i])[:m].astype(np.float64), gv["f_traj"], steps_out=steps)
    P.close()
    if precision == 0:      # landscape-weighted cold start (profiles/r5_landscape.md): 3 482 steps on these 20 matrices, 4 044 from the start column as drawn
        assert sum(steps) <= 3750, sum(steps)


@pytest.mark.parametrize("form", ["auto", "panel", "panel_records", "gather", "panel_mixed"])
def test_teacher_forced_config4_all_twenty_iterates(form):
    """BASELINE.json configs[3] (ER N = 100k, 2M candidates): lambda_2 on ALL 20 iterates of the reference's loop with
    ARPACK (tol 1e-13, residual <= 2e-13) standing in for the sparse LU that does not finish at this size
    (tests/golden/er100k_arpack.npz, generator `ER100K_ITERS=20 make_golden.py er100k_arpack`: the reference's
    MAC.laplacian / solve_subset_box_lp / update, SciPy eigsh).  These are the iterates the bench runs: nnz 0.7 M .. 4.0 M,
    the dense ones (6-19) are where the column-panel step spends its steps.  Forms: the automatic choice (gather step on
    the sparse iterates, panel step on the dense ones), the panel step forced onto every iterate -- round 6's shifted recurrence with
    its 8-byte operand (panel_u.h: k_pan_mul8 + k_pan_finu, 6 x 42 cells) and round 3's record form (k_pan_mul + k_pan_fin, 12 x 21) --, the
    gather step forced onto every iterate; and the panel step in the MIXED mode (machip_set_precision(1), BASELINE configs[4]'s technique at the one
    config whose step is bound by bytes): fp64 tile values until the residual estimate is below 3e-4 ||L||, their fp32 copy (6 bytes per entry
    instead of 10) from 32 steps behind that point on -- same stop rule on the fp64 matrix, lambda_2 to 1e-8 on every iterate.  (The one-launch panel step and the diagonally preconditioned LOBPCG forms of round 4
    -- measured slower, profiles/r4_c4_one_launch_step.md -- are compiled only with -DMACHIP_EXPERIMENTS and no longer tested here.)"""
    import bench
    w = bench.make_workload("c4")
    gv = load_golden("er100k_arpack")
    m, k = len(w["cw"]), w["k"]
    assert int(gv["m"]) == m and int(gv["k"]) == k and len(gv["lam_traj"]) == 20
    assert float(np.max(gv["residual"])) < 1e-12
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    bits = gv["ref_s_bits"]
    vert = lambda i: np.unpackbits(bits[i])[:m].astype(np.float64)    # noqa: E731
    P.set_options(**{"auto": {}, "panel": {"panel": 1}, "panel_records": {"panel": 1, "panel_u": 0}, "gather": {"panel": 0}, "panel_mixed": {"panel": 1}}[form])
    if form == "panel_mixed":
        P.set_precision(1)
    steps, stats = [], []
    _teacher_forced(P, k, w["x0"], vert, gv["lam_traj"], steps_out=steps, stats_out=stats)
    P.close()
    if form == "panel_mixed":       # the late steps really read fp32 tiles (a solve shorter than ~150 steps never gets there), and nothing restarted
        # (measured: 838 of 4 226 steps on the bench trajectory -- Lanczos converges superlinearly, the estimate passes 3e-4 ||L|| late)
        assert sum(1 for st in stats if st["steps_lowp"] > 0) >= 15 and sum(st["steps_lowp"] for st in stats) > 0.1 * sum(steps), [(st["lanczos_steps"], st["steps_lowp"]) for st in stats]
        assert all(st["restarts"] == 0 for st in stats)
    assert sum(steps) <= 4250, sum(steps)       # landscape-weighted cold start: 3 913 steps on these 20 matrices, 4 632 from the start column as drawn


def test_landscape_field_matches_numpy_and_weighted_cold_start_needs_fewer_steps():
    """Late round 5 (kernels.h k_land_*, solver.h landscape_start): a cold Lanczos start is the stored start vector times
    (u / max u)^128, u = three Jacobi sweeps on L u = 1 from u = 1/diag -- the Fiedler vector of a sparse random graph is localised
    on the peaks of that landscape (participation ratio 1-5 on every iterate of configs[1] / configs[3]).  (a) machip_landscape
    against NumPy on the assembled Laplacian, both SpMV row mappings; (b) on the first two matrices of the configs[1] trajectory the
    weighted start ends at the same lambda_2 (1e-8, residual rule as ever) in fewer steps; (c) with the option off the solve is
    bit-identical to handing the same vector over as the caller's own guess, which is never weighted; (d) a warm start is not
    weighted either."""
    import bench
    w = bench.make_workload("c2")
    n, m, k = w["n"], len(w["cw"]), w["k"]
    z = reference_start_block(n)[:, 0].copy()
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(z)
    P.set_x(w["x0"])
    ip, ix, da = P.laplacian_csr()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    d = L.diagonal()
    for spmv in (1, 2):
        P.set_option("spmv", spmv)
        assert np.allclose(P.landscape(0), 1.0 / d, rtol=1e-15, atol=0)
        for sweeps in (1, 2, 3, 5):
            u = 1.0 / d
            for _ in range(sweeps):
                u = u + (1.0 - L @ u) / d
            got = P.landscape(sweeps)
            assert np.all(got > 0) and np.allclose(got, u, rtol=1e-12, atol=0), (spmv, sweeps, float(np.abs(got / u - 1).max()))
    P.set_option("spmv", _lib.OPTION_AUTO)
    for bad in (-1, 17):
        with pytest.raises(AssertionError):          # BAD_ARG, like the reference's asserts
            P.landscape(bad)
    x = w["x0"].copy()
    for it in range(2):
        P.set_x(x)
        P.set_option("start_land", _lib.OPTION_AUTO)
        lam_on, v_on, _ = P.fiedler(tol=1e-8); st_on = int(P.stats.lanczos_steps); assert P.stats.residual < 1e-8
        P.set_option("start_land", 0)
        lam_off, v_off, _ = P.fiedler(tol=1e-8); st_off = int(P.stats.lanczos_steps); assert P.stats.residual < 1e-8
        assert abs(lam_on - lam_off) <= LAM_RTOL * lam_off and np.abs(sign_align(v_on, v_off) - v_off).max() <= 2e-6
        assert st_on < 0.95 * st_off, (it, st_on, st_off)         # measured 141 / 155 and 221 / 285
        # (c) the caller's own guess: same bits whether the option is on or off, and the same bits as the unweighted cold start
        lam_g0, v_g0, _ = P.fiedler(tol=1e-8, x0=z)
        P.set_option("start_land", _lib.OPTION_AUTO)
        lam_g1, v_g1, _ = P.fiedler(tol=1e-8, x0=z)
        assert lam_g0 == lam_g1 == lam_off and np.array_equal(v_g0, v_g1) and np.array_equal(v_g0, v_off)
        # (d) warm start from the vector that solve left: option on or off alike
        lam_w1, v_w1, _ = P.fiedler(tol=1e-8, warm_start=True); st_w1 = int(P.stats.lanczos_steps)
        P.set_option("start_land", 0)
        P.fiedler(tol=1e-8, x0=z)
        lam_w0, v_w0, _ = P.fiedler(tol=1e-8, warm_start=True); st_w0 = int(P.stats.lanczos_steps)
        P.set_option("start_land", _lib.OPTION_AUTO)
        assert st_w1 == st_w0 and lam_w1 == lam_w0 and np.array_equal(v_w1, v_w0) and st_w1 < st_on, (st_w1, st_w0, st_on)
        P.set_start(z)
        g = P.gradient()
        s = np.zeros(m); s[np.argpartition(g, -k)[-k:]] = 1.0
        x_prev = x
        x = x + 2.0 / (it + 2) * (s - x)
    # (e) a warm start that turns out no better than a random vector (the localised vector has moved: overlap < 2 / sqrt(n)) sends the next
    # warm requests to the weighted cold start: same bits as asking for a cold solve
    P.set_x(x_prev); P.fiedler(tol=1e-8)
    P.set_x(x)
    lam_c, v_c, _ = P.fiedler(tol=1e-8); st_c = int(P.stats.lanczos_steps)
    P.set_x(x_prev); P.fiedler(tol=1e-8)
    P.set_x(x)
    lam_p, _, _ = P.fiedler(tol=1e-8, warm_start=True); st_p = int(P.stats.lanczos_steps)      # the probe: really warm-started
    assert abs(lam_p - lam_c) <= LAM_RTOL * lam_c and st_p > st_c, (st_p, st_c)
    lam_f, v_f, _ = P.fiedler(tol=1e-8, warm_start=True); st_f = int(P.stats.lanczos_steps)      # warm asked for, cold (weighted) run
    assert lam_f == lam_c and st_f == st_c and np.array_equal(v_f, v_c), (st_f, st_c)
    P.close()


def test_batched_two_handles_match_sequential():
    """BASELINE.json configs[4] batched mode (bench.py --config c5): city10000 and sphere2500 as two handles
    driven by two host threads on one GPU give bit-identical trajectories to running them one after the other."""
    import threading
    gs = [load_golden("g2o_city10000"), load_golden("g2o_sphere2500")]

    def run(g, out, i):
        P = problem_of(g)
        P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
        P.set_x(g["x_init"])
        fs = []
        for it in range(6):
            f, dual, gn = P.fw_step(int(g["k"]), it)
            fs.append(f)
            P.fw_commit()
        out[i] = (np.array(fs), P.get_x())
        P.close()
    seq, par = [None, None], [None, None]
    for i, g in enumerate(gs):
        run(g, seq, i)
    th = [threading.Thread(target=run, args=(g, par, i)) for i, g in enumerate(gs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i, g in enumerate(gs):
        assert np.array_equal(seq[i][0], par[i][0]) and np.array_equal(seq[i][1], par[i][1])
        assert np.allclose(seq[i][0][:2], g["f_traj"][:2], rtol=1e-6)      # (city10000 forks after iteration 1)


# --------------------------------------------------------------------------------------------
def test_edge_cases_duplicates_selfloops_reversed():
    rng = np.random.default_rng(5)
    n = 40
    fi = np.arange(n - 1); fj = fi + 1; fw = rng.random(n - 1) + 0.5
    # duplicate fixed edge, reversed orientation, self loop
    fi = np.concatenate([fi, [3, 10, 7]]); fj = np.concatenate([fj, [4, 9, 7]]); fw = np.concatenate([fw, [0.25, 0.5, 9.0]])
    ci = rng.integers(0, n, 120); cj = rng.integers(0, n, 120); cw = rng.random(120) + 0.1
    ci[0], cj[0] = 3, 4          # candidate on a fixed pair
    ci[1], cj[1] = 20, 30; ci[2], cj[2] = 30, 20   # duplicate candidate pair, reversed
    ci[3] = cj[3] = 11           # candidate self loop
    x = rng.random(120) * (rng.random(120) < 0.6)
    P = _lib.Problem(n, fi, fj, fw, ci, cj, cw)
    P.set_x(x)
    indptr, indices, data = P.laplacian_csr()
    L = sp.csr_matrix((data, indices, indptr), shape=(n, n)); L.sum_duplicates(); L.sort_indices()
    mo = oracle.MacOracle(fi, fj, fw, ci, cj, cw, n)
    Lr = mo.laplacian(x).tocsr(); Lr.sum_duplicates(); Lr.eliminate_zeros(); Lr.sort_indices()
    L.eliminate_zeros()
    assert np.abs((L - Lr)).max() <= 1e-14 * abs(Lr).max()
    lam, v, _ = P.fiedler(x0=reference_start_block(n)[:, 0].copy())
    f, gr = mo.problem(x)
    assert abs(lam - f) <= LAM_RTOL * f
    grad = P.gradient()
    assert np.array_equal(grad, oracle.supergradient(v, ci, cj, cw))
    assert grad[3] == 0.0
    P.close()


def test_errors_like_reference():
    g = load_golden("petersen_solve_k3")
    fixed, cand = edges_of(g, "f"), edges_of(g, "c")
    with pytest.raises(AssertionError):
        MAC(fixed[:3], cand[:2], 10)                       # fewer than n-1 edges (mac.py:47)
    mac = MAC(fixed, cand, 10)
    with pytest.raises(AssertionError):
        mac.solve(2, np.zeros(3))                          # len(x_init) != m (mac.py:183)
    with pytest.raises((AssertionError, TypeError)):
        mac.solve(2, None)                                 # x_init=None unsupported (mac.py:182)
    with pytest.raises(AssertionError):
        _lib.Problem(4, [0, 1, 2], [1, 2, 9], [1., 1., 1.], [0], [3], [1.0])   # node id out of range
    # disconnected K3 u K3: the reference raises (SuperLU singular); here a typed error
    e = [(0, 1), (0, 2), (1, 2), (3, 4), (3, 5), (4, 5)]
    L = weight_graph_lap_from_edge_list([Edge(a, b, 1.0) for a, b in e], 6)
    with pytest.raises(_lib.Disconnected):
        find_fiedler_pair(L)
    with pytest.raises(AssertionError):
        find_fiedler_pair(weight_graph_lap_from_edge_list(fixed + cand, 10), X=np.zeros((10, 2)))


def test_k_ge_m_shortcut_and_k0():
    g = load_golden("petersen_solve_k3")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), 10)
    r, w, val = mac.solve(6, np.zeros(6))
    assert np.all(r == 1) and abs(val - 2.0) < 1e-7
    r0, w0, u0 = mac.solve(0, np.zeros(6), max_iters=3)
    assert r0.sum() == 0 and np.all(w0 == 0)


def test_naive_greedy_init_and_laplacian_accessor():
    g = load_golden("g2o_intel")
    cand = edges_of(g, "c")
    x0 = NaiveGreedy(cand).subset(int(g["k"]))
    assert x0.sum() == int(g["k"])
    mac = MAC(edges_of(g, "f"), cand, int(g["n"]))
    L = mac.laplacian(x0)
    Lr = oracle_of(g).laplacian(x0)
    assert abs(L - Lr).max() <= 1e-12 * abs(Lr).max()
    assert mac.weights.shape == (len(cand),) and mac.edge_list.shape == (len(cand), 2)
    assert abs(mac.L_fixed - oracle.laplacian_from_edges(g["fi"], g["fj"], g["fw"], int(g["n"]))).max() == 0


# --------------------------------------------------------------------------------------------
def make_er(n, p, seed):
    import networkx as nx
    G = nx.fast_gnp_random_graph(n, p, seed=seed)
    e = np.array([(min(a, b), max(a, b)) for a, b in G.edges() if abs(a - b) != 1], dtype=np.int32)
    return e[:, 0].copy(), e[:, 1].copy()


def test_full_size_config2_properties():
    """BASELINE.json configs[1]: ER N=10k p=0.01, chain fixed, K=10%.  lambda_2 against the
    reference value captured in tests/golden/er10k_x0.npz; size-independent identities."""
    g = load_golden("er10k_x0")
    n = 10000
    ci, cj = make_er(n, 0.01, 0)
    m = len(ci)
    assert m == int(g["m"]) and np.array_equal(ci[:512], g["ci_head"]) and np.array_equal(cj[:512], g["cj_head"])
    fi = np.arange(n - 1, dtype=np.int32); fj = fi + 1
    P = _lib.Problem(n, fi, fj, np.ones(n - 1), ci, cj, np.ones(m))
    x0 = np.zeros(m); x0[g["x0_idx"]] = 1.0
    P.set_x(x0)
    lam, v, _ = P.fiedler(tol=1e-8, x0=reference_start_block(n)[:, 0].copy())
    assert abs(lam - float(g["lam"])) <= LAM_RTOL * float(g["lam"])
    assert np.abs(sign_align(v, g["v"]) - g["v"]).max() <= 2e-6
    grad = P.gradient()
    assert abs(grad.sum() - float(g["grad_sum"])) <= 1e-5 * float(g["grad_sum"])
    assert np.abs(grad[:512] - g["grad_head"]).max() <= 1e-5 * np.abs(g["grad_head"]).max()
    # identity lambda_2 = x.g + v^T L_fixed v ; residual rule on an independent SpMV
    Lf = oracle.laplacian_from_edges(fi, fj, np.ones(n - 1), n)
    assert abs(x0 @ grad + v @ (Lf @ v) - lam) <= 1e-9 * lam
    L = oracle.mac_laplacian(Lf, ci.astype(np.int64), cj.astype(np.int64), np.ones(m), x0, n)
    assert np.abs(L @ v - lam * v).sum() / abs(L).sum(axis=1).max() < 1e-8
    # FW iterations: dual bound >= f, x stays feasible, support grows by <= K, bit-exact update
    k = m // 10
    u = np.inf
    x_prev = x0
    for it in range(4):
        f, dual, gn = P.fw_step(k, it)
        u = min(u, dual)
        assert u >= f - 1e-9 * abs(f)
        grad = P.gradient()
        s = P.lp_topk(k)
        assert s.sum() == k
        P.fw_commit()
        x = P.get_x()
        assert np.array_equal(x, x_prev + (2.0 / (it + 2.0)) * (s - x_prev))
        assert x.min() >= 0 and x.max() <= 1 and x.sum() <= k * (1 + 1e-12)
        assert abs(dual - (f + grad @ (s - x_prev))) <= 1e-9 * abs(dual)
        x_prev = x
    P.close()


def test_rccl_path_single_rank():
    """machip_comm_init + the in-place ncclAllGather of the gradient (world size 1 is all a 1-GPU box
    allows; the collective code path, padding and stream ordering are the ones N ranks run)."""
    from mac_amd.dist import shard_bounds
    g = load_golden("er2000_solve")
    k = int(g["k"])
    P0 = problem_of(g); P1 = problem_of(g)
    for P in (P0, P1):
        P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
        P.set_x(g["x_init"])
    P1.comm_init(0, 1, _lib.comm_unique_id())
    assert shard_bounds(P1.m, 0, 1) == (0, P1.m, P1.m)
    for it in range(3):
        a = P0.fw_step(k, it); b = P1.fw_step(k, it)
        assert a == b
        assert np.array_equal(P0.gradient(), P1.gradient())
        P0.fw_commit(); P1.fw_commit()
    assert np.array_equal(P0.get_x(), P1.get_x())
    P0.close(); P1.close()


def test_mixed_precision_matches_fp64_results():
    """machip_set_precision(1) (BASELINE.json configs[4]: fp32 Krylov iterate + fp64 Rayleigh / residual refinement):
    the returned pair obeys the same stop rule and the same bounds as the fp64 mode -- lambda_2 1e-8 against the
    REFERENCE goldens, vector 2e-6, gradient 1e-5 of max|g| -- on the two graphs of configs[4], on intel and on a
    weighted ER graph; part of the steps must really have run in fp32."""
    for nm in ["g2o_sphere2500", "g2o_city10000", "g2o_intel", "er2000_x0"]:
        g = load_golden(nm)
        P = problem_of(g)
        n = int(g["n"])
        x = g["x_init"] if "x_init" in g else g["x"]
        lam_ref = float(g["lam_init"]) if "lam_init" in g else float(g["lam"])
        v_ref = g["v_init"] if "v_init" in g else g["v"]
        g_ref = g["grad_init"] if "grad_init" in g else g["grad"]
        P.set_x(x)
        lam64, v64, _ = P.fiedler(x0=reference_start_block(n)[:, 0].copy())
        st64 = P.stats.asdict()
        P.set_precision(1)
        lam, v, _ = P.fiedler(x0=reference_start_block(n)[:, 0].copy())
        st = P.stats.asdict()
        assert st["steps_lowp"] > 0 and st["steps_lowp"] < st["lanczos_steps"], (nm, st)
        assert st64["steps_lowp"] == 0
        assert st["residual"] < 1e-8
        assert abs(lam - lam_ref) <= LAM_RTOL * lam_ref, (nm, lam, lam_ref)
        assert abs(lam - lam64) <= LAM_RTOL * lam_ref
        assert np.abs(sign_align(v, v_ref) - v_ref).max() <= 2e-6
        grad = P.gradient()
        assert np.abs(grad - g_ref).max() <= 1e-5 * np.abs(g_ref).max()
        assert np.array_equal(grad, oracle.supergradient(v, g["ci"].astype(np.int64), g["cj"].astype(np.int64), g["cw"]))
        P.close()


def test_mixed_precision_solve_trajectory_on_configs4_graphs():
    """Whole MAC.solve in the mixed mode on sphere2500 (and city10000 up to its fork): same trajectory as the
    reference to the fp64 tolerances."""
    g = load_golden("g2o_sphere2500")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), precision=1)
    rounded, w, u = mac.solve(int(g["k"]), g["x_init"], max_iters=20)
    ft = np.array([t[0] for t in mac.trace])
    assert np.allclose(ft, g["f_traj"][:len(ft)], rtol=1e-6)
    assert np.array_equal([t[3] for t in mac.trace], g["supp"][:len(ft)])
    assert abs(u - g["upper"]) <= 1e-5 * abs(g["upper"])
    g = load_golden("g2o_city10000")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), precision=1)
    rounded, w, u = mac.solve(int(g["k"]), g["x_init"], max_iters=2)
    assert np.allclose([t[0] for t in mac.trace], g["f_traj"][:2], rtol=1e-8)


@pytest.mark.parametrize("R", [2, 3, 8])
def test_in_process_ranks_match_single_rank(R):
    """Multi-GPU path as a PRODUCT test on one GPU: R handles of the same problem joined by
    machip_comm_init_local (ranks 0..R-1, one host thread each; candidate shards + in-place all-gather through the
    same compute_gradient call site and the same machip_shard_plan arithmetic as the RCCL path) must reproduce the
    single-rank Frank-Wolfe run bit for bit -- m = 5 971 is not a multiple of 8, so the padded tail is exercised."""
    import threading
    g = load_golden("er2000_solve")
    n, k, iters = int(g["n"]), int(g["k"]), 5
    start = reference_start_block(n)[:, 0].copy()

    def drive(P, out, i):
        try:
            P.set_start(start)
            P.set_x(g["x_init"])
            fs = []
            for it in range(iters):
                f, dual, gn = P.fw_step(k, it)
                fs.append((f, dual, gn))
                P.fw_commit()
            gr = P.gradient()          # collective again: every rank ends with the full vector
            out[i] = (np.array(fs), P.get_x(), gr)
        except Exception as exc:       # a failing rank must not leave its peers in the barrier
            out[i] = exc
            P.close()
    single = [None]
    P0 = problem_of(g)
    drive(P0, single, 0)
    P0.close()
    assert not isinstance(single[0], Exception), single[0]
    Ps = [problem_of(g) for _ in range(R)]
    _lib.comm_init_local(Ps)
    assert _lib.load().machip_comm_mode(Ps[0]._h) == 3      # candidate shard AND row-partitioned eigen-solve
    m = len(g["cw"])
    cover = []
    for r in range(R):
        lo, hi, shard = _lib.shard_plan(m, R, r)
        cover.append((lo, hi))
        assert shard * R >= m
    assert cover[0][0] == 0 and cover[-1][1] == m and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    out = [None] * R
    th = [threading.Thread(target=drive, args=(Ps[r], out, r)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    for r in range(R):
        assert not isinstance(out[r], Exception), out[r]
        for a, b in zip(out[r], single[0]):
            assert np.array_equal(a, b), f"rank {r} of {R} differs from the single-rank run"
    assert np.allclose(single[0][0][:, 0], g["f_traj"][:iters], rtol=1e-6)
    for P in Ps:
        P.close()


@pytest.mark.parametrize("cfg,R,iters", [("c2", 4, 3), ("c4", 2, 2), ("c4", 8, 2)])
def test_row_partitioned_eigensolve_is_bit_identical_at_bench_sizes(cfg, R, iters):
    """The row-partitioned Lanczos step of the in-process communicator (rank r launches its share of every step's
    workgroups on its own copy of L(x) and of the operand, writes records and partial sums into every rank's copy, basis
    sharded by rows) on the BENCH workloads: several row tiles per workgroup, deferred-barrier launch shapes, hundreds
    of steps per solve, graph-captured multi-stream chunks.  f / dual bound / ||g|| / x on every rank are bit-identical
    to a single handle running the same one-kernel step (option panel = 0: the column-panel form is not sharded)."""
    import threading
    import bench
    w = bench.make_workload(cfg)
    n, k = w["n"], w["k"]
    start = reference_start_block(n)[:, 0].copy()
    with _lib.default_options(panel=0):
        def mk():
            return _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])

        def drive(P, out, i):
            try:
                P.set_start(start); P.set_x(w["x0"])
                fs = []
                for it in range(iters):
                    fs.append(P.fw_step(k, it)); P.fw_commit()
                out[i] = (np.array(fs), P.get_x(), int(P.stats.lanczos_steps))
            except Exception as exc:
                out[i] = exc
                P.close()
        single = [None]
        P0 = mk(); drive(P0, single, 0); P0.close()
        assert not isinstance(single[0], Exception), single[0]
        Ps = [mk() for _ in range(R)]
        _lib.comm_init_local(Ps)
        assert _lib.load().machip_comm_mode(Ps[0]._h) == 3
        out = [None] * R
        th = [threading.Thread(target=drive, args=(Ps[r], out, r)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        for r in range(R):
            assert not isinstance(out[r], Exception), out[r]
            assert np.array_equal(out[r][0], single[0][0]) and np.array_equal(out[r][1], single[0][1]), f"rank {r} of {R}"
            assert out[r][2] == single[0][2] > 50
        for P in Ps:
            P.close()


def test_in_process_group_with_replicated_eigensolve_still_works():
    """Option shard_eig = 0: round 2's mode (every rank runs the whole eigen-solve itself) stays available and agrees."""
    import threading
    g = load_golden("er2000_solve")
    k = int(g["k"])
    with _lib.default_options(shard_eig=0):
        Ps = [problem_of(g) for _ in range(2)]
        _lib.comm_init_local(Ps)
        assert _lib.load().machip_comm_mode(Ps[0]._h) == 2
    out = [None, None]

    def drive(i):
        P = Ps[i]
        P.set_start(reference_start_block(int(g["n"]))[:, 0].copy()); P.set_x(g["x_init"])
        out[i] = [P.fw_step(k, it) + (P.fw_commit(),) for it in range(3)]
    th = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert out[0] == out[1] and np.allclose([o[0] for o in out[0]], g["f_traj"][:3], rtol=1e-6)
    for P in Ps:
        P.close()


def test_in_process_group_releases_peers_when_a_rank_dies():
    import threading
    g = load_golden("er300_solve")
    Ps = [problem_of(g) for _ in range(2)]
    _lib.comm_init_local(Ps)
    res = {}

    def lone():
        try:
            Ps[0].set_x(g["x_init"])
            Ps[0].fw_step(int(g["k"]), 0)
            res["ok"] = True
        except _lib.MachipError as exc:
            res["err"] = str(exc)
    t = threading.Thread(target=lone)
    t.start()
    import time
    time.sleep(1.0)
    Ps[1].close()                  # the peer never joins the collective
    t.join(timeout=60)
    assert not t.is_alive() and "err" in res, res
    Ps[0].close()


def test_full_size_config4_properties():
    """BASELINE.json configs[3] / north_star target size: ER N=100k, ~2M candidates, K=10%.  The
    reference cannot finish one solve here (SuperLU fill-in); lambda_2 is checked against
    the committed fixture tests/golden/er100k_x0.npz (scipy.sparse.linalg.eigsh on the reference's own
    MAC.laplacian(x0), generator: make_golden.py er100k_x0) and through size-independent identities evaluated
    with an independent (SciPy) SpMV."""
    import bench
    w = bench.make_workload("c4")
    n, m, k = w["n"], len(w["cw"]), w["k"]
    assert n == 100000 and m == 2001737 and k == 200173
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_x(w["x0"])
    lam, v, _ = P.fiedler(tol=1e-8, x0=reference_start_block(n)[:, 0].copy())
    gx = load_golden("er100k_x0")
    assert int(gx["m"]) == m and np.array_equal(np.nonzero(w["x0"])[0][:512], gx["x0_idx_head"])
    assert abs(lam - float(gx["lam"])) <= 1e-8 * float(gx["lam"])
    assert np.abs(sign_align(v[::997], gx["v_stride"]) - gx["v_stride"]).max() <= 2e-6
    assert P.stats.residual < 1e-8 and abs(np.linalg.norm(v) - 1) < 1e-12 and abs(v.sum()) < 1e-9
    Lf = oracle.laplacian_from_edges(w["fi"], w["fj"], w["fw"], n)
    L = oracle.mac_laplacian(Lf, w["ci"].astype(np.int64), w["cj"].astype(np.int64), w["cw"], w["x0"], n)
    assert np.abs(L @ v - lam * v).sum() / abs(L).sum(axis=1).max() < 1e-8          # nx:246 on scipy's SpMV
    grad = P.gradient()
    assert np.array_equal(grad, oracle.supergradient(v, w["ci"], w["cj"], w["cw"]))  # bit-exact
    assert abs(w["x0"] @ grad + v @ (Lf @ v) - lam) <= 1e-9 * lam
    s = P.lp_topk(k)
    assert s.sum() == k and np.array_equal(s, oracle.solve_subset_box_lp(grad, k))
    u = np.inf
    for it in range(3):
        f, dual, gn = P.fw_step(k, it)
        u = min(u, dual)
        assert u >= f and P.stats.residual < 1e-8
        P.fw_commit()
    x = P.get_x()
    assert x.min() >= 0 and x.max() <= 1 and x.sum() <= k * (1 + 1e-12)
    P.close()


def test_x_block_columns_are_what_the_header_says():
    """find_fiedler_pair's X block (fiedler.py:44): column 0 is the converged Fiedler vector; columns 1..q-1 are the next
    Ritz vectors of the LAST Krylov sequence -- always orthonormal, orthogonal to 1, Rayleigh quotients >= lambda_2 (include/machip.h says so
    and no more).  Compared with dense eigh on the golden graphs: (a) er2000_x0, one sequence of ~250 steps: each column's angle to the
    eigenvector of the same rank is below 1e-3 (measured 5e-7, 5e-6, 1.5e-4) and its Rayleigh quotient equals lambda_3..5 to 1e-6;
    (b) er300_x0 converges within 33 steps: the other Ritz vectors of so short a sequence are no eigenvectors (Rayleigh quotients ~5 against
    lambda_3 = 0.78) -- only the guaranteed properties hold, which is what this test pins for it."""
    for nm, accurate in (("er2000_x0", True), ("er300_x0", False)):
        g = load_golden(nm)
        n = int(g["n"])
        P = problem_of(g)
        P.set_x(g["x"])
        ip, ix, da = P.laplacian_csr()
        L = sp.csr_matrix((da, ix, ip), shape=(n, n))
        ev, V = np.linalg.eigh(L.toarray())
        lam, v, X = P.fiedler(tol=1e-8, x0=reference_start_block(n)[:, 0].copy(), q=4)
        assert np.abs(X.T @ X - np.eye(4)).max() < 1e-12 and np.abs(X.sum(axis=0)).max() < 1e-10 and np.array_equal(X[:, 0], v)
        rho = np.array([X[:, c] @ (L @ X[:, c]) for c in range(4)])
        assert abs(rho[0] - ev[1]) <= LAM_RTOL * ev[1] and np.all(rho[1:] >= ev[1] * (1 - 1e-10)), (nm, rho, ev[1:6])
        if accurate:
            for c in range(1, 4):
                sin_c = np.sqrt(max(0.0, 1.0 - float(V[:, c + 1] @ X[:, c]) ** 2))
                assert sin_c < 1e-3 and abs(rho[c] - ev[c + 1]) <= 1e-6 * ev[c + 1], (nm, c, sin_c, rho[c], ev[c + 1])
        P.close()


def test_landscape_start_keeps_a_floor_under_every_entry():
    """Advisor finding on round 5 (kernels.h k_land_weight): (u / max u)^128 underflows to 0 almost everywhere, so the weighted start sits on
    the lowest-degree vertices alone -- and where the Fiedler vector VANISHES there, Lanczos converges to lambda_3 first and the residual test
    passes it (a true eigenpair).  The graph: two identical random clusters A, B joined through one middle vertex m (A_0 - m - B_0, weak
    edges), and a pendant vertex p hung on m: by the mirror symmetry v_2 is antisymmetric and EXACTLY zero on m and p, while p (degree 0.5
    against ~25) is the landscape's one peak; lambda_2 = 6.7e-5 (the bridge), lambda_3 = 2.0e-2 (the pendant's own mode).  With the floor
    (default 1e-3 of every entry's draw) the solve returns lambda_2; without it (start_floor_e6 = 0) it returns a true eigenvalue, but the
    wrong one whenever nothing else perturbs the start (NumPy emulation: lambda_3 for floors 0 and 1e-6, lambda_2 from 1e-4 on)."""
    rng = np.random.default_rng(5)
    nc = 300
    A = np.triu(rng.random((nc, nc)) < 0.08, 1)
    ai, aj = np.nonzero(A)
    m_, p_ = 2 * nc, 2 * nc + 1
    n = 2 * nc + 2
    fi = np.concatenate([ai, ai + nc, [0, nc, m_]]).astype(np.int32)
    fj = np.concatenate([aj, aj + nc, [m_, m_, p_]]).astype(np.int32)
    fw = np.concatenate([np.ones(2 * len(ai)), [0.02, 0.02, 0.5]])
    W = np.zeros((n, n)); W[fi, fj] = fw; W = W + W.T
    L = np.diag(W.sum(1)) - W
    ev, V = np.linalg.eigh(L)
    assert ev[1] > 1e-6 and ev[2] > 100 * ev[1] and np.abs(V[[m_, p_], 1]).max() < 1e-9       # connected; v_2 vanishes on m and p
    P = _lib.Problem(n, fi, fj, fw, np.array([1], dtype=np.int32), np.array([7], dtype=np.int32), np.array([1.0]))
    P.set_start(reference_start_block(n)[:, 0].copy())
    P.set_x(np.zeros(1))
    lam, v, _ = P.fiedler(tol=1e-10)
    assert P.solve_mode()[0] in (1, 3) and abs(lam - ev[1]) <= 1e-6 * ev[1] and P.stats.residual < 1e-10, (lam, ev[:4], P.solve_mode())
    u = P.landscape(3)
    assert int(np.argmax(u)) == p_                       # the pendant is the peak the weighting moves the start to
    P.set_option("start_floor_e6", 0)
    lam0, _, _ = P.fiedler(tol=1e-10)
    assert min(abs(lam0 - ev[1]) / ev[1], abs(lam0 - ev[2]) / ev[2]) <= 1e-6, (lam0, ev[:4])      # (a true eigenvalue either way; lambda_3 on every run so far)
    P.close()


@pytest.mark.parametrize("opts", [
    {"spmv": 1, "tpr": 4}, {"spmv": 1, "tpr": 16},
    {"g": 4, "block": 256, "unroll": 2}, {"g": 16, "block": 1024},
    {"g": 64, "block": 512}, {"g": 32, "maxgrid": 64},
    {"graph": 0, "chunk": 6, "chunk_near": 2}, {"classic_n": 100000},
    {"stream": 0}, {"stream": 2}, {"stream": 0, "graph": 0, "chunk": 6, "chunk_near": 2}, {"stream": 1, "stream_look": 2, "stream_far": 34},
    {"vcap": 80}, {"asm_g": 8}, {"asm_g": 32, "g": 8, "unroll": 1},
    # column-panel step (panel.h) forced onto small graphs: several panels / row blocks, ragged last panel and tile
    {"panel": 1}, {"panel": 1, "panel_np": 3}, {"panel": 1, "panel_np": 1, "panel_nb": 2},
    # ... in record form (round 3: 16-byte records, k_pan_mul + k_pan_fin; the lines around this one run round 6's shifted recurrence, panel_u.h)
    {"panel": 1, "panel_np": 3, "panel_u": 0}, {"panel": 1, "panel_np": 5, "panel_nb": 7, "panel_b2": 512, "panel_g2": 3, "panel_u": 0},
    {"panel": 1, "panel_np": 7, "graph": 0, "chunk": 6, "panel_b2": 1024, "panel_u": 0}, {"panel": 1, "panel_u": 0, "stream": 0},
    # ... the shifted recurrence where the host does not follow the records (no drift monitor there: the plan falls back to records), captured chunks
    {"panel": 1, "stream": 0}, {"panel": 1, "panel_np": 3, "graph": 1, "chunk": 8},
    {"panel": 1, "panel_np": 5, "panel_nb": 7, "panel_b2": 512, "panel_g2": 3},
    {"panel": 1, "panel_np": 7, "graph": 0, "chunk": 6, "panel_b2": 1024},
    # ... every step walking the tiles forwards (default: odd steps backwards, for the L2's sake)
    {"panel": 1, "panel_rev": 0}, {"panel": 1, "panel_np": 3, "panel_rev": 0, "graph": 1, "chunk": 8},
    # ... several row blocks per workgroup, the panel loaded once (k_pan_mul_multi; round 4): even split, ragged split (cells that have no row block)
    {"panel": 1, "panel_np": 3, "panel_nb": 6, "panel_cells": 2},
    {"panel": 1, "panel_np": 5, "panel_nb": 7, "panel_cells": 3, "graph": 0, "chunk": 6},
    # ... the round-3 layout (tridiagonal band inside the tiles)
    {"panel": 1, "panel_np": 3, "panel_band": 0},
], ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_solver_variants_agree(opts):
    """Every launch shape / row mapping of the fused step kernel, eager vs graph launches, odd chunk
    sizes, the classic two-kernel form and a basis so small that it forces restarts must all give the
    reference's lambda_2.  The variants are entries of the handle's option table (machip_set_option; creation-time ones
    -- asm_g, vcap -- as process defaults around the handle's creation): no environment, no subprocess.  (Round 4's
    one-launch panel step and diagonally preconditioned LOBPCG forms are compiled with -DMACHIP_EXPERIMENTS only.)"""
    for nm in ["er2000_xfrac", "er300_x0"]:
        g = load_golden(nm)
        with _lib.default_options(**opts):
            P = _lib.Problem(int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])
        for k_, v_ in opts.items():
            assert P.get_option(k_) == v_
        P.set_x(g["x"])
        lam, v, _ = P.fiedler(tol=1e-8, x0=reference_start_block(int(g["n"]))[:, 0].copy())
        rel = abs(lam - float(g["lam"])) / float(g["lam"])
        dv = float(np.abs(sign_align(v, g["v"]) - g["v"]).max())
        assert rel <= LAM_RTOL and dv <= 2e-6 and P.stats.residual < 1e-8, (nm, rel, dv, P.stats.residual)
        P.close()


def _stream_cases():
    rng = np.random.default_rng(77)
    # (a) gather step, ER-like (golden er2000); (b) padded fixed-width step on a chain + closures graph beyond the single-workgroup sizes;
    # (c) column-panel step forced onto er2000
    g = load_golden("er2000_xfrac")
    yield "er2000", (int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"]), g["x"], {}
    n = 6001
    fi = np.arange(n - 1, dtype=np.int32)
    a = rng.integers(0, n, 1500); b = rng.integers(0, n, 1500)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    yield "chain6001", (n, fi, fi + 1, rng.uniform(50.0, 300.0, n - 1), ci, cj, rng.uniform(50.0, 150.0, len(ci))), np.ones(len(ci)), {}
    yield "er2000_panel", (int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"]), g["x"], {"panel": 1, "panel_np": 3}


@pytest.mark.parametrize("case", list(_stream_cases()), ids=lambda c: c[0])
def test_streamed_records_end_a_solve_at_the_same_step_whatever_feeds_the_queue(case):
    """Round 5, streamed records (solver.h): every Lanczos step hands its (alpha, l1, beta) to the host, which analyses the
    tridiagonal at a sequence of points that depends on the records alone and ends the solve at the first point whose residual
    estimate is below the trigger.  WHAT feeds the queue -- the timing-driven feeder of the unpartitioned solve (stream = 1), the
    chunk-at-a-time feeder of the row-partitioned solves (stream = 2), an impatient feeder (look 2, no whole chunks ahead) -- must
    not change the final step, lambda_2 or one bit of the vector, run after run; the steps launched beyond the final point are few;
    and the chunk-granular schedule of rounds 1-4 (stream = 0) agrees to the solver tolerance."""
    name, args, x, opts = case
    with _lib.default_options(**opts):
        P = _lib.Problem(*args)
    P.set_x(x)
    P.set_solver(1)
    runs = {}
    for tag, so in (("s1", {"stream": 1}), ("s1b", {"stream": 1}), ("s2", {"stream": 2}), ("s1_impatient", {"stream": 1, "stream_look": 2, "stream_far": 34}),
                    ("s1_eager", {"stream": 1, "graph": 0}), ("s0", {"stream": 0})):
        for k_, v_ in {"stream_look": _lib.OPTION_AUTO, "stream_far": _lib.OPTION_AUTO, "graph": _lib.OPTION_AUTO, **so}.items():
            P.set_option(k_, v_)
        lam, v, _ = P.fiedler(tol=1e-8)
        st = P.stats
        assert st.residual < 1e-8, (name, tag, st.residual)
        runs[tag] = (lam, v.copy(), int(st.lanczos_steps), int(st.steps_timed))
    ref = runs["s1"]
    for tag in ("s1b", "s2", "s1_impatient", "s1_eager"):
        r = runs[tag]
        assert r[0] == ref[0] and r[2] == ref[2] and np.array_equal(r[1], ref[1]), (name, tag, r[0], ref[0], r[2], ref[2])
    # launched beyond the final analysis point: a handful on the bench configs (the forecast errs by a step or two; chunks are even); on a
    # solve this short (er2000: 106 steps) the whole 32-step chunks of the far regime reach step 128 on the forecast made at step 64, and a
    # slow host (the AddressSanitizer build: 160) lets the GPU run one chunk further -- timing decides this number, nothing reads those steps
    assert 0 <= ref[3] - ref[2] <= 64, (name, ref[2], ref[3])
    assert runs["s0"][3] >= runs["s0"][2]
    assert abs(runs["s0"][0] - ref[0]) <= 1e-8 * ref[0] and np.abs(sign_align(runs["s0"][1], ref[1]) - ref[1]).max() <= 2e-6
    P.close()


@pytest.mark.parametrize("hub", [0, 9, 20])
def test_padded_fixed_width_step_is_bit_identical_to_the_csr_step(hub):
    """Pose graphs beyond the single-workgroup kernel (city10000-like: 3-13 entries per row) step on a padded fixed-width
    copy of L(x) (k_ell_build, 8 or 16 slots per row; option `ell`): same products in the same order plus zeros, so lambda_2
    and the vector must equal the CSR step's bit for bit.  hub = 9: a 12-entry row (16-slot form); hub = 20: longer than 16,
    the padded form must not be chosen (and the result is the same anyway)."""
    rng = np.random.default_rng(5 + hub)
    n = 6001
    fi = np.arange(n - 1, dtype=np.int32)
    fw = rng.uniform(50.0, 300.0, n - 1)
    a = rng.integers(0, n, 1500); b = rng.integers(0, n, 1500)
    if hub:
        a = np.r_[a, np.full(hub, 1234)]; b = np.r_[b, rng.choice(np.arange(2000, n), hub, replace=False)]
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = rng.uniform(50.0, 150.0, len(ci))
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(np.ones(len(ci)))
    P.set_solver(1)
    res = {}
    for ell in ("0", "1"):
        P.set_option("ell", int(ell))
        lam, v, _ = P.fiedler()
        assert P.stats.residual < 1e-8
        res[ell] = (lam, v.copy(), int(P.stats.lanczos_steps))
    assert res["0"][0] == res["1"][0] and res["0"][2] == res["1"][2] and np.array_equal(res["0"][1], res["1"][1])
    P.close()


def test_panel_step_multi_round_windows_hub_rows_and_odd_sizes():
    """Corners of the column-panel step (panel.h) the bench matrices do not reach: (a) dense rows -- a worker wave's tiles
    hold more chunks than its registers (kPanCH = 20), so the multi-round path runs; (b) n not a multiple of 64 nor of the
    panel width, last row block mostly empty; (c) a hub row right at the 127-entry limit, one beyond it whose entries spread over the
    panels (every (row, panel) count stays within 127: the panel step serves it) and one whose 200 closures sit inside ONE panel
    (k_pan_rows raises its flag and the solver falls back to the gather step: the forced-panel solve is then bit-identical to the
    gather solve).  lambda_2 of the panel step equals the gather step's to 1e-12 and SciPy's to 1e-8; the vectors agree."""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(11)
    for n, deg, hub, packed in ((70001, 64, 0, False), (66003, 20, 126, False), (66003, 20, 140, False), (66003, 20, 200, True)):
        m0 = n * deg // 2
        a = rng.integers(0, n, m0); b = rng.integers(0, n, m0)
        keep = (a != b) & (np.abs(a - b) != 1)
        lo, hi = np.minimum(a[keep], b[keep]), np.maximum(a[keep], b[keep])
        if hub:          # a hub: node 7 joined to `hub` further nodes (its row: chain 2 + diagonal + closures)
            extra = rng.choice(np.arange(100, 6000 if packed else n), hub, replace=False)     # (forced shape: panels of 8 256 columns)
            lo = np.concatenate([lo, np.full(hub, 7)]); hi = np.concatenate([hi, extra])
        key = np.unique(lo.astype(np.int64) * n + hi)
        ci, cj = (key // n).astype(np.int32), (key % n).astype(np.int32)
        if hub:          # keep the hub row's length exact: drop random edges at node 7 other than the planted ones
            planted = np.isin(ci.astype(np.int64) * n + cj, 7 * np.int64(n) + np.sort(extra))
            drop = ((ci == 7) | (cj == 7)) & ~planted
            ci, cj = ci[~drop], cj[~drop]
        m = len(ci)
        cw = 0.5 + rng.random(m)
        fi = np.arange(n - 1, dtype=np.int32)
        P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, cw)
        P.set_start(reference_start_block(n)[:, 0].copy())
        x = np.ones(m)
        P.set_x(x)
        res = {}
        for mode in ("0", "1"):
            P.set_option("panel", int(mode))
            lam, v, _ = P.fiedler(tol=1e-10)
            res[mode] = (lam, v, int(P.stats.lanczos_steps))
        assert abs(res["1"][0] - res["0"][0]) <= 1e-12 * res["0"][0], (n, deg, hub, res["0"][0], res["1"][0])
        out8 = (C.c_int * 12)()
        nnz_l = n + 2 * (n - 1 + m)
        with _lib.default_options(panel=1):
            assert _lib.load().machip_panel_plan(n, nnz_l, hub + 3 if hub else 127, out8) == 0 and out8[0] == 1   # (the plan admits all four)
        same = res["1"][0] == res["0"][0] and res["1"][2] == res["0"][2] and np.array_equal(res["1"][1], res["0"][1])
        assert same == packed, (n, deg, hub, packed, res["0"][0], res["1"][0], res["0"][2], res["1"][2])
        assert np.abs(sign_align(res["1"][1], res["0"][1]) - res["0"][1]).max() <= 1e-7
        ip, ix, da = P.laplacian_csr()
        L = sp.csr_matrix((da, ix, ip), shape=(n, n))
        assert int(np.diff(ip).max()) == (hub + 3 if hub else int(np.diff(ip).max()))
        v1 = res["1"][1]
        assert np.abs(L @ v1 - res["1"][0] * v1).sum() / abs(L).sum(axis=1).max() < 1e-8      # nx:246 on SciPy's SpMV
        w = spla.eigsh(L, k=2, which="SA", tol=1e-10, ncv=64, v0=np.ones(n) + 0.01 * rng.random(n), return_eigenvectors=False)
        assert abs(np.sort(w)[1] - res["1"][0]) <= 1e-8 * res["1"][0]
        P.close()


def test_shifted_panel_step_across_the_instantiations_the_planner_reaches():
    """The shifted panel step is a table of template instantiations (`k_pan_mul8<LPT, TWT>`: operand columns per lane pair x tiles per
    worker wave; `k_pan_finu<block, panels>`), and the bench sits on ONE of them (9, 3 | 6 panels).  The planner reaches others with the
    size alone -- here seven sizes from 3 500 to 135 500 rows, chosen so that `machip_panel_plan` names seven different (LPT, TWT) pairs, among
    them the 5- and 8-tile forms that only exist beyond n = 1e5: the forced panel solve must run the shifted form (a drift factor is reported),
    agree with the gather step to 1e-12 in lambda_2 and pass the reference's stop rule evaluated with SciPy's SpMV."""
    rng = np.random.default_rng(23)
    pairs = set()
    for n, deg in ((3500, 12), (8000, 12), (12500, 16), (32000, 16), (61001, 14), (104000, 14), (135500, 12)):
        m0 = n * deg // 2
        a = rng.integers(0, n, m0); b = rng.integers(0, n, m0)
        keep = np.abs(a - b) > 1
        key = np.unique(np.minimum(a, b)[keep].astype(np.int64) * n + np.maximum(a, b)[keep])
        ci, cj = (key // n).astype(np.int32), (key % n).astype(np.int32)
        m = len(ci)
        fi = np.arange(n - 1, dtype=np.int32)
        P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, 0.5 + rng.random(m))
        P.set_start(reference_start_block(n)[:, 0].copy())
        P.set_x(np.ones(m))
        nnz = P.assemble()
        out12 = (C.c_int * 12)()
        with _lib.default_options(panel=1):
            assert _lib.load().machip_panel_plan(n, nnz, 127, out12) == 0
        assert out12[0] == 1 and out12[8] == 1 and out12[11] == 1, (n, list(out12))       # shifted form, one cell per workgroup
        pairs.add((out12[9], out12[10]))
        res = {}
        for mode in (0, 1):
            P.set_option("panel", mode)
            lam, v, _ = P.fiedler(tol=1e-10)
            res[mode] = (lam, v, P.stats.drift, P.solve_mode()[0])
        assert res[1][3] == 2 and res[1][2] > 0.0 and res[0][2] == 0.0, (n, res[0][2:], res[1][2:])
        assert abs(res[1][0] - res[0][0]) <= 1e-12 * res[0][0], (n, list(out12), res[0][0], res[1][0])
        ip, ix, da = P.laplacian_csr()
        L = sp.csr_matrix((da, ix, ip), shape=(n, n))
        v1 = res[1][1]
        assert np.abs(L @ v1 - res[1][0] * v1).sum() / abs(L).sum(axis=1).max() < 1e-8, (n, list(out12))
        assert np.abs(sign_align(v1, res[0][1]) - res[0][1]).max() <= 1e-6
        P.close()
    assert len(pairs) == 7 and {(8, 5), (7, 8), (9, 3)} <= pairs, pairs


def test_automatic_mode_picks_the_multi_cell_panel_step_at_n_200000():
    """k_pan_mul_multi (several row blocks per workgroup, the panel kept in LDS) is what the AUTOMATIC mode
    takes from ~33 entries per row at n = 150 000 .. 400 000 (plan.h), but rounds 1-4 only ever ran it forced onto er2000 through an
    option.  Here the product picks it by itself: n = 200 000, ~36 entries per row, no option set; machip_panel_plan confirms the
    shape (more (row block, panel) cells than one wave of 256 workgroups), lambda_2 equals SciPy's eigsh to 1e-8 and the pair
    passes the reference's stop rule (nx:246) evaluated with SciPy's SpMV."""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(5)
    n, deg = 200000, 33
    mraw = n * deg // 2
    a = rng.integers(0, n, mraw); b = rng.integers(0, n, mraw)
    keep = np.abs(a - b) > 1
    key = np.unique(np.minimum(a, b)[keep].astype(np.int64) * n + np.maximum(a, b)[keep])
    ci, cj = (key // n).astype(np.int32), (key % n).astype(np.int32)
    m = len(ci)
    cw = 0.5 + rng.random(m)
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, cw)
    P.set_start(reference_start_block(n)[:, 0].copy())
    P.set_x(np.ones(m))
    nnz = P.assemble()
    out8 = (C.c_int * 12)()
    assert _lib.load().machip_panel_plan(n, nnz, 127, out8) == 0
    on, NP, Cc, NB = out8[0], out8[1], out8[2], out8[3]
    assert on == 1 and NB * NP > 256 and nnz / n >= 33.0, (list(out8), nnz / n)       # several cells per workgroup: k_pan_mul_multi
    lam, v, _ = P.fiedler(tol=1e-8)
    assert P.solve_mode()[0] == 2                                                       # the column-panel step served the solve
    ip, ix, da = P.laplacian_csr()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    assert np.abs(L @ v - lam * v).sum() / abs(L).sum(axis=1).max() < 1e-8            # nx:246 on SciPy's SpMV
    w = spla.eigsh(L, k=2, which="SA", tol=1e-10, ncv=64, v0=np.ones(n) + 0.01 * rng.random(n), return_eigenvectors=False)
    assert abs(np.sort(w)[1] - lam) <= 1e-8 * lam
    P.close()


def test_panel_band_split_when_band_entries_are_missing_or_weightless():
    """Band split of the column-panel step (panel.h, PanView::band: diagonal and columns r -/+ 1 stay out of the tiles, k_pan_fin adds
    them) on graphs WITHOUT a complete chain: the fixed edges are a random spanning tree (most rows have no neighbour at r -/+ 1, some have
    one, a few both; rows 0 and n - 1 have one-sided bands), candidates include pairs (r, r + 1) so that a band entry can come from a
    candidate and be zero-weighted by x.  lambda_2 / vector of the band-split form = the round-3 layout (option panel_band = 0) = the
    gather step to 1e-12, SciPy to 1e-8; several panels and row blocks, n not a multiple of anything."""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(23)
    n = 7001
    par = np.array([rng.integers(max(0, i - 40), i) for i in range(1, n)])          # tree: node i hangs on an earlier node nearby
    fi = np.minimum(par, np.arange(1, n)).astype(np.int32); fj = np.maximum(par, np.arange(1, n)).astype(np.int32)
    fw = 0.5 + rng.random(n - 1)
    m0 = 40000
    a = rng.integers(0, n, m0); b = rng.integers(0, n, m0)
    a[:500] = rng.integers(0, n - 1, 500); b[:500] = a[:500] + 1                     # band entries that come from candidates
    keep = a != b
    key = np.unique(np.minimum(a, b)[keep].astype(np.int64) * n + np.maximum(a, b)[keep])
    ci, cj = (key // n).astype(np.int32), (key % n).astype(np.int32)
    m = len(ci)
    cw = 0.5 + rng.random(m)
    x = rng.random(m); x[rng.random(m) < 0.4] = 0.0
    P = _lib.Problem(n, fi, fj, fw, ci, cj, cw)
    P.set_start(reference_start_block(n)[:, 0].copy())
    P.set_x(x)
    res = {}
    for tag, opts in (("gather", {"panel": 0}), ("band", {"panel": 1, "panel_np": 5, "panel_nb": 3}),
                      ("tiles", {"panel": 1, "panel_np": 5, "panel_nb": 3, "panel_band": 0})):
        for k_ in ("panel", "panel_band", "panel_np", "panel_nb"):
            P.set_option(k_, None)
        P.set_options(**opts)
        lam, v, _ = P.fiedler(tol=1e-10)
        res[tag] = (lam, v)
    for tag in ("band", "tiles"):
        assert abs(res[tag][0] - res["gather"][0]) <= 1e-12 * res["gather"][0], (tag, res[tag][0], res["gather"][0])
        assert np.abs(sign_align(res[tag][1], res["gather"][1]) - res["gather"][1]).max() <= 1e-7
    assert not np.array_equal(res["band"][1], res["gather"][1])                       # (the panel form really ran: other roundings)
    ip, ix, da = P.laplacian_csr()
    L = sp.csr_matrix((da, ix, ip), shape=(n, n))
    w = spla.eigsh(L, k=2, which="SA", tol=1e-10, ncv=64, v0=np.ones(n) + 0.01 * rng.random(n), return_eigenvectors=False)
    assert abs(np.sort(w)[1] - res["band"][0]) <= 1e-8 * res["band"][0]
    P.close()


def test_degenerate_inputs():
    # n = 2: single fixed edge, no candidates (the smallest problem the reference's asserts admit)
    mac = MAC([Edge(0, 1, 2.5)], [], 2)
    assert abs(mac.evaluate_objective(np.zeros(0)) - 5.0) < 1e-9        # lambda_2 of w*[[1,-1],[-1,1]] = 2w
    r, w, val = mac.solve(0, np.zeros(0))
    assert r.shape == (0,) and abs(val - 5.0) < 1e-9
    # fixed edges alone are disconnected; the candidates connect the graph
    fixed = [Edge(0, 1, 1.0), Edge(2, 3, 1.0)]
    cand = [Edge(1, 2, 2.0), Edge(0, 3, 0.5)]
    mac = MAC(fixed, cand, 4)
    with pytest.raises(_lib.Disconnected):
        mac.evaluate_objective(np.zeros(2))
    mo = oracle.MacOracle([0, 2], [1, 3], [1., 1.], [1, 0], [2, 3], [2.0, 0.5], 4)
    for x in (np.array([1.0, 0.0]), np.array([0.3, 0.9]), np.ones(2)):
        assert abs(mac.evaluate_objective(x) - oracle.dense_fiedler(mo.laplacian(x))[0]) < 1e-9
    f, g = mac.problem(np.array([0.3, 0.9]))
    fo, go = mo.problem(np.array([0.3, 0.9]))
    assert abs(f - fo) < 1e-9 and np.allclose(g, go, rtol=1e-6, atol=1e-12)
    # x entries at / below the selection threshold are dropped exactly like mac.py:85
    lam_thr = mac.evaluate_objective(np.array([1.0, 1e-10]))
    assert abs(lam_thr - oracle.dense_fiedler(mo.laplacian(np.array([1.0, 0.0])))[0]) < 1e-9
    lam_above = mac.evaluate_objective(np.array([1.0, 2e-10]))
    assert lam_above > lam_thr


def test_wide_weight_range_and_large_ids():
    """Weights spanning 8 decades and a sparse id space (isolated-looking high ids are still nodes)."""
    rng = np.random.default_rng(11)
    n = 500
    fi = np.arange(n - 1); fj = fi + 1
    fw = 10.0 ** rng.uniform(-3, 3, n - 1)
    ci = rng.integers(0, n, 3000); cj = rng.integers(0, n, 3000)
    keep = ci != cj
    ci, cj = ci[keep], cj[keep]
    cw = 10.0 ** rng.uniform(-4, 4, len(ci))
    x = rng.random(len(ci))
    P = _lib.Problem(n, fi, fj, fw, ci, cj, cw)
    P.set_x(x)
    lam, v, _ = P.fiedler(x0=reference_start_block(n)[:, 0].copy())
    mo = oracle.MacOracle(fi, fj, fw, ci, cj, cw, n)
    lam_d, v_d, _ = oracle.dense_fiedler(mo.laplacian(x))
    assert abs(lam - lam_d) <= LAM_RTOL * lam_d
    assert P.stats.residual < 1e-8
    assert np.array_equal(P.gradient(), oracle.supergradient(v, ci, cj, cw))
    P.close()


def test_madow_rounding_with_objective_reruns():
    """rounding="madow" with random_rounding_max_iters > 1 re-evaluates lambda_2 per draw
    (mac/utils/rounding.py:63-75 -> MAC.evaluate_objective); exact-K selection must hold."""
    g = load_golden("g2o_intel")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    k = int(g["k"])
    np.random.seed(3)
    rounded, w, u = mac.solve(k, g["x_init"], max_iters=6, rounding="madow", random_rounding_max_iters=3)
    assert rounded.sum() == k and set(np.unique(rounded)) <= {0.0, 1.0}
    assert mac.evaluate_objective(rounded) <= u + 1e-9
    r2, w2, u2, rt = mac.solve(k, g["x_init"], max_iters=6, return_rounding_time=True, fallback=True)
    assert rt >= 0 and r2.sum() == k and np.allclose(w, w2, atol=1e-12)


def test_eval_batch_matches_single_evaluations():
    """machip_eval_batch (MAC.evaluate_objective_batch): B selection vectors solved concurrently on the handle's
    evaluation lanes give, entry by entry, the value the one-at-a-time MAC.evaluate_objective gives (identical
    kernels and start vector, so bit-identical), against the reference goldens where they exist; the handle's own
    state (x, warm-start vector) is untouched."""
    g = load_golden("g2o_intel")
    # (bit-identity holds mode for mode: the lanes keep to the single-CU Lanczos kernel, a standalone automatic solve may take the
    # exact chain + closures preconditioner on this graph -- round 4 -- and then agrees to the solver tolerance, checked below)
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="hip_lanczos")
    m, k = len(g["cw"]), int(g["k"])
    rs = np.random.RandomState(11)
    X = np.zeros((13, m))
    X[0] = g["x_init"]; X[1] = 1.0; X[2] = g["rounded"]; X[3] = g["unrounded"]
    for b in range(4, 13):
        X[b, rs.choice(m, k, replace=False)] = 1.0
    mac._dev.set_x(g["x_init"])
    lam = mac.evaluate_objective_batch(X)
    assert abs(lam[0] - g["lam_init"]) <= LAM_RTOL * g["lam_init"]
    assert abs(lam[1] - g["lam_all"]) <= LAM_RTOL * g["lam_all"]
    assert abs(lam[2] - g["lam_rounded"]) <= LAM_RTOL * g["lam_rounded"]
    assert np.array_equal(mac._dev.get_x(), g["x_init"])
    single = np.array([mac.evaluate_objective(X[b]) for b in range(13)])
    assert np.array_equal(lam, single)
    lam2 = mac.evaluate_objective_batch(X[::-1])          # lanes are reusable; order of arrival does not matter
    assert np.array_equal(lam2, lam[::-1])
    mac_auto = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    assert np.array_equal(mac_auto.evaluate_objective_batch(X), lam)               # lanes: the same kernels whatever the handle's mode
    assert np.allclose([mac_auto.evaluate_objective(X[b]) for b in range(13)], lam, rtol=LAM_RTOL, atol=0)
    # a disconnected selection is reported per entry, not as a failure of the call
    gp = load_golden("petersen_solve_k3")
    P = _lib.Problem(10, np.arange(8), np.arange(8) + 1, np.ones(8), np.array([0, 2]), np.array([9, 5]), np.ones(2))
    lamp, st = P.eval_batch(np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]))
    assert st[1] == _lib.DISCONNECTED and st[0] == _lib.OK and st[2] == _lib.OK and lamp[0] > 0
    P.close()


def test_round2_advice_regressions():
    """Fixes of the round-2 advisor findings, each pinned:
    (a) evaluation lanes track the handle's start vector PER LANE (a small batch after machip_set_start used to leave
        the lanes it did not touch on the old vector) and the x0 argument of machip_fiedler counts as a change;
    (b) MAC.evaluate_objective_batch raises on a disconnected entry like evaluate_objective does;
    (c) find_fiedler_pair reuses one cached handle and computes exactly what a fresh handle computes;
    (d) MAC.Cache.Q holds an ndarray (the reference's slot type), and the warm start still works;
    (e) a handle cannot join two communicators."""
    g = load_golden("g2o_intel")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="hip_lanczos")   # (one mode on lanes and handle: bit-identity is mode for mode)
    n, m, k = int(g["n"]), len(g["cw"]), int(g["k"])
    rng = np.random.default_rng(5)
    X = np.zeros((8, m))
    for b in range(8):
        X[b, rng.choice(m, k, replace=False)] = 1.0
    dev = mac._dev
    lam_a = mac.evaluate_objective_batch(X)                    # creates 8 lanes on start vector A (the reference's)
    startB = rng.normal(size=n)
    dev.set_start(startB)
    _ = mac.evaluate_objective_batch(X[:2])                    # refreshes lanes 0-1 only
    lam_b = mac.evaluate_objective_batch(X)                    # lanes 2..7 must not be on A any more
    single_b = np.array([mac.evaluate_objective(X[b]) for b in range(8)])
    assert np.array_equal(lam_b, single_b)
    startC = rng.normal(size=n)
    dev.set_x(X[0]); lamC, _, _ = dev.fiedler(x0=startC, want_vec=False)      # x0 path replaces the start vector too
    lam_c = mac.evaluate_objective_batch(X)
    assert lam_c[0] == lamC and np.array_equal(lam_c, [mac.evaluate_objective(X[b]) for b in range(8)])
    assert np.allclose(lam_a, lam_c, rtol=1e-7)
    # (b)
    macp = MAC([Edge(i, i + 1, 1.0) for i in range(8)], [Edge(0, 9, 1.0), Edge(2, 5, 1.0)], 10)
    with pytest.raises(_lib.Disconnected):
        macp.evaluate_objective(np.array([0.0, 1.0]))
    with pytest.raises(_lib.Disconnected):
        macp.evaluate_objective_batch(np.array([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]]))
    # (c)
    L = mac.laplacian(g["x_init"])
    first = find_fiedler_pair(L)
    _lib.load().machip_release_cache()
    fresh = find_fiedler_pair(L)                                # fresh handle
    again = find_fiedler_pair(L)                                # cached handle
    gs = load_golden("g2o_sphere2500")
    other = find_fiedler_pair(MAC(edges_of(gs, "f"), edges_of(gs, "c"), int(gs["n"])).laplacian(gs["x_init"]))   # other n: cache replaced
    back = find_fiedler_pair(L)
    for r in (fresh, again, back):
        assert r[0] == first[0] and np.array_equal(r[1], first[1]) and np.array_equal(r[2], first[2])
    assert abs(other[0] - gs["lam_init"]) <= LAM_RTOL * gs["lam_init"]
    assert abs(first[0] - g["lam_init"]) <= LAM_RTOL * g["lam_init"]
    # (d)
    cache = MAC.Cache()
    f0, g0 = mac.problem(g["x_init"], cache=cache)
    assert isinstance(cache.Q, np.ndarray) and cache.Q.shape == (n, 1) and abs(np.linalg.norm(cache.Q) - 1) < 1e-12
    f1, g1 = mac.problem(g["x_init"], cache=cache)             # warm start from the previous vector
    assert abs(f1 - f0) <= LAM_RTOL * f0
    # (e)
    P1, P2 = problem_of(g), problem_of(g)
    hs = (C.c_void_p * 2)(P1._h, P2._h)
    lib = _lib.load()
    assert lib.machip_comm_init_local(hs, 2) == _lib.OK
    assert lib.machip_comm_init_local(hs, 2) == _lib.BAD_ARG
    assert lib.machip_comm_init(P1._h, 0, 1, C.create_string_buffer(128)) == _lib.BAD_ARG
    P1.close(); P2.close()


def test_madow_multi_try_is_batched_and_picks_like_the_sequential_loop():
    """round_madow(max_iters > 1) (mac/utils/rounding.py:63-75): the draws consume the random stream exactly as the
    reference's loop does and the winner of the batched evaluation is the winner of the one-by-one loop."""
    from mac_amd.utils.rounding import round_madow
    g = load_golden("g2o_sphere2500")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
    k = int(g["k"])
    w = g["unrounded"]
    a = round_madow(w, k, seed=np.random.RandomState(42), value_fn=mac.evaluate_objective, max_iters=6)
    b = round_madow(w, k, seed=np.random.RandomState(42), max_iters=6, batch_value_fn=mac.evaluate_objective_batch)
    assert np.array_equal(a, b) and a.sum() == k
    one = round_madow(w, k, seed=np.random.RandomState(42))
    assert np.array_equal(one, g["madow"])                 # first draw == the reference's single draw (golden)
    assert mac.evaluate_objective(b) >= mac.evaluate_objective(one) - 1e-12


def test_device_round_nearest_matches_oracle():
    """machip_round_nearest == oracle.round_nearest (the reference's tie-broken top-k, rounding.py:30-42)
    bit for bit, including heavy ties in the rounded selection weights and in the edge weights."""
    rng = np.random.default_rng(21)
    g = load_golden("er2000_solve")
    n, m = int(g["n"]), len(g["cw"])
    cases = []
    cw_tied = np.round(g["cw"], 1)                       # many equal edge weights
    for cw in (g["cw"], cw_tied, np.ones(m)):
        P = _lib.Problem(n, g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], cw)
        xs = [g["unrounded"], np.round(rng.random(m), 1), np.round(rng.random(m), 2) * (rng.random(m) < 0.3),
              np.zeros(m), np.full(m, 0.5), rng.random(m)]
        for x in xs:
            P.set_x(x)
            for k in (0, 1, 17, int(g["k"]), m // 2, m - 1, m):
                got = P.round_nearest(k, decimals=10)
                exp = oracle.round_nearest(x, k, cw, 10)
                assert np.array_equal(got, exp), (k,)
                cases.append(k)
            got = P.round_nearest(int(g["k"]), decimals=None)     # plain top-k branch (rounding.py:21-28)
            assert got.sum() == int(g["k"])
            kth = np.sort(x)[-int(g["k"])]
            assert np.all(got[x > kth] == 1) and np.all(got[x < kth] == 0)
        P.close()
    assert len(cases) == 3 * 6 * 7


def test_solve_rounding_matches_reference_goldens():
    for nm in ("g2o_intel", "g2o_sphere2500", "er2000_solve"):
        g = load_golden(nm)
        mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]))
        mac._dev.set_x(g["unrounded"])
        assert np.array_equal(mac._dev.round_nearest(int(g["k"]), decimals=10), g["rounded"])


def test_one_launch_select_equals_the_multi_launch_select():
    """Short candidate lists (<= 32 768: every pose graph) run the whole top-K select in one single-workgroup launch
    (k_sel_small, option `sel_small`); it must leave exactly the selection of the six-pass form -- rounding with its
    prefer-high tie rule and the LP vertex with its lowest-index rule, on tie-heavy and on generic iterates, for every k from
    nothing to everything -- and both must equal the oracle's."""
    rng = np.random.default_rng(17)
    for n, m in ((40, 7), (600, 1000), (3000, 32768)):
        a = rng.integers(0, n, 3 * m); b = rng.integers(0, n, 3 * m)
        keep = np.abs(a - b) > 1
        pairs = np.unique(np.stack([np.minimum(a, b)[keep], np.maximum(a, b)[keep]], 1), axis=0)[:m]
        m = len(pairs)
        ci, cj = pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int32)
        cw = rng.choice(np.array([1.0, 2.0, 3.5]), size=m)
        fi = np.arange(n - 1, dtype=np.int32)
        P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, cw)
        P.set_start(reference_start_block(n)[:, 0].copy())
        for kind in ("ties", "generic"):
            x = rng.choice(np.array([0.0, 0.0, 1.0 / 3.0, 0.4, 0.4, 0.6, 2.0 / 3.0, 1.0]), size=m) if kind == "ties" else rng.random(m)
            P.set_x(x)
            P.assemble(); P.fiedler(want_vec=False); g = P.gradient()
            for k in sorted({0, 1, m // 7, m // 2, m - 1, m}):
                got = {}
                for flag in ("0", "1"):
                    P.set_option("sel_small", int(flag))
                    got[flag] = (P.round_nearest(k, decimals=10), P.lp_topk(k))
                assert np.array_equal(got["0"][0], got["1"][0]) and np.array_equal(got["0"][1], got["1"][1]), (n, kind, k)
                assert np.array_equal(got["1"][0], oracle.round_nearest(x, k, cw, 10)), (n, kind, k)
                s = got["1"][1]
                assert int(s.sum()) == k and (k in (0, m) or g[s > 0].min() >= g[s == 0].max())
        P.close()


def test_selection_is_deterministic_on_massive_ties():
    """Regression for a race in the radix select: the last workgroup could read a histogram bin before
    every workgroup's (fire-and-forget) add had been performed at the memory side; with rounded iterates
    nearly all keys share one or two bins and the selection came out a few elements too large in ~20 % of
    the calls right after a config-2 solve (tools/round_check.py).  Repeated calls must give exactly k
    ones, identical to the oracle's."""
    n = 10000
    ci, cj = make_er(n, 0.01, 0)
    m = len(ci); k = m // 10
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(m))
    P.set_start(reference_start_block(n)[:, 0].copy())
    x0 = np.zeros(m); x0[np.random.default_rng(0).choice(m, k, replace=False)] = 1.0
    P.set_x(x0)
    for it in range(20):
        P.fw_step(k, it); P.fw_commit()
    w = P.get_x()
    want = oracle.round_nearest(w, k, np.ones(m), 10)
    assert int(want.sum()) == k
    for rep in range(40):
        r = P.round_nearest(k, decimals=10)
        assert int(r.sum()) == k and np.array_equal(r, want), rep
    # a synthetic all-ties input as well
    rng = np.random.default_rng(3)
    x = rng.choice(np.array([0.0, 0.0, 0.0, 1.0 / 3.0, 0.4, 0.4, 0.6, 2.0 / 3.0, 1.0]), size=m)
    P.set_x(x)
    want = oracle.round_nearest(x, k, np.ones(m), 10)
    for rep in range(10):
        assert np.array_equal(P.round_nearest(k, decimals=10), want), rep
    P.close()


def test_config2_first_lp_vertices_equal_the_exact_ones():
    """Where the C2 trajectories fork (see the next test): the top-K sets s_0, s_1 of the HIP path against
    the ones an exact dense eigen-solve gives (tests/golden/er10k_exact_topk.npz: numpy eigh of the
    reference's own L(x)): lambda_2 equal to 1e-11, the sets equal up to the one boundary entry that a
    1e-8-accurate eigenvector cannot decide (the reference's own s_1 misses exactly one of 50 053 too)."""
    g = load_golden("er10k_exact_topk")
    n = 10000
    ci, cj = make_er(n, 0.01, 0)
    m, k = len(ci), int(g["k"])
    assert m == int(g["m"])
    assert k - len(np.intersect1d(g["ref_s0"], g["exact_s0"])) == 0
    assert k - len(np.intersect1d(g["ref_s1"], g["exact_s1"])) == 1      # the fork
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(m))
    P.set_start(reference_start_block(n)[:, 0].copy())
    x = np.zeros(m); x[np.random.default_rng(0).choice(m, k, replace=False)] = 1.0
    for it in range(2):
        P.set_x(x)
        lam, _, _ = P.fiedler()
        assert abs(lam - float(g[f"exact_lam{it}"])) <= 1e-11 * lam
        assert abs(float(g[f"ref_f{it}"]) - float(g[f"exact_lam{it}"])) <= LAM_RTOL * lam
        P.gradient(want=False)
        s = P.lp_topk(k)
        # the 50 053-rd and 50 054-th largest gradient entries differ by 4-6e-6 relative (golden boundary_gap_rel):
        # an eigenvector that satisfies the 1e-8 residual rule decides that one place either way, depending on the
        # rounding of the launch shape in use -- like the reference's own s_1 -- and nothing else
        assert len(np.setdiff1d(np.nonzero(s)[0], g[f"exact_s{it}"])) <= 1
        assert float(g[f"boundary_gap_rel{it}"]) < 1e-5
        x = x + 2.0 / (it + 2) * (s - x)
    P.close()


def test_config2_twenty_iterations_match_reference_trajectory():
    """The bench workload itself (BASELINE.json configs[1], 20 Frank-Wolfe iterations from the bench's
    x0, stop tests off) against the trajectory the REAL reference produced (tests/golden/er10k_solve.npz,
    about two CPU-hours there).  While both runs hold the same x (iterations 0 and 1: x1 is the first LP
    vertex) lambda_2 agrees to 1e-8.  From then on the comparison is necessarily looser: with unit
    weights the 50 053-rd and 50 054-th largest of 500 534 gradient entries differ by ~1e-9 relative,
    less than what either solver's 1e-8 residual leaves in its eigenvector, so a handful of top-K
    choices differ ("ties in g can legitimately flip argpartition choices") and the two
    trajectories drift apart like any two runs of this degenerate problem would.  Measured: 3e-9 at
    iteration 2, 3e-7 at 3-4, percent level from 6 on, same dual bound to 6e-7, same end value to 0.4 %.
    The HIP trajectory itself does not move when its tolerance is tightened to 1e-11."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "er10k_solve.npz")):
        pytest.skip("er10k_solve fixture not generated (tests/golden/make_golden.py er10k_solve)")
    g = load_golden("er10k_solve")
    n = 10000
    ci, cj = make_er(n, 0.01, 0)
    m, k = len(ci), int(g["k"])
    assert m == int(g["m"])
    fi = np.arange(n - 1, dtype=np.int32)
    P = _lib.Problem(n, fi, fi + 1, np.ones(n - 1), ci, cj, np.ones(m))
    P.set_start(reference_start_block(n)[:, 0].copy())
    x0 = np.zeros(m); x0[g["x0_idx"]] = 1.0
    P.set_x(x0)
    fs, supp, u = [], [], np.inf
    for it in range(len(g["f_traj"])):
        f, dual, gn = P.fw_step(k, it)
        u = min(u, dual)
        fs.append(f); supp.append(int(P.stats.support))
        P.fw_commit()
    fs, ref = np.array(fs), np.asarray(g["f_traj"])
    rel = np.abs(fs - ref) / ref
    assert np.all(rel[:2] <= LAM_RTOL), rel[:2]                 # identical inputs -> identical lambda_2
    assert np.array_equal(supp[:3], g["supp"][:3])              # |supp(x_2)| = |s_0 u s_1| barely moves with a few swaps
    assert np.all(rel[2:5] <= 5e-6), rel[2:5]                   # a few swapped top-K entries out of 50 053
    assert np.all(rel <= 0.12), rel                             # drifted apart, same regime
    assert np.all(np.abs(np.array(supp) - g["supp"]) <= 0.02 * g["supp"])
    assert abs(u - float(g["upper"])) <= 1e-5 * abs(float(g["upper"]))   # dual bound is set early
    assert abs(fs[-1] - ref[-1]) <= 0.03 * ref[-1]
    w = P.get_x()
    assert abs(np.count_nonzero(w) - int(g["unrounded_nnz"])) <= 0.02 * int(g["unrounded_nnz"])
    assert abs(w.sum() - k) < 1e-6 and abs(float(g["unrounded_sum"]) - k) < 1e-6
    assert int(P.round_nearest(k, decimals=10).sum()) == k == len(g["rounded_idx"])
    P.close()


def test_find_fiedler_pair_on_pose_graph_laplacians_uses_the_structured_modes():
    """The bare find_fiedler_pair(L) surface (mac/utils/fiedler.py:9-44) detects a chain-like Laplacian
    from the CSR itself: same pair as the forced Lanczos path, far fewer dependent launches on a stiff one."""
    g = load_golden("g2o_kitti_05")
    mac = MAC(edges_of(g, "f"), edges_of(g, "c"), int(g["n"]), fiedler_method="hip_lanczos")
    L = mac.laplacian(g["x_init"])
    lam_ref = mac.evaluate_objective(g["x_init"]); steps_ref = mac.last_stats["lanczos_steps"]
    lam, v, X = find_fiedler_pair(L)
    assert abs(lam - lam_ref) <= LAM_RTOL * lam_ref and abs(lam - g["lam_init"]) <= LAM_RTOL * g["lam_init"]
    assert X.shape == (int(g["n"]), 4) and np.abs(X.T @ X - np.eye(4)).max() < 1e-6 and np.allclose(X[:, 0], v)
    indptr, indices, data = L.indptr, L.indices, L.data
    _, _, _, st = _lib.fiedler_csr(indptr, indices, data, int(g["n"]))
    assert st.lanczos_steps * 5 < steps_ref and st.residual < 1e-8


def test_large_sparse_chain_spmv_and_preconditioned_solve():
    """n = 200 000 rows of ~3 entries: the row-tile SpMV runs with one lane per row (257 row offsets staged by
    256 threads -- a staging bug there went unnoticed until this size), the tridiagonal solve of the
    preconditioned mode runs its multi-workgroup form (n > 16 384).  SpMV against SciPy for every variant,
    lambda_2 through the stop rule and against SciPy's shift-invert Lanczos."""
    import scipy.sparse.linalg as spla
    n, nc = 200000, 3000
    rng = np.random.default_rng(1)
    fi = np.arange(n - 1, dtype=np.int32); fw = rng.uniform(100, 1000, n - 1)
    a = rng.integers(0, n, nc); b = np.clip(a + rng.integers(-3000, 3000, nc), 0, n - 1)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = rng.uniform(100, 300, len(ci))
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(np.ones(len(ci)))
    L = oracle.mac_laplacian(oracle.laplacian_from_edges(fi, fi + 1, fw, n), ci.astype(np.int64), cj.astype(np.int64), cw, np.ones(len(ci)), n)
    v = rng.standard_normal(n)
    ref = L @ v
    for variant in (0, 1, 2):
        y = P.spmv(v, variant=variant)
        assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max(), variant
    lam, vec, _ = P.fiedler()
    assert P.stats.residual < 1e-8 and P.stats.lanczos_steps < 6000          # preconditioned mode (auto)
    assert np.abs(L @ vec - lam * vec).sum() / abs(L).sum(axis=1).max() < 1e-8
    w = spla.eigsh(L + 1e-3 * sp.identity(n, format="csr"), k=2, sigma=0, which="LM", return_eigenvectors=False)
    lam_ref = np.sort(w)[1] - 1e-3
    assert abs(lam - lam_ref) <= 1e-6 * lam_ref
    P.close()


def test_non_finite_input_fails_fast():
    """A NaN edge weight poisons L(x): the solve must come back with an error within moments (the host waits
    for poisoned record slots to be overwritten -- with a bounded budget), not hang.  (A NaN in x itself is
    simply an inactive candidate, as in the reference: `x > tol` is False.)"""
    import time
    g = load_golden("er300_x0")
    cw = np.array(g["cw"], dtype=float)
    k = int(np.argmax(g["x"] > 0.5))
    cw[k] = np.nan
    P = _lib.Problem(int(g["n"]), g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], cw)
    P.set_x(g["x"])
    t0 = time.perf_counter()
    with pytest.raises((_lib.MachipError, AssertionError)):
        P.fiedler(max_steps=2000)
    assert time.perf_counter() - t0 < 30.0
    P.close()
    x = np.array(g["x"], dtype=float); x[3] = np.nan
    P = problem_of(g)
    P.set_x(x)
    x0 = np.where(np.isnan(x), 0.0, x)
    lam, _, _ = P.fiedler()
    P.set_x(x0)
    assert lam == P.fiedler()[0]
    P.close()


@pytest.mark.parametrize("nm", ["intel", "kitti_05", "city10000", "kitti_02", "ais2klinik"])
def test_exact_chain_plus_closures_preconditioner(nm):
    """woodbury.h: with at most 2 048 active closures the preconditioned mode inverts L + sigma I exactly
    (chain tridiagonal + low-rank closures, capacitance matrix inverted by the hand-written k_gj_step) and converges in a handful of
    iterations; same pair as the tridiagonal-preconditioned mode and as an independent SciPy solve."""
    import scipy.sparse.linalg as spla
    g = load_golden("g2o_" + nm)
    out = {}
    for flag in ("1", "0"):
        P = problem_of(g)
        P.set_option("woodbury", int(flag))
        P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
        x = np.zeros(len(g["cw"])); x[: min(len(x), 1000)] = 1.0          # <= 2 048 active closures
        x[::3] *= 0.37
        P.set_x(x)
        P.set_solver(2)
        lam, v, _ = P.fiedler()
        out[flag] = (float(lam), int(P.stats.lanczos_steps), float(P.stats.residual), float(np.abs(v).sum()))
        P.close()
    x = np.zeros(len(g["cw"])); x[: min(len(x), 1000)] = 1.0
    x[::3] *= 0.37
    n = int(g["n"])
    L = oracle.mac_laplacian(oracle.laplacian_from_edges(g["fi"], g["fj"], g["fw"], n), g["ci"].astype(np.int64), g["cj"].astype(np.int64), g["cw"], x, n)
    w = spla.eigsh(L + 1e-3 * sp.identity(n, format="csr"), k=2, sigma=0, which="LM", return_eigenvectors=False)
    lam_ref = np.sort(w)[1] - 1e-3
    for flag in ("1", "0"):
        assert abs(out[flag][0] - lam_ref) <= 1e-7 * lam_ref and out[flag][2] < 1e-8
    assert abs(out["1"][0] - out["0"][0]) <= LAM_RTOL * lam_ref
    assert out["1"][1] <= 40 and out["1"][1] < out["0"][1]          # a handful of iterations instead of dozens to hundreds
    assert abs(out["1"][3] - out["0"][3]) <= 1e-5 * out["0"][3]      # same vector (its 1-norm)


def test_exact_preconditioner_on_a_large_chain_with_weak_links():
    """n = 50 000 (> 16 384: batched multi-workgroup column solves), chain weights over three decades -- weak
    links bridged only by closures, the case the tridiagonal preconditioner cannot handle -- and 400 closures:
    the exact (Woodbury) preconditioner converges in a few dozen iterations; lambda_2 against SciPy."""
    import scipy.sparse.linalg as spla
    n, nc = 50000, 400
    rng = np.random.default_rng(7)
    fi = np.arange(n - 1, dtype=np.int32); fw = 10.0 ** rng.uniform(0, 3, n - 1)
    a = rng.integers(0, n, nc); b = rng.integers(0, n, nc)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = 10.0 ** rng.uniform(0, 2.5, len(ci))
    x = rng.uniform(0.2, 1.0, len(ci))
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(x)
    lam, vec, _ = P.fiedler()
    assert P.stats.residual < 1e-8 and P.stats.lanczos_steps <= 60
    L = oracle.mac_laplacian(oracle.laplacian_from_edges(fi, fi + 1, fw, n), ci.astype(np.int64), cj.astype(np.int64), cw, x, n)
    assert np.abs(L @ vec - lam * vec).sum() / abs(L).sum(axis=1).max() < 1e-8
    shift = 1e-6
    w = spla.eigsh(L + shift * sp.identity(n, format="csr"), k=2, sigma=0, which="LM", return_eigenvectors=False)
    lam_ref = np.sort(w)[1] - shift
    assert abs(lam - lam_ref) <= 1e-4 * lam_ref          # the stop rule itself resolves a lambda_2 this small no better
    P.close()


def test_stiff_chain_with_thousands_of_closures_escalates_to_the_exact_preconditioner():
    """Fuzz seed 129 of tools/fuzz_modes.py (round 1's non-converging class: > 2 048 active closures AND
    lambda_2 / ||L||_inf ~ 1e-10; n = 36 874, 5 568 closures of which ~3 900 active): the tridiagonal preconditioner
    crawls, the solve escalates to the exact (Woodbury) one -- second tier, up to 16 384 closures -- and converges in
    a few dozen iterations, in the automatic mode; residual checked with SciPy's SpMV, lambda_2 against SciPy's
    shift-invert Lanczos."""
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(1000 + 129)
    n = int(rng.choice([rng.integers(260, 3072), rng.integers(3072, 16384), rng.integers(16384, 60000)]))
    ncl = int(rng.integers(1, max(2, int(n * rng.choice([0.005, 0.05, 0.3])))))
    fi = np.arange(n - 1, dtype=np.int32)
    fw = 10.0 ** rng.uniform(0, rng.choice([0.5, 2, 3]), n - 1)
    a = rng.integers(0, n, ncl); span = int(rng.choice([50, 3000, n]))
    b = np.clip(a + rng.integers(-span, span + 1, ncl), 0, n - 1)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = 10.0 ** rng.uniform(0, 2.5, len(ci))
    x = rng.random(len(ci)); x[rng.random(len(ci)) < 0.3] = 0.0
    assert n == 36874 and len(ci) == 5568
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(x)
    lam, vec, _ = P.fiedler()
    st = P.stats.asdict()
    assert st["support"] > 2048 and st["residual"] < 1e-8
    assert st["lanczos_steps"] < 20000          # (200 000 iterations without converging before the second tier existed)
    L = oracle.mac_laplacian(oracle.laplacian_from_edges(fi, fi + 1, fw, n), ci.astype(np.int64), cj.astype(np.int64), cw, x, n)
    assert np.abs(L @ vec - lam * vec).sum() / abs(L).sum(axis=1).max() < 1e-8
    shift = 1e-9
    w = spla.eigsh(L + shift * sp.identity(n, format="csr"), k=2, sigma=0, which="LM", return_eigenvectors=False)
    lam_ref = np.sort(w)[1] - shift
    assert abs(lam - lam_ref) <= 1e-2 * lam_ref     # lambda_2 = 1.07e-7 = 1.2e-10 ||L||: the residual rule resolves it no better
    P.close()


def test_budget_sweep_driver_reproduces_the_reference_budget():
    """tools/g2o_sweep.py (the loop of examples/g2o_experiment.py:306-336: one device-resident problem, one MAC.solve per
    budget, nearest + Madow rounding, batched evaluation of the four selections): its 20 % row on intel equals the
    reference's golden for that budget (warm starts allowed: same optimum), and lambda_2 grows with the budget."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import g2o_sweep
    g = load_golden("g2o_intel")
    rows = g2o_sweep.sweep(os.path.join(ROOT, "tests", "golden", "data", "intel.g2o"), pcts=(0.1, 0.2, 0.4), verbose=False)
    r = rows[1]
    assert r["k"] == int(g["k"])
    assert abs(r["naive"] - g["lam_init"]) <= 1e-8 * g["lam_init"]
    assert abs(r["upper"] - g["upper"]) <= 1e-5 * g["upper"]
    assert abs(r["nearest"] - g["lam_rounded"]) <= 1e-6 * g["lam_rounded"] or np.array_equal(r["result"], g["rounded"])
    assert np.array_equal(r["madow_x"], g["madow"])
    assert abs(r["madow"] - oracle_of(g).evaluate_objective(g["madow"])) <= 1e-7 * r["madow"]
    assert rows[0]["nearest"] <= rows[1]["nearest"] <= rows[2]["nearest"]
    for r in rows:
        assert r["naive"] <= r["nearest"] * (1 + 1e-9) and r["unrounded"] <= r["upper"] * (1 + 1e-9)


@pytest.mark.parametrize("nm", ["intel", "sphere2500"])
def test_concurrent_budget_sweep_is_bit_identical_to_sequential_solves(nm):
    """MAC.solve_sweep / machip_fw_sweep (the budget sweep of examples/g2o_experiment.py:306-336 run concurrently on the
    evaluation lanes): every budget's (rounded, unrounded, upper) and lambda_2 trajectory are BIT-identical to MAC.solve
    for that budget on a fresh MAC object running the same eigen-solver mode, whatever lane took it, with more budgets than
    lanes, and aggregate throughput is well above the one-at-a-time loop (the small pose graphs leave most of the chip
    idle).  Round 4: the lanes keep to the single-CU Lanczos kernel (throughput), a standalone handle in the automatic mode
    may take the exact chain + closures preconditioner (latency) -- so bit-identity is asserted with the Lanczos mode on both
    sides, and the automatic standalone solve must agree with the sweep to the solver tolerance."""
    import time
    g = load_golden("g2o_" + nm)
    fixed, cand, n = edges_of(g, "f"), edges_of(g, "c"), int(g["n"])
    m = len(cand)
    pcts = (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.25, 0.35)      # 11 budgets > 8 lanes
    ks = [int(p_ * m) for p_ in pcts]
    naive = NaiveGreedy(cand)
    inits = [naive.subset(k) for k in ks]
    seq, traces = [], []
    t_seq = 0.0
    for k, x0 in zip(ks, inits):
        mac1 = MAC(fixed, cand, n, fiedler_method="hip_lanczos")     # fresh handle: clean solver state
        mac1.evaluate_objective(x0)                                  # (first-use costs -- graph capture, lazy buffers -- outside the clock)
        t0 = time.perf_counter()
        seq.append(mac1.solve(k, x0, max_iters=20))
        t_seq += time.perf_counter() - t0
        traces.append([t[0] for t in mac1.trace])
    mac = MAC(fixed, cand, n, fiedler_method="hip_lanczos")
    mac.solve_sweep(ks[:8], inits[:8], max_iters=2)                  # creates the lanes / captures their graphs
    t0 = time.perf_counter()
    par = mac.solve_sweep(ks, inits, max_iters=20)
    t_par = time.perf_counter() - t0
    # automatic mode: the sweep's lanes give the Lanczos results bit for bit, a standalone automatic solve the same to 1e-8
    mac_auto = MAC(fixed, cand, n)
    par_auto = mac_auto.solve_sweep(ks[:3], inits[:3], max_iters=20)
    for j in range(3):
        assert np.array_equal(par_auto[j][1], par[j][1])
    mac_auto.solve(ks[1], inits[1], max_iters=20)
    fa = np.array([t[0] for t in mac_auto.trace]); fl = np.array(traces[1])
    assert len(fa) == len(fl) and np.allclose(fa[:2], fl[:2], rtol=1e-8) and np.allclose(fa, fl, rtol=1e-6)
    for j in range(len(ks)):
        assert np.array_equal(par[j][1], seq[j][1]), j               # unrounded x: bit-identical
        assert np.array_equal(par[j][0], seq[j][0]) and par[j][2] == seq[j][2]
        ft = mac.sweep_trace[j]
        assert np.array_equal(ft[:len(traces[j])], traces[j]) and np.all(np.isnan(ft[len(traces[j]):]))
    # the 20 % budget is the reference's golden run
    j = ks.index(int(g["k"]))
    assert abs(par[j][2] - g["upper"]) <= 1e-5 * abs(g["upper"])
    assert t_seq / t_par >= 2.0, (t_seq, t_par)                      # measured 3-5x (profiles/r3_sweep.txt); generous margin
    # k >= m shortcut and the Madow branch go through the same entry
    both = mac.solve_sweep([m, ks[1]], [np.ones(m), inits[1]], max_iters=20, rounding="madow", seed=np.random.RandomState(42))
    assert np.array_equal(both[0][0], np.ones(m)) and np.array_equal(both[1][1], seq[1][1]) and both[1][0].sum() == ks[1]


def test_fw_run_equals_the_step_by_step_loop_and_reports_solver_modes():
    """machip_fw_run (round 5: the loop of frankwolfe.py:53-76 on the C side -- what MAC.solve and bench.py drive) against the same loop
    driven one machip_fw_step / machip_fw_commit at a time: f, dual bound, ||g||, iteration count under the reference's stop tests
    (mac.py:196-200: gap 1e-4, gradient 1e-8) and the final x are bit-identical, also when a stop test fires early (intel, 80 %
    budget); machip_solve_mode names the launch group that served each solve: single-workgroup kernel, then fused gather step on er2000, the column-panel step when
    forced, the exact chain + closures mode on intel, the single-workgroup kernel on sphere2500's dense iterate, the padded form on city10000."""
    for nm, frac, iters in (("er2000_solve", None, 6), ("g2o_intel", 0.8, 20)):
        g = load_golden(nm)
        m = len(g["cw"])
        k = int(g["k"]) if frac is None else int(frac * m)
        x0 = g["x_init"] if frac is None else NaiveGreedy(edges_of(g, "c")).subset(k)
        P = problem_of(g); P.set_start(reference_start_block(int(g["n"]))[:, 0].copy()); P.set_x(x0)
        ref, u = [], np.inf
        for i in range(iters):
            f, dual, gn = P.fw_step(k, i)
            u = min(u, dual); ref.append((f, dual, gn))
            if gn < 1e-8 or (u - f) < 1e-4 * abs(f):
                break
            P.fw_commit()
        x_ref = P.get_x(); P.close()
        P = problem_of(g); P.set_start(reference_start_block(int(g["n"]))[:, 0].copy()); P.set_x(x0)
        r = P.fw_run(k, iters, gap_tol=1e-4, grad_tol=1e-8)
        assert r["iters"] == len(ref) and r["upper"] == u
        assert [(a, b, c) for a, b, c in zip(r["f"], r["dual"], r["gnorm"])] == ref
        assert np.array_equal(P.get_x(), x_ref)
        assert len(r["stats"]) == r["iters"] and all(int(s_.nnz) > 0 for s_ in r["stats"])
        mds = [md[0] for md in r["modes"]]
        if nm == "er2000_solve":      # a chain + random closures on 2 000 nodes: the single-workgroup kernel while the closures fit it, the fused gather step after
            assert set(mds) <= {1, 4} and mds[-1] == 1, r["modes"]
        else:                         # intel: the exact chain + closures mode (<= 700 active closures), its closure count reported
            assert set(mds) <= {4, 7} and mds[0] == 7 and r["modes"][0][1] > 0, r["modes"]
        if nm == "g2o_intel":
            assert len(ref) < iters                       # (the duality-gap test fired: the reference stops early on this budget too)
        P.close()
    g = load_golden("er2000_xfrac")
    P = problem_of(g); P.set_option("panel", 1); P.set_x(g["x"]); P.fiedler(want_vec=False)
    assert P.solve_mode()[0] == 2
    P.close()
    for nm, want in (("g2o_sphere2500", 4), ("g2o_city10000", 3)):
        g = load_golden(nm)
        P = problem_of(g); P.set_x(np.ones(len(g["cw"]))); P.fiedler(want_vec=False)
        assert P.solve_mode()[0] == want, (nm, P.solve_mode())
        P.close()


def test_sweep_lanes_capture_graphs_while_other_threads_create_handles():
    """Round-4 advisor finding: the lanes' CU-masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags),
    so any legacy-stream call of the library -- round 4 still issued hipMemcpy / hipMemset in machip_create, machip_fiedler_csr
    and prepare_lanes -- serialises with every lane and, while a lane captures a chunk graph, fails with 'operation would make the
    legacy stream depend on a capturing blocking stream'.  Every copy now names the handle's own stream: a budget sweep (lanes
    capturing their graphs for the first time) runs while other threads create / destroy handles and call find_fiedler_pair;
    nothing may fail and every result equals the quiet run's."""
    import threading
    g = load_golden("g2o_intel")
    fixed, cand, n = edges_of(g, "f"), edges_of(g, "c"), int(g["n"])
    m = len(cand)
    ks = [int(p_ * m) for p_ in (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8)]
    inits = [NaiveGreedy(cand).subset(k) for k in ks]
    quiet = MAC(fixed, cand, n, fiedler_method="hip_lanczos").solve_sweep(ks, inits, max_iters=6)
    L5 = weight_graph_lap_from_edge_list([Edge(i, j, 1.0) for i in range(5) for j in range(i + 1, 5)], 5)
    errs, stop = [], threading.Event()

    def churn(kind):
        try:
            while not stop.is_set():
                if kind == 0:
                    P = problem_of(load_golden("er300_x0")); P.set_x(load_golden("er300_x0")["x"]); P.fiedler(want_vec=False); P.close()
                else:
                    lam = find_fiedler_pair(L5)[0]
                    assert abs(lam - 5.0) < 1e-6
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=churn, args=(i % 2,)) for i in range(3)]
    for t in th:
        t.start()
    try:
        mac = MAC(fixed, cand, n, fiedler_method="hip_lanczos")          # fresh handle: its lanes are created and capture inside the sweep
        noisy = mac.solve_sweep(ks, inits, max_iters=6)
    finally:
        stop.set()
        for t in th:
            t.join(timeout=120)
    assert not errs, errs
    for a, b in zip(noisy, quiet):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_er_size_budget_sweep_equals_fresh_sequential_solves():
    """bench.py --config c2s / c4s: the reference's budget sweep at ER size -- four budgets of the
    BASELINE.json configs[1] graph through machip_fw_sweep, chip-filling step kernels of the lanes interleaving on the GPU.
    Every budget's relaxed x, dual bound and lambda_2 trajectory must equal a FRESH handle running the same loop one budget at a
    time in the same solver mode (Lanczos pinned on both: the lanes never take other modes on this graph anyway), bit for bit,
    with 2 and with 4 lanes."""
    import bench
    w = bench.make_workload("c2")
    n, m, k = w["n"], len(w["cw"]), w["k"]
    ks = [max(1, int(round(k * (0.5 + b / 3)))) for b in range(4)]
    rng = np.random.default_rng(0)
    X0 = np.zeros((4, m))
    for b, kb in enumerate(ks):
        X0[b, rng.choice(m, kb, replace=False)] = 1.0
    start = reference_start_block(n)[:, 0].copy()
    seq = []
    for b in range(4):
        P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
        P.set_solver(1); P.set_start(start); P.set_x(X0[b])
        fs, u = [], np.inf
        for it in range(5):
            f, dual, gn = P.fw_step(ks[b], it)
            fs.append(f); u = min(u, dual)
            P.fw_commit()
        seq.append((np.array(fs), P.get_x(), u))
        P.close()
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_solver(1); P.set_start(start)
    for lanes in (2, 4):
        P.set_option("lanes", lanes)
        r = P.fw_sweep(ks, X0, max_iters=5, gap_tol=0.0, grad_tol=0.0, want_rounded=False)
        assert list(r["iters"]) == [5] * 4 and list(r["status"]) == [0] * 4
        for b in range(4):
            assert np.array_equal(r["f_traj"][b], seq[b][0]), (lanes, b)
            assert np.array_equal(r["x"][b], seq[b][1]) and r["upper"][b] == seq[b][2], (lanes, b)
    P.close()


@pytest.mark.parametrize("nm", ["intel", "sphere2500"])
def test_concurrent_budget_sweep_matches_the_reference_on_every_budget(nm):
    """MAC.solve_sweep against the REFERENCE's own budget sweep (tests/golden/g2o_sweep_<name>.npz, generated by running
    examples/g2o_experiment.py:306-336's loop -- NaiveGreedy init, MAC.solve(max_iters = 20), nearest rounding -- through the
    real reference for 10 .. 90 % of the loop closures): per budget the number of Frank-Wolfe iterations (the 60-90 %
    budgets of intel stop early on the duality gap), the lambda_2 trajectory to 1e-6 like the other trajectory tests, the dual
    bound, the relaxed x and the rounded selection."""
    g = load_golden("g2o_" + nm); gs = load_golden("g2o_sweep_" + nm)
    cand = edges_of(g, "c")
    mac = MAC(edges_of(g, "f"), cand, int(g["n"]))
    m = len(cand)
    ks = [int(k) for k in gs["ks"]]
    naive = NaiveGreedy(cand)
    res = mac.solve_sweep(ks, [naive.subset(k) for k in ks], max_iters=20)
    for j, k in enumerate(ks):
        ref = gs["f_traj"][j]
        nref = int(np.sum(~np.isnan(ref)))
        ft = mac.sweep_trace[j]
        assert int(np.sum(~np.isnan(ft))) == nref, (j, nref)
        assert np.allclose(ft[:nref], ref[:nref], rtol=1e-6), j
        rounded, w, u = res[j]
        assert abs(u - gs["upper"][j]) <= 1e-5 * gs["upper"][j]
        assert np.abs(w - gs["unrounded"][j]).max() <= 1e-6
        ref_rounded = np.unpackbits(gs["rounded_bits"][j])[:m].astype(np.float64)
        assert rounded.sum() == k == ref_rounded.sum()
        assert np.array_equal(rounded, ref_rounded) or \
            abs(mac.evaluate_objective(rounded) - mac.evaluate_objective(ref_rounded)) <= 1e-6 * mac.evaluate_objective(ref_rounded)


def test_bench_refuses_more_ranks_than_gpus():
    """bench.py --gpus N without a launcher spawns N ranks itself and must refuse -- loudly, non-zero -- when fewer than
    N devices are visible (round 1: `--gpus 8` silently ran and reported a 1-GPU job)."""
    ndev = _lib.device_count()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ndev + 1), "--config", "c3", "--steps", "2",
                        "--warmup", "0", "--no-cpu", "--no-pmc"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert '"metric"' not in r.stdout
    # and a launcher environment that disagrees with --gpus is an error too
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29555")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c3", "--steps", "2", "--warmup", "0",
                         "--no-cpu", "--no-pmc"], env=env2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


def test_bench_line_contract_on_a_small_config():
    """One JSON line with the contract's fields, `roofline` from the in-solve step timer and both CPU baselines."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5a", "--steps", "6", "--warmup", "1", "--min-seconds", "0.2",
                        "--no-pmc"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_strong", "repeats"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["vs_baseline"] is None and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches_timed"] > 0
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["host_cores"] >= 1 and d["cpu_baseline"]["kind"] == "port"
    if "cpu_parity_lambda2_rel" in d:
        assert d["cpu_parity_lambda2_rel"] < 1e-8
    else:       # (a host so slow that the bounded CPU leg did not finish one iteration: the line must say so)
        assert d["cpu_baseline"].get("upper_bound") is True


def test_graft_entry_smoke():
    import importlib
    importlib.import_module("__graft_entry__").smoke()


def test_trajectory_is_bit_reproducible_run_to_run():
    """Two runs of the same Frank-Wolfe trajectory (fresh handles, fused multi-workgroup path and single-workgroup path)
    give bit-identical lambda_2, step counts and iterates: no floating-point atomics, fixed reduction orders, a
    chunk scheduler that decides on step counts only."""
    for nm in ["er2000_solve", "g2o_intel", "g2o_city10000"]:
        g = load_golden(nm)
        runs = []
        for rep in range(2):
            P = problem_of(g)
            P.set_start(reference_start_block(int(g["n"]))[:, 0].copy())
            P.set_x(g["x_init"])
            fs, st = [], []
            for it in range(4):
                f, dual, gn = P.fw_step(int(g["k"]), it)
                fs.append((f, dual, gn)); st.append(int(P.stats.lanczos_steps))
                P.fw_commit()
            runs.append((np.array(fs), st, P.get_x()))
            P.close()
        assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1] and np.array_equal(runs[0][2], runs[1][2]), nm


def test_column_panel_step_is_bit_reproducible_run_to_run():
    """The panel form deals rows of equal length to tiles in ARRIVAL order (LDS atomics in k_pan_build), so two builds of the same
    matrix differ by a permutation among such rows.  Results must not: the product-sum of k_pan_mul rounds the same way at every
    chunk position (no contraction), which makes a row's sum independent of the slot it lands in.  Regression test for a last-digit
    run-to-run difference of lambda_2 found by tools/soak.sh: six solves of a dense configs[3] iterate (multi-round chunk ranges),
    panel step forced, must agree to the last bit -- lambda_2, the vector and the step count."""
    import bench
    w = bench.make_workload("c4")
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    P.set_x(w["x0"])
    for it in range(7):
        P.fw_step(w["k"], it); P.fw_commit()
    P.set_option("panel", 1)
    outs = []
    for rep in range(6):
        P.assemble()                                   # new solve, panel form rebuilt
        lam, v, _ = P.fiedler()
        outs.append((lam, v.copy(), int(P.stats.lanczos_steps)))
    for o in outs[1:]:
        assert o[0] == outs[0][0] and o[2] == outs[0][2] and np.array_equal(o[1], outs[0][1])
    P.close()


def test_shifted_panel_recurrence_reports_its_drift_factor_and_survives_a_tripped_monitor():
    """Round 6 (mac_amd/csrc/panel_u.h): the column-panel step multiplies the 8-byte vector u = t - sigma v and RECURS the product
    L v_j = (L u_{j-1} - (alpha_{j-1} - alpha_{j-2}) w_{j-1}) / beta_j instead of gathering v_j.  A rounding error of w is carried forward
    by |alpha_{j-1} - alpha_{j-2}| / beta_j per step; the host accumulates that factor from the tridiagonal records
    (machip_solve_stats.drift) and ends the sequence if it passes option panel_u_amp.  On a dense configs[3] iterate: (a) the factor stays
    tiny (alpha_j settles within a few steps) and lambda_2 / the step count equal the record form's (which gathers v_j itself: drift 0);
    (b) with the limit forced down to 2 the monitor trips at the first analysis point, the solve goes on in the two-kernel form from the
    Ritz vector it has (a restart) and still meets the reference's stop rule with the same lambda_2."""
    import bench
    w = bench.make_workload("c4")
    P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
    P.set_start(reference_start_block(w["n"])[:, 0].copy())
    P.set_x(w["x0"])
    for it in range(7):
        P.fw_step(w["k"], it); P.fw_commit()
    P.set_option("panel", 1)
    P.assemble()
    lam_u, v_u, _ = P.fiedler()
    st_u = P.stats.asdict()
    assert P.solve_mode()[0] == 2 and 1.0 <= st_u["drift"] < 1e3 and st_u["restarts"] == 0, st_u
    P.set_option("panel_u", 0)
    P.assemble()
    lam_r, v_r, _ = P.fiedler()
    st_r = P.stats.asdict()
    assert st_r["drift"] == 0.0 and abs(lam_u - lam_r) <= 1e-12 * lam_r and abs(st_u["lanczos_steps"] - st_r["lanczos_steps"]) <= 4, (lam_u, lam_r, st_u, st_r)
    assert np.abs(sign_align(v_u, v_r) - v_r).max() <= 1e-7
    P.set_option("panel_u", None)
    P.set_option("panel_u_amp", 2)
    P.assemble()
    lam_t, v_t, _ = P.fiedler()
    st_t = P.stats.asdict()
    assert st_t["drift"] > 2.0 and st_t["restarts"] >= 1 and st_t["residual"] < 1e-8, st_t
    assert abs(lam_t - lam_r) <= 1e-8 * lam_r and np.abs(sign_align(v_t, v_r) - v_r).max() <= 2e-6
    P.close()


def test_new_entry_points_reject_bad_arguments():
    """BAD_ARG (-> AssertionError, like the reference's asserts) instead of undefined behaviour on the round-2 entry points."""
    g = load_golden("er300_solve")
    P = problem_of(g)
    m = len(g["cw"])
    with pytest.raises(AssertionError):
        P.set_precision(2)
    # options read at creation are refused on an existing handle (they would silently do nothing: advisor finding on round 5) ...
    for nm in sorted(_lib.CREATION_OPTIONS):
        with pytest.raises(ValueError):
            P.set_option(nm, 8)
        P.set_option(nm, None)
    # ... and MAC(options=...) hands them to the constructor
    from mac_amd.solvers import MAC
    from mac_amd.utils.graphs import Edge
    fixed = [Edge(int(a), int(b), float(w_)) for a, b, w_ in zip(g["fi"], g["fj"], g["fw"])]
    cand = [Edge(int(a), int(b), float(w_)) for a, b, w_ in zip(g["ci"], g["cj"], g["cw"])]
    mac = MAC(fixed, cand, int(g["n"]), options={"vcap": 96, "asm_g": 8, "chunk": 6})
    assert mac._dev.get_option("vcap") == 96 and mac._dev.get_option("asm_g") == 8 and mac._dev.get_option("chunk") == 6
    with pytest.raises(AssertionError):
        _lib.shard_plan(10, 0, 0)
    with pytest.raises(AssertionError):
        _lib.shard_plan(10, 4, 4)
    lam, st = P.eval_batch(np.zeros((0, m)))                      # empty batch is fine
    assert lam.shape == (0,)
    with pytest.raises(AssertionError):
        P.eval_batch(np.zeros((2, m + 1)))                         # wrong width (caught by the binding)
    Q = problem_of(load_golden("er2000_solve"))
    with pytest.raises(AssertionError):
        _lib.comm_init_local([P, Q])                               # handles of different problems
    _lib.comm_init_local([P])                                      # a group of one is legal ...
    with pytest.raises(AssertionError):
        _lib.comm_init_local([P])                                  # ... joining twice is not
    P.set_x(g["x_init"])
    f, d, gn = P.fw_step(int(g["k"]), 0)                           # and a one-rank group behaves like no group
    assert abs(f - g["f_traj"][0]) <= LAM_RTOL * g["f_traj"][0]
    # more vectors than lanes, mixed precision lanes, order independence
    P2 = problem_of(load_golden("g2o_sphere2500")); g2 = load_golden("g2o_sphere2500")
    P2.set_start(reference_start_block(int(g2["n"]))[:, 0].copy())
    X = np.stack([g2["x_init"], g2["rounded"], g2["unrounded"]] * 7)           # 21 > 8 lanes
    lam64, st64 = P2.eval_batch(X)
    assert np.all(st64 == 0) and np.array_equal(lam64[:3], lam64[3:6]) and np.array_equal(lam64[:3], lam64[18:21])
    P2.set_precision(1)
    lam32, st32 = P2.eval_batch(X)
    assert np.all(st32 == 0) and np.allclose(lam32, lam64, rtol=1e-8)
    assert abs(lam64[0] - g2["lam_init"]) <= LAM_RTOL * g2["lam_init"]
    for h in (P, Q, P2):
        h.close()


# --------------------------------------------------------------------------------------------
# Inter-process communicator with a row-partitioned eigen-solve (round 4: machip_comm_init_ipc)
# --------------------------------------------------------------------------------------------
IPC_WORKER = r'''
import os, sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from mac_amd import _lib
from mac_amd.dist import FileGroup, attach_ipc, detach_ipc
from mac_amd.utils.fiedler import reference_start_block
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
wl, iters, die_at = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
for k_, v_ in json.loads(sys.argv[4] if len(sys.argv) > 4 else "{}").items():
    _lib.set_default_option(k_, v_)           # options of the handle created below (and of its solver), on every rank alike
if wl.startswith("golden:"):
    from conftest import load_golden
    g = load_golden(wl[7:])
    n, k, x0 = int(g["n"]), int(g["k"]), g["x_init"]
    P = _lib.Problem(n, g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"])
else:
    import bench
    w = bench.make_workload(wl)
    n, k, x0 = w["n"], w["k"], w["x0"]
    P = _lib.Problem(n, w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(n)[:, 0].copy())
dist = FileGroup(rank, world) if world > 1 else None
if dist is not None:
    attach_ipc(P, dist, rank, world, timeout_s=float(os.environ.get("IPC_TIMEOUT", "10")))
P.set_x(x0)
out, smodes, step_us = [], [], []
try:
    for it in range(iters):
        if rank == 1 and it == die_at:
            os._exit(17)                      # a rank dies mid-run: no goodbye, no abort flag
        f, dual, gn = P.fw_step(k, it)
        out.append((f.hex(), dual.hex(), gn.hex(), int(P.stats.lanczos_steps)))
        smodes.append(P.solve_mode()[0]); step_us.append(1e3 * P.stats.step_ms / max(1, P.stats.steps_timed))
        P.fw_commit()
    x = P.get_x()
    print("RESULT", json.dumps({"rank": rank, "out": out, "xsum": float(x.sum()).hex(), "xdot": float(x @ np.arange(len(x))).hex(),
                                "mode": int(_lib.load().machip_comm_mode(P._h)), "solver_modes": smodes, "step_us": step_us}), flush=True)
    if dist is not None:
        detach_ipc(P, dist)
    P.close()
    if dist is not None:
        dist.close()
except _lib.MachipError as e:
    print("ERROR", json.dumps({"rank": rank, "status": e.status, "msg": str(e), "done": len(out)}), flush=True)
    os._exit(3)
'''


def _run_ipc_job(world, wl, iters, die_at=-1, timeout=300, env_extra=None, opts=None):
    import json
    import uuid
    key = uuid.uuid4().hex
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MACHIP_RDZV_KEY=key, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, "-c", IPC_WORKER, wl, str(iters), str(die_at), json.dumps(opts or {})], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        lines = [l for l in so.splitlines() if l.startswith(("RESULT", "ERROR"))]
        res.append((p.returncode, lines[-1] if lines else "", se[-1500:]))
    return [(rc, (ln.split(" ", 1)[0], json.loads(ln.split(" ", 1)[1])) if ln else None, se) for rc, ln, se in res]


@pytest.mark.parametrize("world,wl,iters", [(2, "golden:er2000_solve", 5), (3, "golden:er2000_solve", 4), (2, "c2", 4), (2, "c4", 3)])
def test_ipc_row_partitioned_eigensolve_between_processes_is_bit_identical(world, wl, iters):
    """machip_comm_init_ipc: `world` PROCESSES on this one GPU (what bench.py --gpus N launches with MACHIP_SHARE_GPU=1), the
    peers' record / partial-sum / vector / gradient buffers mapped through hipIpcOpenMemHandle, every rank launching its share
    of each fused Lanczos step and ordering the steps by flag words in device memory -- no host in the loop, no RCCL (it refuses
    two ranks on one device: the gradient shards travel through the mapped buffers too).  f / dual / ||g|| / step counts of
    every iteration and the final x must equal the single-process run BIT FOR BIT on every rank."""
    single = _run_ipc_job(1, wl, iters)[0]
    assert single[0] == 0 and single[1][0] == "RESULT", single
    multi = _run_ipc_job(world, wl, iters)
    for rc, msg, se in multi:
        assert rc == 0 and msg and msg[0] == "RESULT", (rc, msg, se)
        assert msg[1]["out"] == single[1][1]["out"] and msg[1]["xsum"] == single[1][1]["xsum"] and msg[1]["xdot"] == single[1][1]["xdot"]
        assert msg[1]["mode"] == 5, msg[1]["mode"]              # the last eigen-solve really ran row-partitioned between the processes


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_row_partitioned_panel_step_is_bit_identical_to_one_rank(world):
    """VERDICT r5 item 2 (round 6, solver.h launch_chunk_ipc_pan): the row-partitioned eigen-solve between processes in COLUMN-PANEL form.  The
    shifted recurrence partitions by the row kernel's workgroups: a rank launches `k_pan_mul8` for the row blocks its rows lie in (a block
    that straddles two ranks is multiplied by both) and `k_pan_finu` for its workgroups, which write their rows of the next 8-byte operand and
    their six sums into every rank's copy (the operand lives in the IPC-mapped record buffers).  configs[3] with the panel step forced, five
    iterations, `world` processes on this one GPU: f / dual / ||g|| / step counts and the final x equal the one-process panel run bit for bit,
    every solve ran the panel step (mode 2) row-partitioned (comm mode 5).  (Round 5's library lost 2 of 30 such runs in GATHER form to a
    timing-dependent analysis point of the chunk feeder -- solver.h `more_coming` -- which this test family caught only now.)"""
    single = _run_ipc_job(1, "c4", 5, opts={"panel": 1})[0]
    assert single[0] == 0 and single[1][0] == "RESULT" and set(single[1][1]["solver_modes"]) == {2}, single
    multi = _run_ipc_job(world, "c4", 5, opts={"panel": 1})
    for rc, msg, se in multi:
        assert rc == 0 and msg and msg[0] == "RESULT", (rc, msg, se)
        assert msg[1]["out"] == single[1][1]["out"] and msg[1]["xsum"] == single[1][1]["xsum"] and msg[1]["xdot"] == single[1][1]["xdot"]
        assert msg[1]["mode"] == 5 and set(msg[1]["solver_modes"]) == {2}, msg[1]
        if world == 2:      # (measured 28-29 us: mul8 + finu + the publish / wait launch per step, two processes time-sharing the GPU; one rank alone: 13.5)
            assert max(msg[1]["step_us"][1:]) < 60.0, msg[1]["step_us"]      # (a bound on gross regressions only: two processes share one GPU with whatever else the box runs)
    print("panel step, us per Lanczos step: one rank", [round(t, 2) for t in single[1][1]["step_us"]], f"{world} ranks on one GPU", [round(t, 2) for t in multi[0][1][1]["step_us"]])


def test_ipc_ranks_agree_on_where_every_solve_ends_run_after_run():
    """Round 6 regression (solver.h, `more_coming`): the chunk feeder of a row-partitioned solve analysed the tridiagonal at the END OF THE
    QUEUE whenever the records up to there had landed before the loop came by -- short of the next analysis point, and only on the rank whose
    timing happened to be so; from there the ranks' analysis points differed, one went to the explicit check while the other waited for more
    steps, and both sat out the time limit: 2 of 30 two-rank configs[3] runs of round 5's library, 3 of 14 with the panel step.  Eight
    two-process runs of five iterations each (panel step forced) must all finish, with identical results."""
    first = None
    for rep in range(8):
        res = _run_ipc_job(2, "c4", 5, opts={"panel": 1}, env_extra={"IPC_TIMEOUT": "5"})
        for rc, msg, se in res:
            assert rc == 0 and msg and msg[0] == "RESULT", (rep, rc, msg, se)
        out = (res[0][1][1]["out"], res[0][1][1]["xsum"])
        assert (res[1][1][1]["out"], res[1][1][1]["xsum"]) == out
        first = first or out
        assert out == first, rep


@pytest.mark.parametrize("wl,opts", [("golden:er2000_solve", {"vcap": 80}), ("golden:g2o_kitti_05", {}), ("golden:g2o_city10000", {}),
                                     ("golden:er2000_solve", {"graph": 0, "chunk": 6})])
def test_ipc_communicator_with_restarts_and_unpartitioned_solver_modes(wl, opts):
    """The inter-process communicator when the eigen-solve is NOT one plain partitioned sequence: a basis so small that the
    sequence restarts (the restart continues in the classic two-kernel form, replicated on every rank), a pose graph whose solves
    run in the preconditioned mode (replicated; only the gradient is exchanged), city10000 (padded fixed-width step: replicated;
    then partitioned gather steps after a restart), eager launches with odd chunk lengths.  Two processes on one GPU, bit-identical
    to one process run with the same settings."""
    single = _run_ipc_job(1, wl, 3, opts=opts)[0]
    assert single[0] == 0 and single[1][0] == "RESULT", single
    for rc, msg, se in _run_ipc_job(2, wl, 3, opts=opts):
        assert rc == 0 and msg and msg[0] == "RESULT", (rc, msg, se)
        assert msg[1]["out"] == single[1][1]["out"] and msg[1]["xsum"] == single[1][1]["xsum"] and msg[1]["xdot"] == single[1][1]["xdot"]


def test_ipc_peers_of_a_dead_rank_return_an_error_within_the_time_limit():
    """A rank that dies mid-solve (os._exit: no goodbye) must not hang its peers: their next device-side wait times out after
    the communicator's limit (2 s here), later waits return at once, and the call comes back with MACHIP_RCCL_ERROR."""
    import time
    t0 = time.time()
    res = _run_ipc_job(2, "golden:er2000_solve", 6, die_at=2, timeout=120, env_extra={"IPC_TIMEOUT": "2"})
    el = time.time() - t0
    assert res[1][0] == 17                                   # the rank that died
    rc, msg, se = res[0]
    assert rc == 3 and msg and msg[0] == "ERROR", (rc, msg, se)
    assert msg[1]["status"] == _lib.RCCL_ERROR and msg[1]["done"] == 2 and "peer" in msg[1]["msg"]
    assert el < 60.0, el


def test_bench_two_ranks_emit_all_three_multi_gpu_legs_in_one_line():
    """ONE `bench.py --gpus N` invocation must yield all the multi-GPU evidence -- the candidate-shard pass
    (headline value, hipEvent-timed gradient and exchange), the replicas pass (per-rank figures) and the row-partitioned (IPC)
    eigen-solve pass (us per Lanczos step) -- and a leg that fails first contact is reported under `errors` while the others
    still print.  Two rank processes on this one GPU (MACHIP_SHARE_GPU=1: RCCL refuses two ranks on a device, so the shard
    leg exchanges its gradient through the IPC-mapped buffers with the eigen-solve replicated, comm_mode 6)."""
    import json
    import time
    t_start = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c2", "--steps", "4", "--warmup", "1",
                        "--min-seconds", "0.1", "--max-repeats", "2"], capture_output=True, text=True, cwd=ROOT, timeout=900,
                       env=dict({k_: v_ for k_, v_ in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")},
                                MACHIP_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    elapsed = time.time() - t_start
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    # (a multi-rank line carries no per-rank extras -- the PMC child passes, the CPU baselines and the same-node point
    # belong to the N = 1 line, rank 0 only, as the bench contract says -- and the whole run stays far from the 600 s RCCL watchdog)
    assert elapsed < 180.0, elapsed
    assert "cpu_baseline" not in d and d["roofline"]["traffic"] is None
    assert d["n_gpus"] == 2 and d["legs_run"] == ["shard", "replicas", "ipc_eig"] and isinstance(d["errors"], dict)
    for leg in ("shard", "replicas", "ipc_eig"):
        assert leg in d or leg in d["errors"], leg
    sh, rp, ip = d["shard"], d["replicas"], d["ipc_eig"]
    assert d["value"] == sh["value"] > 0 and d["scaling"] == "strong" and sh["comm_mode"] == 6
    assert sh["grad_us"] > 0 and sh["exchange_us"] > 0
    assert rp["scaling"] == "weak" and len(rp["per_rank"]) == 2 and rp["per_rank"][0]["K"] < rp["per_rank"][1]["K"] and rp["value"] > 0
    assert ip["comm_mode"] in (4, 5) and ip["us_per_lanczos_step"] > 0 and ip["bit_identical_to_shard_leg"] is True
    assert d["roofline"]["frac"] > 0 and d["config"]["workload"].startswith("configs[1]")


def test_bench_dry_run_checks_first_contact_on_one_gpu():
    """`bench.py --gpus 2 --dry` (MACHIP_SHARE_GPU=1: both ranks on this GPU): device visibility, peer-access row, IPC
    exchange and two Frank-Wolfe iterations of a tiny problem through the whole communicator stack, reported as one JSON line;
    the eigen-solve of the last iteration really ran row-partitioned (comm_mode 5) and both ranks hold the same lambda_2 bits."""
    import json
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry"], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MACHIP_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["dry"] and d["ok"] and d["results_equal_on_all_ranks"] and len(d["ranks"]) == 2
    for rk in d["ranks"]:
        assert rk["steps"]["ipc_exchange"] == "ok" and rk["steps"]["two_fw_iterations"] == "ok" and rk["steps"]["comm_mode"] == 5
        assert rk["peer_access_row"][rk["device"]] == 1


def test_exact_preconditioner_counts_fixed_off_chain_edges_as_closures():
    """ADVICE r3 (solver.h wb_alloc): the Woodbury buffers are sized from the ACTIVE CANDIDATES, but every off-chain entry of L(x) is a
    closure -- fixed edges off the chain included.  400 fixed closures + 40 candidates: the first extraction finds more closures than
    the buffers hold (256), the buffers must grow and the exact preconditioner must run (a handful of iterations), not be skipped
    silently for the tridiagonal one (hundreds on this weakly linked chain).  lambda_2 against SciPy's shift-invert Lanczos."""
    import scipy.sparse.linalg as spla
    n = 6000
    rng = np.random.default_rng(11)
    chain_i = np.arange(n - 1, dtype=np.int32)
    cw_chain = 10.0 ** rng.uniform(0, 2.5, n - 1)
    a = rng.integers(0, n, 400); b = np.clip(a + rng.integers(2, 3000, 400), 0, n - 1)
    keep = b - a > 1
    fa, fb = a[keep].astype(np.int32), b[keep].astype(np.int32)
    fi = np.r_[chain_i, fa]; fj = np.r_[chain_i + 1, fb]; fw = np.r_[cw_chain, 10.0 ** rng.uniform(0, 2, len(fa))]
    c1 = rng.integers(0, n - 50, 40).astype(np.int32); c2 = (c1 + rng.integers(2, 40, 40)).astype(np.int32)
    cw = rng.uniform(1.0, 50.0, 40)
    x = rng.uniform(0.3, 1.0, 40)
    P = _lib.Problem(n, fi, fj, fw, c1, c2, cw)
    P.set_x(x)
    P.set_solver(2)
    lam, v, _ = P.fiedler()
    its = int(P.stats.lanczos_steps)
    assert P.stats.residual < 1e-8 and its <= 40, its
    L = oracle.mac_laplacian(oracle.laplacian_from_edges(fi, fj, fw, n), c1.astype(np.int64), c2.astype(np.int64), cw, x, n)
    sh = 1e-6 * lam
    w = np.sort(spla.eigsh((L + sh * sp.identity(n)).tocsc(), k=2, sigma=0, which="LM", tol=0, return_eigenvectors=False)) - sh
    assert abs(lam - w[1]) <= 1e-7 * w[1], (lam, w[1])
    P.close()
