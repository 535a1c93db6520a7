"""Pins oracle/ against golden vectors produced by the real reference
(tests/golden/make_golden.py) and the known answers of the reference's tests."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_golden, sign_align


@pytest.mark.parametrize("nm,exact", [("k5", 5.0), ("p2", 2.0), ("p3", 1.0),
                                      ("p50", 2 - 2 * np.cos(np.pi / 50)),
                                      ("c12", 2 - 2 * np.cos(2 * np.pi / 12)), ("star9", 1.0)])
def test_closed_forms(nm, exact):
    g = load_golden("fiedler_" + nm)
    L = oracle.laplacian_from_edges(g["ei"], g["ej"], g["ew"], int(g["n"]))
    lam, v, X = oracle.find_fiedler_pair(L)
    assert np.isclose(lam, exact, rtol=1e-9)          # tests/utils/test_fiedler.py:26-33 for k5
    assert abs(lam - g["lam"]) <= 1e-10 * max(1.0, abs(exact))
    assert X.shape == g["X"].shape
    assert abs(np.linalg.norm(v) - 1) < 1e-12 and abs(v.sum()) < 1e-8


def test_laplacian_builder_matches_reference():
    g = load_golden("laplacian_petersen_weighted")   # tests/utils/test_graphs.py:27-50
    L = oracle.laplacian_from_edges(g["ei"], g["ej"], g["ew"], 10)
    assert np.array_equal(L.toarray(), g["L_dense"])


@pytest.mark.parametrize("nm", ["petersen_x0", "er300_x0", "er300_xfrac", "er2000_x0", "er2000_xfrac"])
def test_fiedler_and_gradient(nm):
    g = load_golden(nm)
    n = int(g["n"])
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], n)
    L = mo.laplacian(g["x"]).tocsr(); L.sort_indices()
    assert np.array_equal(L.indptr, g["L_indptr"]) and np.array_equal(L.indices, g["L_indices"])
    assert np.allclose(L.data, g["L_data"], rtol=0, atol=1e-13 * np.abs(g["L_data"]).max())
    f, grad = mo.problem(g["x"])
    assert abs(f - g["lam"]) <= 1e-10 * abs(g["lam"])
    lam_d, v_d, _ = oracle.dense_fiedler(L)
    assert abs(lam_d - g["lam"]) <= 1e-9 * abs(g["lam"])
    _, v, X = oracle.find_fiedler_pair(L)
    assert np.abs(sign_align(v, g["v"]) - g["v"]).max() < 1e-6
    assert np.allclose(grad, g["grad"], rtol=1e-5, atol=1e-9 * np.abs(g["grad"]).max())
    # supergradient is bit-exact given the same vector
    assert np.array_equal(oracle.supergradient(g["v"], g["ci"], g["cj"], g["cw"]), g["grad"])


def test_petersen_fw_trajectory():
    g = load_golden("petersen_solve_k3")
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], 10)
    tr = []
    rounded, w, u = mo.solve(3, g["x_init"], max_iters=5, trace=tr)
    assert np.allclose([t[0] for t in tr], g["f_traj"], rtol=1e-9)
    assert np.allclose(w, g["unrounded"], atol=1e-12)
    assert np.array_equal(rounded, g["rounded"])
    assert abs(u - g["upper"]) < 1e-9
    assert abs(mo.evaluate_objective(np.zeros(6)) - g["lam_tree"]) < 1e-10
    assert abs(mo.evaluate_objective(np.ones(6)) - 2.0) < 1e-9


def test_petersen_sweep():
    rows = load_golden("petersen_sweep")["rows"]
    g = load_golden("petersen_solve_k3")
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], 10)
    for pct, k, l_init, l_un, l_r, up in rows:
        k = int(k)
        xi = np.zeros(6); xi[:k] = 1.0
        r, un, u = mo.solve(k, xi, max_iters=100)
        assert abs(mo.evaluate_objective(xi) - l_init) < 1e-9
        assert abs(mo.evaluate_objective(un) - l_un) < 1e-6
        assert abs(u - up) < 1e-6
        assert mo.evaluate_objective(un) >= l_init - 1e-12   # tests/solvers/test_mac.py:60


@pytest.mark.parametrize("nm", ["er300_solve", "er2000_solve"])
def test_er_fw_trajectory(nm):
    g = load_golden(nm)
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], int(g["n"]))
    tr = []
    rounded, w, u = mo.solve(int(g["k"]), g["x_init"], max_iters=int(g["max_iters"]), trace=tr)
    assert np.allclose([t[0] for t in tr], g["f_traj"], rtol=1e-8)
    assert np.array_equal([t[3] for t in tr], g["supp"])
    assert np.allclose(w, g["unrounded"], atol=1e-9)
    assert abs(u - g["upper"]) <= 1e-8 * abs(g["upper"])
    assert np.array_equal(rounded, g["rounded"])


@pytest.mark.parametrize("nm", ["intel", "sphere2500", "kitti_05", "city10000", "kitti_02", "ais2klinik"])
def test_pose_graph_goldens(nm):
    g = load_golden("g2o_" + nm)
    i, j, kap, n = oracle.parse_g2o_edges(os.path.join(GOLDEN, "data", nm + ".g2o"))
    fixed = oracle.split_chain_edges(i, j)
    assert n == int(g["n"])
    assert np.array_equal(i[fixed], g["fi"]) and np.array_equal(j[fixed], g["fj"])
    assert np.array_equal(i[~fixed], g["ci"]) and np.array_equal(j[~fixed], g["cj"])
    assert np.allclose(kap[fixed], g["fw"], rtol=1e-13) and np.allclose(kap[~fixed], g["cw"], rtol=1e-13)
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], n)
    k = int(g["k"])
    x0 = oracle.naive_greedy_subset(g["cw"], k)
    f0, g0 = mo.problem(x0)
    if np.array_equal(x0, g["x_init"]):     # argpartition ties (city/sphere have equal weights)
        assert abs(f0 - g["lam_init"]) <= 1e-9 * g["lam_init"]
    f0, g0 = mo.problem(g["x_init"])
    assert abs(f0 - g["lam_init"]) <= 1e-9 * g["lam_init"]
    assert np.allclose(g0, g["grad_init"], rtol=1e-4, atol=1e-8 * np.abs(g["grad_init"]).max())
    assert abs(mo.evaluate_objective(np.ones(len(g["cw"]))) - g["lam_all"]) <= 1e-9 * g["lam_all"]
    tr = []
    iters = 3 if nm in ("city10000", "ais2klinik") else 6        # (a city10000 iteration costs the oracle ~1.5 s)
    rounded, w, u = mo.solve(k, g["x_init"], max_iters=iters, trace=tr)
    assert np.allclose([t[0] for t in tr], g["f_traj"][:iters], rtol=1e-7)
    assert np.array_equal([t[3] for t in tr], g["supp"][:iters])
    assert np.array_equal(oracle.round_madow_base(g["unrounded"], k, float(g["madow_u"])), g["madow"])


def test_fw_toy_problems():
    g = load_golden("fw_toy")     # tests/optimization/test_frankwolfe.py:24-52
    box = lambda gr: (gr > 0).astype(float)
    x, u = oracle.frank_wolfe(np.ones(3) * 0.7, lambda z: (-float(z @ z), -2.0 * z), box, maxiter=200)
    assert np.allclose(x, g["x_box"], atol=1e-12) and np.allclose(x, 0, atol=1e-2)
    x2, u2 = oracle.frank_wolfe(np.array([1.0, 0.0]),
                                lambda z: (-float((z - 0.5) @ (z - 0.5)), -2.0 * (z - 0.5)),
                                lambda gr: oracle.solve_subset_box_lp(gr, 1), maxiter=300)
    assert np.allclose(x2, g["x_subset"], atol=1e-12)
    assert np.allclose(x2, [0.5, 0.5], atol=0.01)


def test_rounding():
    g = load_golden("rounding")
    assert np.array_equal(oracle.round_nearest(g["w"], int(g["k"]), g["weights"], 10), g["nearest_tb"])
    assert np.array_equal(oracle.round_madow_base(g["madow_in"], int(g["k"]), float(g["madow_u"])), g["madow"])
    assert oracle.round_nearest(np.arange(5.0), 0).sum() == 0


def test_city10000_reference_vertices():
    """The oracle reproduces the reference's LP vertices (top-K sets) on city10000 for the first iterations
    (tests/golden/city10000_vertices.npz: every vertex of the reference's 20-iteration run)."""
    g = load_golden("g2o_city10000"); gv = load_golden("city10000_vertices")
    assert np.array_equal(gv["x_init"], g["x_init"]) and np.allclose(gv["f_traj"], g["f_traj"], rtol=0, atol=0)
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], int(g["n"]))
    k = int(g["k"])
    x = g["x_init"].copy()
    for it in range(3):
        f, gr = mo.problem(x)
        s = oracle.solve_subset_box_lp(gr, k)
        assert abs(f - gv["f_traj"][it]) <= 1e-9 * abs(f)
        assert np.array_equal(np.nonzero(s)[0], gv["ref_s"][it])
        x = x + oracle.naive_stepsize(it) * (s - x)


def test_budget_sweep_goldens_early_stops():
    """tests/golden/g2o_sweep_intel.npz (the reference's budget sweep, make_golden.py g2o_sweep): the oracle's Frank-Wolfe stops
    where the reference's does -- the 70 / 80 / 90 % budgets end after 7 / 4 / 3 iterations on the duality-gap test
    (frankwolfe.py:70-74) -- with the same lambda_2 trajectory and dual bound."""
    g = load_golden("g2o_intel"); gs = load_golden("g2o_sweep_intel")
    mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], int(g["n"]))
    assert int(gs["m"]) == len(g["cw"]) and int(gs["ks"][1]) == int(g["k"])
    assert np.allclose(gs["f_traj"][1], g["f_traj"], rtol=1e-7) and abs(gs["upper"][1] - g["upper"]) <= 1e-9 * g["upper"]
    for j in (6, 7, 8):
        k = int(gs["ks"][j])
        trace = []
        rounded, w, u = mo.solve(k, oracle.naive_greedy_subset(g["cw"], k), max_iters=20, trace=trace)
        ref = gs["f_traj"][j]
        nref = int(np.sum(~np.isnan(ref)))
        assert len(trace) == nref and nref < 20
        assert np.allclose([t[0] for t in trace], ref[:nref], rtol=1e-7)
        assert abs(u - gs["upper"][j]) <= 1e-8 * gs["upper"][j]
        assert np.allclose(w, gs["unrounded"][j], atol=1e-9)
