"""NumPy/SciPy restatement of the reference's Frank-Wolfe / Fiedler hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written from the maths of
the reference, vectorised, no reference source copied.  Every function names
the reference lines it restates; paths are relative to /root/reference, and
``nx:`` means networkx 3.4.2 ``networkx/linalg/algebraicconnectivity.py`` (the
un-vendored third-party module the reference delegates the eigen-solve to at
mac/utils/fiedler.py:42; requirements.txt:2 leaves it unpinned, 3.4.2 is what
the image ships).  The sparse LU below is scipy 1.15.3 ``splu`` (SuperLU), the
same third-party arithmetic nx:90-98 calls.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spla

__all__ = [
    "laplacian_from_edges", "mac_laplacian", "tracemin_fiedler",
    "find_fiedler_pair", "supergradient", "solve_subset_box_lp",
    "naive_stepsize", "frank_wolfe", "round_nearest", "round_madow_base",
    "naive_greedy_subset", "MacOracle", "dense_fiedler", "eigsh_fiedler", "split_chain_edges",
    "parse_g2o_edges",
]


# --------------------------------------------------------------------------
# Laplacian assembly
# --------------------------------------------------------------------------
def laplacian_from_edges(ei, ej, ew, n):
    """L = sum_e w_e (e_i - e_j)(e_i - e_j)^T as CSR, duplicates summed.

    Restates mac/utils/graphs.py:13-48 and :58-98 (four COO triplets per edge:
    (i,i,+w), (j,j,+w), (i,j,-w), (j,i,-w), then csr_matrix(coo_matrix(...))).
    """
    ei = np.asarray(ei, dtype=np.int64).ravel()
    ej = np.asarray(ej, dtype=np.int64).ravel()
    ew = np.asarray(ew, dtype=np.float64).ravel()
    rows = np.concatenate([ei, ej, ei, ej])
    cols = np.concatenate([ei, ej, ej, ei])
    data = np.concatenate([ew, ew, -ew, -ew])
    return sp.csr_matrix(sp.coo_matrix((data, (rows, cols)), shape=(n, n)))


def mac_laplacian(L_fixed, ci, cj, cw, x, n, min_selection_weight_tol=1e-10):
    """L(x) = L_fixed + sum_{k: x_k > tol} x_k w_k L_k  (mac/solvers/mac.py:74-89)."""
    x = np.asarray(x, dtype=np.float64)
    idx = np.where(x > min_selection_weight_tol)[0]
    prod = x[idx] * cw[idx]
    return L_fixed + laplacian_from_edges(ci[idx], cj[idx], prod, n)


# --------------------------------------------------------------------------
# TraceMIN-Fiedler (the reference's eigen-solver)
# --------------------------------------------------------------------------
def tracemin_fiedler(L, X, tol=1e-8, max_outer=100000):
    """TraceMIN-Fiedler with a grounded sparse LU, un-normalised Laplacian.

    Restates nx:151-256 for ``normalized=False, method='tracemin_lu'``:
      project (nx:209-213), ground the densest column and factor (nx:220-227,
      nx:90-98), Lnorm = ||L||_inf (nx:232), then loop (nx:236-254):
      thin QR, W = L X, H = X^T W, eigh, Ritz vectors, stop when
      ||W y0 - s0 x0||_1 / Lnorm < tol, else X <- (inv(W^T X) W^T)^T with
      W = A^{-1} X, project.
    Returns (sigma ascending [q], X [n,q], n_outer).
    """
    L = sp.csr_matrix(L, dtype=np.float64)
    n = L.shape[0]
    X = np.array(X, dtype=np.float64, copy=True)
    q = X.shape[1]

    def project(Z):
        Z -= Z.sum(axis=0, keepdims=True) / n

    A = sp.csc_matrix(L, dtype=np.float64, copy=True)
    g = int(np.diff(A.indptr).argmax())
    A = A.tolil()
    A[g, g] = np.inf
    A = sp.csc_matrix(A)
    lu = spla.splu(A, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                   options={"Equil": True, "SymmetricMode": True})

    Lnorm = abs(L).sum(axis=1).max()
    project(X)
    n_outer = 0
    while True:
        n_outer += 1
        X = np.linalg.qr(X)[0]
        W = np.asarray(L @ X)
        H = X.T @ W
        sigma, Y = scipy.linalg.eigh(H)
        X = X @ Y
        res = np.abs(W @ Y[:, 0] - sigma[0] * X[:, 0]).sum() / Lnorm
        if res < tol or n_outer >= max_outer:
            break
        W = np.empty_like(X, order="F")
        for c in range(q):
            W[:, c] = lu.solve(np.ascontiguousarray(X[:, c]))
        X = (scipy.linalg.inv(W.T @ X) @ W.T).T
        X = np.array(X)
        project(X)
    return sigma, np.asarray(X), n_outer


def find_fiedler_pair(L, X=None, tol=1e-8, seed=None):
    """(lambda_2, v_2, X) as mac/utils/fiedler.py:9-44 returns them.

    The start block is RandomState(7).normal(size=(q, n)).T with
    q = min(4, n-1) (fiedler.py:27-32), which makes the result deterministic.
    """
    n = L.shape[0]
    if seed is None:
        seed = np.random.RandomState(7)
    q = min(4, n - 1)
    if X is None:
        X = np.asarray(seed.normal(size=(q, n))).T
    assert X.shape[0] == n and X.shape[1] == q
    sigma, X, _ = tracemin_fiedler(L, X, tol=tol)
    return sigma[0], X[:, 0], X


def eigsh_fiedler(L, tol=1e-10):
    """NOT the reference's solver: SciPy's ARPACK Lanczos for the two smallest eigenvalues of L
    (0 and lambda_2), the "strong CPU baseline" of SURVEY section 8(d).  Returns (lambda_2, v_2)."""
    n = L.shape[0]
    v0 = np.random.RandomState(7).normal(size=n)
    w, V = spla.eigsh(sp.csr_matrix(L, dtype=np.float64), k=2, which="SA", tol=tol, v0=v0, ncv=min(n - 1, 64))
    o = np.argsort(w)
    return float(w[o[1]]), V[:, o[1]]


def dense_fiedler(L):
    """Independent cross-check (not the reference): dense eigh, returns
    (lambda_2, v_2, all eigenvalues)."""
    w, V = np.linalg.eigh(np.asarray(sp.csr_matrix(L).todense()))
    return w[1], V[:, 1], w


# --------------------------------------------------------------------------
# Supergradient, LP oracle, Frank-Wolfe driver
# --------------------------------------------------------------------------
def supergradient(v, ci, cj, cw):
    """g_k = (w_k (v_i - v_j)) (v_i - v_j) for ALL candidates, in exactly the
    operation order of mac/solvers/mac.py:117-124 (bit-exact with it)."""
    d = v[ci] - v[cj]
    return (cw * d) * d


def solve_subset_box_lp(g, k):
    """Indicator of the k largest entries of g (constraints.py:12-22 ->
    rounding.py:21-28, argpartition; ties arbitrary)."""
    s = np.zeros(len(g))
    if k > 0:
        s[np.argpartition(g, -k)[-k:]] = 1.0
    return s


def naive_stepsize(i):
    """2/(i+2)  (frankwolfe.py:7-8)."""
    return 2.0 / (i + 2.0)


def frank_wolfe(initial, problem, solve_lp, maxiter=50,
                relative_duality_gap_tol=1e-5, grad_norm_tol=1e-10, trace=None):
    """Frank-Wolfe ascent, restating frankwolfe.py:10-79 (dual bound from the
    pre-update x at :62, stop tests :65-74, update :76).  ``trace`` (a list) if
    given receives (f, u, ||g||, |supp x|) per iteration."""
    x = initial
    u = float("inf")
    for i in range(maxiter):
        f, g = problem(x)
        s = solve_lp(g)
        u = min(u, f + g @ (s - x))
        gn = np.linalg.norm(g)
        if trace is not None:
            trace.append((float(f), float(u), float(gn), int(np.count_nonzero(x > 1e-10))))
        if gn < grad_norm_tol:
            return x, u
        if (u - f) < relative_duality_gap_tol * abs(f):
            return x, u
        x = x + naive_stepsize(i) * (s - x)
    return x, u


# --------------------------------------------------------------------------
# Rounding (post-loop)
# --------------------------------------------------------------------------
def round_nearest(w, k, weights=None, break_ties_decimal_tol=None):
    """rounding.py:7-42.  Without tie-breakers: top-k indicator.  With them:
    round w to ``break_ties_decimal_tol`` decimals and take the top k under the
    lexicographic order (rounded w, edge weight)."""
    w = np.asarray(w, dtype=np.float64)
    out = np.zeros(len(w))
    if k <= 0:
        return out
    if weights is None or break_ties_decimal_tol is None:
        out[np.argpartition(w, -k)[-k:]] = 1.0
        return out
    tw = w.round(decimals=break_ties_decimal_tol)
    order = np.lexsort((np.asarray(weights, dtype=np.float64), tw))  # last key primary
    out[order[-k:]] = 1.0
    return out


def round_madow_base(w, k, u):
    """Systematic (Madow) sampling with offset u in [0,1) (rounding.py:78-95):
    pick index t for each i in 0..k-1 with cumsum_excl[t] <= u+i < cumsum[t]."""
    w = np.asarray(w, dtype=np.float64)
    sumw = np.cumsum(w)
    pi = np.concatenate([[0.0], sumw[:-1]])
    x = np.zeros(len(w))
    for i in range(k):
        t = u + i
        x[(pi <= t) & (t < sumw)] = 1.0
    return x


def naive_greedy_subset(weights, k):
    """Top-k by edge weight (mac/solvers/baseline.py:7-14), used as x_init."""
    return solve_subset_box_lp(np.asarray(weights, dtype=np.float64), k)


# --------------------------------------------------------------------------
# The MAC object (hot-path methods only)
# --------------------------------------------------------------------------
class MacOracle:
    """Hot-path subset of mac.solvers.MAC (mac/solvers/mac.py:16-225) on flat
    arrays: fixed (fi,fj,fw), candidates (ci,cj,cw), n nodes."""

    def __init__(self, fi, fj, fw, ci, cj, cw, n, fiedler_tol=1e-8,
                 min_selection_weight_tol=1e-10, fiedler="tracemin"):
        assert fiedler in ("tracemin", "eigsh")
        self.fiedler = fiedler
        self.n = int(n)
        self.ci = np.asarray(ci, dtype=np.int64)
        self.cj = np.asarray(cj, dtype=np.int64)
        self.cw = np.asarray(cw, dtype=np.float64)
        self.L_fixed = laplacian_from_edges(fi, fj, fw, n)
        self.tol = fiedler_tol
        self.min_sel = min_selection_weight_tol

    def laplacian(self, x):
        return mac_laplacian(self.L_fixed, self.ci, self.cj, self.cw, x, self.n, self.min_sel)

    def evaluate_objective(self, x):
        return find_fiedler_pair(self.laplacian(x), tol=self.tol)[0]

    def problem(self, x):
        """(lambda_2, supergradient); mac.py:104-128 (always tracemin_lu, tol
        1e-8, cold start: SURVEY section 0 items 1-2)."""
        if self.fiedler == "eigsh":     # bench.py's strong CPU baseline only
            f, v = eigsh_fiedler(self.laplacian(x))
        else:
            f, v, _ = find_fiedler_pair(self.laplacian(x))
        return f, supergradient(v, self.ci, self.cj, self.cw)

    def solve(self, k, x_init, max_iters=5, relative_duality_gap_tol=1e-4,
              grad_norm_tol=1e-8, rounding="nearest", trace=None, madow_u=None):
        """mac.py:130-225 without the (broken) fallback branch."""
        m = len(self.cw)
        if k >= m:
            r = np.ones(m)
            return r, r, self.evaluate_objective(r)
        w, u = frank_wolfe(np.asarray(x_init, dtype=np.float64), self.problem,
                           lambda g: solve_subset_box_lp(g, k), maxiter=max_iters,
                           relative_duality_gap_tol=relative_duality_gap_tol,
                           grad_norm_tol=grad_norm_tol, trace=trace)
        if rounding == "madow":
            rounded = round_madow_base(w, k, madow_u)
        else:
            rounded = round_nearest(w, k, weights=self.cw, break_ties_decimal_tol=10)
        return rounded, w, u


# --------------------------------------------------------------------------
# g2o ingestion
# --------------------------------------------------------------------------
def parse_g2o_edges(path):
    """(i, j, kappa, num_poses) from a .g2o file, restating
    examples/pose_graph_utils.py:228-351 + :381-396: EDGE_SE2 -> kappa = I33;
    EDGE_SE3:QUAT -> kappa = 3 / (2 tr(inv(I[3:6,3:6]))); num_poses = max id + 1."""
    I, J, K = [], [], []
    with open(path, "r") as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "EDGE_SE2":
                v = [float(t) for t in tok[1:12]]
                I.append(int(v[0])); J.append(int(v[1])); K.append(v[10])
            elif tok[0] == "EDGE_SE3:QUAT":
                v = [float(t) for t in tok[1:31]]
                # upper-triangular 6x6 information, row-major: v[9:30]
                info = np.zeros((6, 6))
                info[np.triu_indices(6)] = v[9:30]
                info = info + np.triu(info, 1).T
                kappa = 3.0 / (2.0 * np.trace(np.linalg.inv(info[3:6, 3:6])))
                I.append(int(v[0])); J.append(int(v[1])); K.append(kappa)
    I = np.asarray(I, dtype=np.int64); J = np.asarray(J, dtype=np.int64)
    n = int(max(I.max(), J.max())) + 1 if len(I) else 0
    return I, J, np.asarray(K, dtype=np.float64), n


def split_chain_edges(i, j):
    """Boolean mask of 'fixed' (odometry) edges: |i-j| <= 1
    (examples/pose_graph_utils.py:18-45)."""
    return np.abs(np.asarray(j) - np.asarray(i)) <= 1
