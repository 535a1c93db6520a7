"""find_fiedler_pair with the reference's signature (mac/utils/fiedler.py:9-44), computed
on the MI355X by libmachip (Lanczos on L restricted to 1-perp, stop rule = the
reference's ||Lv - lambda v||_1 / ||L||_inf < tol)."""
from __future__ import annotations

import numpy as np
from scipy.sparse import csr_matrix

from mac_amd import _lib

HIP_METHODS = ("hip", "hip_lanczos", "hip_lobpcg")
# method strings of the reference (fiedler.py:38-42 / nx:215-229): accepted so existing call
# sites run unchanged; the eigen-solve is still the HIP one.
REFERENCE_METHODS = ("tracemin_lu", "tracemin_cholesky", "tracemin_pcg")

try:  # the reference raises networkx.NetworkXError for unknown methods (nx:229)
    from networkx import NetworkXError as _BaseErr
except Exception:  # pragma: no cover
    _BaseErr = ValueError


class UnknownFiedlerMethod(_BaseErr):
    pass


def check_method(method):
    if method not in HIP_METHODS and method not in REFERENCE_METHODS:
        raise UnknownFiedlerMethod(f"Unknown linear system solver: {method}")


def solver_mode(method):
    """libmachip eigen-solver mode for a method string (machip_set_solver): 'hip' and the reference's
    direct-solver flavours pick automatically, 'hip_lanczos' forces the Lanczos path, and the
    reference's preconditioned flavour 'tracemin_pcg' (nx:22-76) -- like 'hip_lobpcg' -- asks for the
    preconditioned mode (LOBPCG + tridiagonal chain solve)."""
    check_method(method)
    return {"hip_lanczos": 1, "hip_lobpcg": 2, "tracemin_pcg": 2}.get(method, 0)


def reference_start_block(n, seed=None):
    """The reference's start block: RandomState(7).normal(size=(q, n)).T, q = min(4, n-1)
    (fiedler.py:27-32)."""
    if seed is None:
        seed = np.random.RandomState(7)
    q = min(4, n - 1)
    return np.asarray(seed.normal(size=(q, n))).T


def find_fiedler_pair(L, X=None, method="hip", tol=1e-8, seed=None):
    """Second-smallest eigenpair of the graph Laplacian ``L`` (any scipy sparse matrix).

    Returns ``(lambda_2, v_2, X)`` like the reference: X is n x q (q = min(4, n-1)), its
    column 0 is v_2 (unit norm, orthogonal to 1) -- the ONLY column that obeys the stop rule.  The
    other columns are the next Ritz vectors of the solve's last Krylov sequence (orthonormal,
    orthogonal to 1 and to v_2, Rayleigh quotients >= lambda_2): close to v_3, v_4, ... after a
    long sequence (1e-3 .. 1e-7 on er2000_x0), an arbitrary orthonormal completion after a short one
    or after the preconditioned modes, which keep no Krylov basis (include/machip.h, machip_fiedler;
    tests/test_gpu_parity.py::test_x_block_columns_are_what_the_header_says).  The reference's
    TraceMIN block converges q vectors together (fiedler.py:44); nothing on the hot path reads more
    than column 0 (mac.py:112-114).  ``X`` (if given) supplies the start vector in column 0.
    """
    check_method(method)
    L = csr_matrix(L, dtype=np.float64)
    n = L.shape[0]
    q = min(4, n - 1)
    if X is None:
        X = reference_start_block(n, seed)
    assert X.shape[0] == L.shape[0]      # fiedler.py:35-36
    assert X.shape[1] == q
    L.sum_duplicates()
    lam, v, Xo, _ = _lib.fiedler_csr(L.indptr, L.indices, L.data, n, tol=tol,
                                      x0=np.ascontiguousarray(X[:, 0]), q=q)
    Xo = np.asfortranarray(Xo)
    return lam, Xo[:, 0], Xo
