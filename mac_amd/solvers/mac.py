"""MAC: maximise algebraic connectivity by Frank-Wolfe, same public surface as the
reference class (mac/solvers/mac.py:16-225), hot path on the MI355X.

Resident on the GPU for the lifetime of the object: the union sparsity pattern of
fixed + candidate edges, candidate endpoints/weights, x, the gradient, the Fiedler
vector and the Lanczos basis.  Per Frank-Wolfe iteration only three scalars
(f, dual bound, ||g||) cross PCIe.
"""
from __future__ import annotations

from dataclasses import dataclass
from timeit import default_timer as timer
from typing import Optional

import numpy as np
from scipy.sparse import csr_matrix

from mac_amd import _lib
from mac_amd.utils import fiedler as _fiedler
from mac_amd.utils.graphs import edges_to_arrays, weight_graph_lap_from_edge_list
from mac_amd.utils.rounding import round_madow, round_nearest


class MAC:
    @dataclass
    class Cache:
        """Warm-start slot (mac/solvers/mac.py:17-20).  In the reference the cache is
        written back with the old block (mac.py:126-127) and never takes effect; here
        ``use_cache=True`` warm-starts each eigen-solve from the previous Fiedler vector,
        which stays on the device.  ``Q`` mirrors the last Ritz block when requested."""
        Q: Optional[np.ndarray] = None

    def __init__(self, fixed_edges, candidate_edges, num_nodes, fiedler_method="hip",
                 fiedler_tol=1e-8, min_selection_weight_tol=1e-10, device=0, max_lanczos_steps=0, precision=0, options=None):
        """Arguments as mac/solvers/mac.py:22-24, plus (keyword-only in spirit, defaults keep the reference's call
        sites unchanged): ``device`` (GPU ordinal), ``max_lanczos_steps`` (0 = library default), ``precision``
        (0 = fp64 throughout; 1 = fp32 Krylov iterate + fp64 Rayleigh / residual refinement, BASELINE.json configs[4];
        the returned pairs obey the same stop rule either way) and ``options`` (dict: entries of the handle's option table,
        machip_set_option -- launch shapes / thresholds; nothing a reference call site needs)."""
        _fiedler.check_method(fiedler_method)
        num_edges = len(fixed_edges) + len(candidate_edges)
        assert (num_nodes - 1) <= num_edges                        # mac.py:47
        assert num_edges <= 0.5 * num_nodes * (num_nodes - 1)      # mac.py:52

        self.L_fixed = weight_graph_lap_from_edge_list(fixed_edges, num_nodes)   # mac.py:55
        self.num_nodes = num_nodes
        fi, fj, fw = edges_to_arrays(fixed_edges)
        ci, cj, cw = edges_to_arrays(candidate_edges)
        self.weights = cw                                            # mac.py:58-65
        self.edge_list = np.stack([ci.astype(np.int64), cj.astype(np.int64)], axis=1) if len(cw) else \
            np.zeros((0, 2), dtype=np.int64)
        self.fiedler_method = fiedler_method
        self.fiedler_tol = fiedler_tol
        self.min_selection_weight_tol = min_selection_weight_tol
        self.max_lanczos_steps = max_lanczos_steps
        # (options read when the handle is MADE -- assembly group width, basis budget -- go in as process defaults around its creation)
        options = dict(options or {})
        at_creation = {k: options.pop(k) for k in list(options) if k in _lib.CREATION_OPTIONS}
        with _lib.default_options(**at_creation):
            self._dev = _lib.Problem(num_nodes, fi, fj, fw, ci, cj, cw,
                                     min_selection_weight_tol=min_selection_weight_tol, device=device)
        # every cold eigen-solve starts from column 0 of the reference's block (fiedler.py:27-32)
        self._dev.set_start(_fiedler.reference_start_block(num_nodes)[:, 0].copy())
        self._dev.set_solver(_fiedler.solver_mode(fiedler_method))
        self.precision = int(precision)
        self._dev.set_precision(self.precision)
        if options:
            self._dev.set_options(**options)
        self.last_stats = None

    # -------------------------------------------------------------------------------
    def laplacian(self, x):
        """L(x) as scipy CSR, assembled on the device (mac.py:74-89)."""
        self._dev.set_x(np.asarray(x, dtype=np.float64))
        indptr, indices, data = self._dev.laplacian_csr()
        L = csr_matrix((data, indices, indptr), shape=(self.num_nodes, self.num_nodes))
        L.sum_duplicates()
        L.sort_indices()
        return L

    def evaluate_objective(self, x):
        """lambda_2(L(x)) with the configured tolerance (mac.py:91-102)."""
        self._dev.set_x(np.asarray(x, dtype=np.float64))
        lam, _, _ = self._dev.fiedler(tol=self.fiedler_tol, max_steps=self.max_lanczos_steps,
                                      want_vec=False)
        self.last_stats = self._dev.stats.asdict()
        return lam

    def evaluate_objective_batch(self, X):
        """evaluate_objective for every row of X (B x m) in one call: the solves run concurrently on the device
        (machip_eval_batch).  What round_madow(max_iters > 1) and budget sweeps call in a loop in the reference."""
        lam, st = self._dev.eval_batch(np.asarray(X, dtype=np.float64), tol=self.fiedler_tol, max_steps=self.max_lanczos_steps)
        self.last_stats = None           # (statistics belong to single solves)
        if np.any(st == _lib.NOT_CONVERGED):
            raise _lib.NotConverged(_lib.NOT_CONVERGED, "an eigen-solve of the batch hit the step cap")
        if np.any(st == _lib.DISCONNECTED):
            # same behaviour as evaluate_objective on the same x (and as the reference, whose sparse LU raises "Factor is
            # exactly singular" on a disconnected selection): the loop the batch replaces would have raised here
            b = int(np.nonzero(st == _lib.DISCONNECTED)[0][0])
            raise _lib.Disconnected(_lib.DISCONNECTED, f"entry {b} of the batch: lambda_2 ~ 0, the graph is not connected")
        return lam

    def problem(self, x, cache=None):
        """(lambda_2(L(x)), supergradient) (mac.py:104-128).  The reference always solves
        with tol 1e-8 here (mac.py:115); so does this."""
        self._dev.set_x(np.asarray(x, dtype=np.float64))
        warm = cache is not None and cache.Q is not None
        f, v, _ = self._dev.fiedler(tol=1e-8, max_steps=self.max_lanczos_steps,
                                    warm_start=warm, want_vec=cache is not None)
        self.last_stats = self._dev.stats.asdict()
        gradf = self._dev.gradient()
        if cache is not None:
            # the reference's slot holds an n x q ndarray (mac/solvers/mac.py:17-20); here: the converged Fiedler vector
            # as an n x 1 block.  The copy the next solve warm-starts from stays on the device.
            cache.Q = v.reshape(-1, 1)
        return f, gradf

    def solve_sweep(self, ks, x_inits, rounding="nearest", max_iters=5, relative_duality_gap_tol=1e-4, grad_norm_tol=1e-8,
                    use_cache=False, seed=None):
        """``solve`` for several budgets of this graph at once: the loop of examples/g2o_experiment.py:306-336
        (``for pct: MAC.solve(k, w_init, ...)``) run concurrently on the device (machip_fw_sweep: one evaluation lane per
        budget, up to 16 at a time -- option "lanes").  ``ks`` budgets, ``x_inits`` the matching initial selections.  Returns a
        list of ``(rounded, unrounded, upper)`` in the order of ``ks``.  Each obeys the same stop rules from the same start
        vector as ``solve`` for that budget on a fresh MAC object.  The two are BIT-IDENTICAL when the eigen-solver mode is pinned
        (``fiedler_method="hip_lanczos"``: same kernels on a lane and on the handle -- unless an eigen-solve needs more Lanczos
        steps in one sequence than a lane's share of the basis memory holds).  Under the automatic mode (``"hip"``) they agree
        to the solver tolerance only: a standalone handle may take the exact chain + closures mode (small pose graphs up to 700
        active closures, larger ones on a long forecast), which a lane keeps to 256 closures so as not to starve the other
        lanes -- a 1e-8 difference in lambda_2 can move a near-tie of the top-k LP and with it the rounded selection
        (include/machip.h, machip_fw_sweep).  rounding: "nearest" (device, tie-broken by edge weight like mac.py:209) or
        "madow" (host, one draw per budget from ``seed``)."""
        m = len(self.weights)
        ks = [int(k) for k in ks]
        assert len(ks) == len(x_inits)
        out = [None] * len(ks)
        idx = [i for i, k in enumerate(ks) if k < m]
        for i, k in enumerate(ks):
            if k >= m:                                               # mac.py:173-180
                ones = np.ones(m)
                out[i] = (ones, ones, self.evaluate_objective(ones))
        if idx:
            X0 = np.stack([np.asarray(x_inits[i], dtype=np.float64) for i in idx])
            assert X0.shape[1] == m                                  # mac.py:183
            r = self._dev.fw_sweep([ks[i] for i in idx], X0, max_iters=max_iters, gap_tol=relative_duality_gap_tol,
                                   grad_tol=grad_norm_tol, tol=1e-8, max_steps=self.max_lanczos_steps, warm_start=use_cache,
                                   round_decimals=10, want_rounded=(rounding != "madow"))
            for j, i in enumerate(idx):
                if r["status"][j] != _lib.OK:
                    _lib.check(int(r["status"][j]))
                w = r["x"][j]
                rounded = round_madow(w, ks[i], seed=seed) if rounding == "madow" else r["rounded"][j]
                out[i] = (rounded, w, float(r["upper"][j]))
            self.sweep_trace = r["f_traj"]
        return out

    def solve(self, k, x_init=None, rounding="nearest", fallback=False, max_iters=5,
              relative_duality_gap_tol=1e-4, grad_norm_tol=1e-8, random_rounding_max_iters=1,
              verbose=False, return_rounding_time=False, use_cache=False):
        """Frank-Wolfe on the relaxation, then rounding (mac.py:130-225); returns
        ``(rounded, unrounded, upper_bound[, rounding_time])``."""
        m = len(self.weights)
        if k >= m:                                                   # mac.py:173-180
            result = np.ones(m)
            val = self.evaluate_objective(result)
            if return_rounding_time:
                return result, result, val, 0.0
            return result, result, val

        assert len(x_init) == m                                       # mac.py:183
        dev = self._dev
        dev.set_x(np.asarray(x_init, dtype=np.float64))
        # frankwolfe.py:53-76 as mac.py:196-200 calls it, run on the C side (machip_fw_run: no return to Python between iterations)
        r = dev.fw_run(k, max_iters, gap_tol=relative_duality_gap_tol, grad_tol=grad_norm_tol, tol=1e-8,
                       max_steps=self.max_lanczos_steps, warm_start=bool(use_cache))
        u = float(r["upper"])
        self.trace = []
        ub = float("inf")
        for i in range(r["iters"]):
            ub = min(ub, float(r["dual"][i]))
            st = r["stats"][i]
            self.trace.append((float(r["f"][i]), ub, float(r["gnorm"][i]), int(st.support), int(st.lanczos_steps)))
            if verbose:
                print(f"[mac_amd] it {i}: f={r['f'][i]:.12g} u={ub:.12g} |g|={r['gnorm'][i]:.3g} "
                      f"supp={st.support} lanczos={st.lanczos_steps}")
        w = dev.get_x()

        start = timer()
        if rounding == "madow":
            rounded = round_madow(w, k, value_fn=self.evaluate_objective, max_iters=random_rounding_max_iters,
                                  batch_value_fn=self.evaluate_objective_batch)
        else:
            # rounding == "nearest" (mac.py:209), on the device-resident x (machip_round_nearest)
            rounded = dev.round_nearest(k, decimals=10)
        rounding_time = timer() - start

        if fallback:
            # The reference references an undefined name here (mac.py:218, NameError); the
            # evident intent -- keep the initial point if rounding made it worse -- is implemented.
            if self.evaluate_objective(rounded) < self.evaluate_objective(x_init):
                rounded = np.asarray(x_init, dtype=np.float64)

        if return_rounding_time:
            return rounded, w, u, rounding_time
        return rounded, w, u
