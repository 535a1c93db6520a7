"""Generic Frank-Wolfe driver with the reference's signature and semantics
(mac/optimization/frankwolfe.py:10-79).  ``problem`` and ``solve_lp`` are callables, so a
device-resident problem plugs in unchanged; ``MAC.solve`` uses the fused device loop
(machip_fw_step) which applies the same rules."""
import numpy as np


def naive_stepsize(k):
    return 2.0 / (k + 2.0)


def frank_wolfe(initial, problem, solve_lp, stepsize=None, maxiter=50,
                relative_duality_gap_tol=1e-5, grad_norm_tol=1e-10, verbose=False):
    if stepsize is None:
        stepsize = lambda x, g, s, k: naive_stepsize(k)  # noqa: E731
    x = initial
    u = float("inf")
    for i in range(maxiter):
        f, gradf = problem(x)
        s = solve_lp(gradf)
        u = min(u, f + gradf @ (s - x))
        if np.linalg.norm(gradf) < grad_norm_tol:
            if verbose:
                print("Gradient norm is approximately 0. Found optimal solution")
            return x, u
        if (u - f) < relative_duality_gap_tol * abs(f):
            if verbose:
                print("Duality gap tolerance reached, found optimal solution")
            return x, u
        x = x + stepsize(x, gradf, s, i) * (s - x)
    if verbose:
        print("Reached maximum number of iterations, returning best solution")
    return x, u
