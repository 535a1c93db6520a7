"""Projection-free ascent for concave maximisation over a compact convex set.

Same call signature and semantics as the reference driver (mac/optimization/frankwolfe.py:10-79):
``problem(x) -> (value, supergradient)``, ``solve_lp(g) -> argmax_{s in C} <g, s>``, open-loop step
2/(k+2) unless ``stepsize`` is given.  Both callables are plug-in points, so a device-resident
problem slots in unchanged; ``MAC.solve`` itself runs the fused device loop (machip_fw_step),
which applies exactly these rules.
"""
import numpy as np


def naive_stepsize(k):
    """Open-loop Frank-Wolfe step for iteration k = 0, 1, ... (the first step jumps to the LP vertex)."""
    return 2.0 / (k + 2.0)


def _report(verbose, text):
    if verbose:
        print(text)


def frank_wolfe(initial, problem, solve_lp, stepsize=None, maxiter=50,
                relative_duality_gap_tol=1e-5, grad_norm_tol=1e-10, verbose=False):
    """Returns ``(x, upper)``: the last iterate and the tightest dual bound
    ``min_k f(x_k) + <g_k, s_k - x_k>`` seen.  When a stop test fires, the iterate returned is the
    one the test was evaluated at (not yet moved), as in the reference."""
    step_rule = stepsize if stepsize is not None else (lambda _x, _g, _s, it: naive_stepsize(it))
    iterate, upper = initial, float("inf")
    for it in range(maxiter):
        value, grad = problem(iterate)
        vertex = solve_lp(grad)
        upper = min(upper, value + grad @ (vertex - iterate))      # bound from the pre-update point
        if np.linalg.norm(grad) < grad_norm_tol:
            _report(verbose, f"frank_wolfe: |g| below {grad_norm_tol:g} at iteration {it}, stationary point")
            return iterate, upper
        if (upper - value) < relative_duality_gap_tol * abs(value):
            _report(verbose, f"frank_wolfe: relative duality gap below {relative_duality_gap_tol:g} at iteration {it}")
            return iterate, upper
        iterate = iterate + step_rule(iterate, grad, vertex, it) * (vertex - iterate)
    _report(verbose, f"frank_wolfe: iteration budget {maxiter} exhausted")
    return iterate, upper
