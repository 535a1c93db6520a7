"""Rounding onto {0,1}^m with |w| = k (mac/utils/rounding.py).  Post-loop, once per solve:
host NumPy (vectorised); SURVEY section 8(f) rank 2."""
from __future__ import annotations

import numpy as np


def round_nearest(w, k, weights=None, break_ties_decimal_tol=None):
    """mac/utils/rounding.py:7-42.  Plain top-k, or -- with ``weights`` and a decimal
    tolerance -- top-k under the lexicographic key (round(w, tol), weight)."""
    w = np.asarray(w, dtype=np.float64)
    rounded = np.zeros(len(w))
    if k <= 0:
        return rounded
    if weights is None or break_ties_decimal_tol is None:
        rounded[np.argpartition(w, -k)[-k:]] = 1.0
        return rounded
    # Top-k under the lexicographic key (round(w, tol), weight), as the reference's structured
    # argpartition (rounding.py:33-38).  O(m): one partition for the k-th rounded value, then only
    # the entries tied with it are ordered by weight (stable, so equal weights keep index order --
    # the same set np.lexsort((weights, tw))[-k:] selects).
    tw = w.round(decimals=break_ties_decimal_tol)
    m = len(tw)
    if k >= m:
        rounded[:] = 1.0
        return rounded
    thr = np.partition(tw, m - k)[m - k]
    above = tw > thr
    rounded[above] = 1.0
    need = k - int(above.sum())
    if need > 0:
        ties = np.nonzero(tw == thr)[0]
        if need < len(ties):
            wt = np.asarray(weights, dtype=np.float64)[ties]
            ties = ties[np.argsort(wt, kind="stable")[-need:]]
        rounded[ties] = 1.0
    return rounded


def round_random(w, k):
    """Independent Bernoulli(w_i) rounding (mac/utils/rounding.py:44-61)."""
    w = np.asarray(w, dtype=np.float64)
    return (w > np.random.rand(len(w))).astype(np.float64)


def round_madow_base(w, k, seed=None):
    """Madow systematic sampling (mac/utils/rounding.py:78-95): one uniform offset u, pick
    the index whose cumulative-weight interval contains u + i for i = 0..k-1."""
    u = np.random.rand() if seed is None else seed.rand()
    w = np.asarray(w, dtype=np.float64)
    sumw = np.cumsum(w)
    pi = np.concatenate([[0.0], sumw[:-1]])
    x = np.zeros(len(w))
    targets = u + np.arange(k)
    idx = np.searchsorted(sumw, targets, side="right")      # first t with targets < sumw[t]
    ok = idx < len(w)
    idx = idx[ok]
    hit = pi[idx] <= targets[ok]
    x[idx[hit]] = 1.0
    assert np.sum(x) == k, f"Error: {np.sum(x)} != {k}"
    return x


def round_madow(w, k, seed=None, value_fn=None, max_iters=1, batch_value_fn=None):
    """mac/utils/rounding.py:63-75: best of ``max_iters`` Madow draws under ``value_fn``.

    ``batch_value_fn`` (extension): a function of a (B, m) array returning B values; the draws are then generated
    first -- the random stream is consumed exactly as in the sequential loop, one uniform per draw, ``value_fn``
    never touches it -- and evaluated in one batched call (MAC.evaluate_objective_batch); the first maximum wins,
    like the strict ``>`` of the reference's loop.  The cumulative sums stay on the host: a parallel scan rounds
    differently from ``np.cumsum`` and could move a pick across an interval boundary."""
    if (value_fn is None and batch_value_fn is None) or max_iters == 1:
        return round_madow_base(w, k, seed)
    if batch_value_fn is not None:
        draws = np.stack([round_madow_base(w, k, seed) for _ in range(max_iters)])
        return draws[int(np.argmax(batch_value_fn(draws)))]
    best_x, best_val = None, -np.inf
    for _ in range(max_iters):
        x = round_madow_base(w, k, seed)
        val = value_fn(x)
        if val > best_val:
            best_val, best_x = val, x
    return best_x
