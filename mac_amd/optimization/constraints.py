"""Direction-finding LPs of the Frank-Wolfe driver (mac/optimization/constraints.py).  Inside
``MAC.solve`` the subset-box LP runs on the GPU (radix select, k_sel_pass); these host
versions serve generic callers of ``frank_wolfe``."""
import numpy as np

from mac_amd.utils.rounding import round_nearest


def solve_subset_box_lp(g, k):
    """argmax g.x s.t. 0<=x<=1, |x|_0<=k: indicator of the k largest (constraints.py:12-22)."""
    return round_nearest(g, k)


def solve_box_lp(g):
    """argmax g.x s.t. 0<=x<=1: indicator of the positive entries (constraints.py:24-37)."""
    s = np.zeros_like(g)
    s[g > 0.0] = 1.0
    return s
