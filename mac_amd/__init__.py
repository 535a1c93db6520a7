"""mac_amd -- MI355X (gfx950) implementation of the Frank-Wolfe / Fiedler hot path of
MarineRoboticsGroup/mac, behind the reference's own API.

    from mac_amd.solvers import MAC, NaiveGreedy          # mac/solvers/__init__.py:1-2
    from mac_amd.utils.fiedler import find_fiedler_pair    # mac/utils/fiedler.py:9
    from mac_amd.utils.graphs import Edge                  # mac/utils/graphs.py:11

Host code is plain Python + NumPy; all arithmetic of the path runs in hand-written HIP
kernels reached through the C ABI of ``libmachip.so`` (include/machip.h) via ctypes.
There is no CPU fallback: without the library or without a GPU the product raises.
"""
__version__ = "0.1.0"
