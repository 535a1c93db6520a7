"""Import-compatibility package: put ``<repo>/compat`` on ``PYTHONPATH`` (next to the repo root) and code written against
the reference keeps its import lines --

    from mac.solvers import MAC, NaiveGreedy            # mac/solvers/__init__.py:1-2 of the reference
    from mac.utils.graphs import Edge
    from mac.utils.fiedler import find_fiedler_pair
    from mac.utils.rounding import round_madow, round_nearest
    from mac.optimization.frankwolfe import frank_wolfe

-- and runs the MI355X implementation (``mac_amd``): this package holds no code of its own, it registers the ``mac_amd``
modules under the reference's module names.  The reference's baselines outside the hot path (``mac.solvers.greedy_esp``,
``greedy_eig``, ``mac.utils.cholesky``: SURVEY section 8, out of scope) are not provided: importing them raises ImportError, as
it does in the reference without its optional SuiteSparse dependency.
"""
import importlib
import sys

_ALIASES = {
    "mac.solvers": "mac_amd.solvers",
    "mac.solvers.mac": "mac_amd.solvers.mac",
    "mac.solvers.baseline": "mac_amd.solvers.baseline",
    "mac.utils": "mac_amd.utils",
    "mac.utils.graphs": "mac_amd.utils.graphs",
    "mac.utils.fiedler": "mac_amd.utils.fiedler",
    "mac.utils.rounding": "mac_amd.utils.rounding",
    "mac.utils.conversions": "mac_amd.utils.graphs",       # nx_to_mac lives with the graph helpers here
    "mac.optimization": "mac_amd.optimization",
    "mac.optimization.frankwolfe": "mac_amd.optimization.frankwolfe",
    "mac.optimization.constraints": "mac_amd.optimization.constraints",
}
for _alias, _real in _ALIASES.items():
    sys.modules[_alias] = importlib.import_module(_real)
solvers = sys.modules["mac.solvers"]
utils = sys.modules["mac.utils"]
optimization = sys.modules["mac.optimization"]
