"""Public solver surface of the drop-in (the names mac/solvers/__init__.py exports)."""
from mac_amd.solvers.baseline import NaiveGreedy  # noqa: F401
from mac_amd.solvers.mac import MAC  # noqa: F401

__all__ = ["MAC", "NaiveGreedy"]
