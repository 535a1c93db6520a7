from .mac import MAC  # noqa: F401
from .baseline import NaiveGreedy  # noqa: F401
