"""Weight-greedy baseline with the reference's class name and method (mac/solvers/baseline.py:3-14):
keep the k heaviest candidate edges.  It doubles as the usual sparse initial point of ``MAC.solve``
(examples/g2o_experiment.py:312-315)."""
import numpy as np

from mac_amd.utils.rounding import round_nearest


class NaiveGreedy:
    def __init__(self, edges):
        self.weights = np.fromiter((e[2] for e in edges), dtype=np.float64, count=len(edges))

    def subset(self, k):
        """0/1 vector with ones on the k largest weights (ties resolved as numpy.argpartition does)."""
        return round_nearest(self.weights, k)
