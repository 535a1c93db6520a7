"""NaiveGreedy (mac/solvers/baseline.py:3-14): top-k candidates by weight; the usual
x_init of MAC.solve (examples/g2o_experiment.py:312-315)."""
import numpy as np


class NaiveGreedy:
    def __init__(self, edges):
        self.weights = np.array([e.weight for e in edges])

    def subset(self, k):
        solution = np.zeros(len(self.weights))
        if k > 0:
            solution[np.argpartition(self.weights, -k)[-k:]] = 1.0
        return solution
