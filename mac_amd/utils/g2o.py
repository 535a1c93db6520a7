"""g2o pose-graph ingestion for the MAC hot path.

Restates the part of examples/pose_graph_utils.py the solver needs -- read_g2o_file
(228-351), split_edges (18-45), rpm_to_mac (381-396) -- without its plotting / evo / SE-Sync
dependencies: each EDGE line becomes one graph edge weighted by the rotation concentration
kappa (EDGE_SE2: kappa = I33; EDGE_SE3:QUAT: kappa = 3 / (2 tr(inv(I[3:6,3:6])))).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from mac_amd.utils.graphs import Edge


def read_g2o_edges(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray, int]:
    """(i int64[E], j int64[E], kappa float64[E], num_poses) in file order."""
    I, J, K = [], [], []
    se3_rows = []          # (edge position, 6 rotational-information entries)
    with open(path, "r") as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "EDGE_SE2":
                I.append(int(float(tok[1]))); J.append(int(float(tok[2]))); K.append(float(tok[11]))
            elif tok[0] == "EDGE_SE3:QUAT":
                I.append(int(float(tok[1]))); J.append(int(float(tok[2]))); K.append(0.0)
                u = [float(t) for t in tok[10:31]]      # upper triangle of the 6x6 information
                # rows: 0:(0..5) 6:(1..5) 11:(2..5) 15:(3..5) 18:(4..5) 20:(5)
                se3_rows.append((len(K) - 1, (u[15], u[16], u[17], u[18], u[19], u[20])))
    K = np.asarray(K, dtype=np.float64)
    if se3_rows:
        pos = np.fromiter((p for p, _ in se3_rows), dtype=np.int64, count=len(se3_rows))
        a = np.asarray([r for _, r in se3_rows], dtype=np.float64)
        M = np.empty((len(a), 3, 3))
        M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = a[:, 0], a[:, 1], a[:, 2]
        M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = a[:, 1], a[:, 3], a[:, 4]
        M[:, 2, 0], M[:, 2, 1], M[:, 2, 2] = a[:, 2], a[:, 4], a[:, 5]
        tr = np.trace(np.linalg.inv(M), axis1=1, axis2=2)
        K[pos] = 3.0 / (2.0 * tr)
    I = np.asarray(I, dtype=np.int64)
    J = np.asarray(J, dtype=np.int64)
    n = int(max(I.max(), J.max())) + 1 if len(I) else 0     # zero-based ids (pose_graph_utils.py:348)
    return I, J, K, n


def split_chain(i, j) -> np.ndarray:
    """True for 'fixed' odometry edges (|i - j| <= 1), False for loop closures
    (pose_graph_utils.py:18-45)."""
    return np.abs(np.asarray(j) - np.asarray(i)) <= 1


def read_g2o_file(path: str) -> Tuple[List[Edge], int]:
    """Edge list weighted by kappa + number of poses (read_g2o_file + rpm_to_mac)."""
    i, j, k, n = read_g2o_edges(path)
    return [Edge(int(a), int(b), float(c)) for a, b, c in zip(i, j, k)], n


def split_edges(edges: List[Edge]) -> Tuple[List[Edge], List[Edge]]:
    """(chain edges, loop-closure edges) with the reference's rule (pose_graph_utils.py:18-45)."""
    chain = [e for e in edges if abs(e.j - e.i) <= 1]
    loops = [e for e in edges if abs(e.j - e.i) > 1]
    return chain, loops
