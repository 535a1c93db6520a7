"""Graph types and Laplacian builders with the reference's names and semantics
(mac/utils/graphs.py).  Edge lists are converted once to flat SoA arrays (int32 ids,
float64 weights): that is the layout the HIP kernels consume."""
from __future__ import annotations

from collections import namedtuple
from typing import List, Sequence, Tuple

import numpy as np
from scipy.sparse import coo_matrix, csr_matrix

# mac/utils/graphs.py:11
Edge = namedtuple("Edge", ["i", "j", "weight"])


def edges_to_arrays(edges: Sequence[Edge]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """List[Edge] -> (i int32[m], j int32[m], w float64[m])."""
    m = len(edges)
    ei = np.fromiter((e[0] for e in edges), dtype=np.int64, count=m)
    ej = np.fromiter((e[1] for e in edges), dtype=np.int64, count=m)
    ew = np.fromiter((e[2] for e in edges), dtype=np.float64, count=m)
    return ei.astype(np.int32), ej.astype(np.int32), ew


def _lap_from_arrays(ei, ej, ew, num_nodes) -> csr_matrix:
    ei = np.asarray(ei, dtype=np.int64).ravel()
    ej = np.asarray(ej, dtype=np.int64).ravel()
    ew = np.asarray(ew, dtype=np.float64).ravel()
    rows = np.concatenate([ei, ej, ei, ej])
    cols = np.concatenate([ei, ej, ej, ei])
    data = np.concatenate([ew, ew, -ew, -ew])
    return csr_matrix(coo_matrix((data, (rows, cols)), shape=[num_nodes, num_nodes]))


def weight_graph_lap_from_edge_list(edges: List[Edge], num_nodes: int) -> csr_matrix:
    """Weighted graph Laplacian from a list of Edge (mac/utils/graphs.py:13-48).
    One-off host-side setup (MAC.__init__ / tests); the per-iteration assembly of L(x) is
    the HIP kernel pair k_asm_count / k_asm_fill."""
    ei, ej, ew = edges_to_arrays(edges)
    return _lap_from_arrays(ei, ej, ew, num_nodes)


def weight_reduced_graph_lap_from_edge_list(edges: List[Edge], num_nodes: int) -> csr_matrix:
    """mac/utils/graphs.py:51-55."""
    return weight_graph_lap_from_edge_list(edges, num_nodes)[1:, 1:]


def weight_graph_lap_from_edges(edges, weights, num_nodes: int) -> csr_matrix:
    """Laplacian from an [s,2] index array and s weights (mac/utils/graphs.py:58-98)."""
    edges = np.asarray(edges).reshape(-1, 2)
    assert len(edges) == len(weights)
    return _lap_from_arrays(edges[:, 0], edges[:, 1], weights, num_nodes)


def select_edges(edges, w):
    """Edges whose selection weight is exactly 1 (mac/utils/graphs.py:101-111)."""
    assert len(edges) == len(w), f"Selection mask length {len(w)} does not match number of edges {len(edges)}"
    return [e for e, wi in zip(edges, w) if wi == 1.0]


def nx_to_mac(G) -> List[Edge]:
    """networkx graph -> Edge list with i < j (mac/utils/conversions.py:9-31)."""
    out = []
    for (a, b, d) in G.edges(data=True):
        w = d.get("weight", 1.0)
        out.append(Edge(a, b, w) if a < b else Edge(b, a, w))
    return out
