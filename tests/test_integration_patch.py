"""INTEGRATION.md sections 2a / 2b are the patch a maintainer of the reference would apply at its two plugin points
(mac/utils/fiedler.py:38-42, the `method` string; mac/solvers/mac.py:58-65 + 104-128, the `problem` callable of
frank_wolfe).  These tests take the LITERAL '+' lines of those sections out of INTEGRATION.md, splice them into minimal stubs
of the two reference interfaces (skeletons written here from the interface description in SURVEY 8(b): signatures, the
RandomState(7) start block, the edge arrays -- no reference source), and call through them:
  * without a GPU the call must reach libmachip.so through ctypes and fail LOUDLY (NO_DEVICE) -- the patch text is
    syntactically valid, names every argument the binding takes, and there is no CPU fallback behind it;
  * on a GPU (`-m gpu`) `find_fiedler_pair(L, method='hip')` reproduces the reference's K5 known answer
    (tests/utils/test_fiedler.py:26-33) and a reference golden, and `MAC.problem` the golden (f, gradient) of the Petersen case.
"""
import os
import re
import types

import numpy as np
import pytest
import scipy.sparse as sps

from conftest import ROOT, load_golden
from mac_amd import _lib
from mac_amd.utils.graphs import Edge


def plus_lines(section):
    """'+' lines of the first ```python block under the heading that starts with `section` ('+' -> ' ': original indentation)."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    at = txt.index(section)
    block = re.search(r"```python\n(.*?)```", txt[at:], flags=re.S).group(1)
    return block.splitlines()


def stub_fiedler_module():
    added = [" " + l[1:] for l in plus_lines("### 2a.") if l.startswith("+")]
    assert any("method == 'hip'" in l for l in added) and any("fiedler_csr" in l for l in added)
    src = "\n".join([
        "import numpy as np",
        "import scipy as sp",
        "import scipy.sparse",
        "def find_fiedler_pair(L, X=None, method='tracemin_lu', tol=1e-8, seed=None):",
        "    q = min(4, L.shape[0] - 1)",
        "    if X is None:",
        "        X = np.random.RandomState(7).normal(size=(q, L.shape[0])).T",
        "    assert X.shape == (L.shape[0], q)",
        "    if method == 'tracemin_cholesky':",
        "        raise NotImplementedError('stub: needs sksparse')",
        *added,
        "    else:",
        "        raise NotImplementedError('stub: networkx path')",
        "    return (sigma[0], X[:, 0], X)",
    ])
    mod = types.ModuleType("stub_fiedler")
    exec(compile(src, "INTEGRATION.md#2a", "exec"), mod.__dict__)
    return mod


def stub_mac_class():
    lines = plus_lines("### 2b.")
    cut = next(i for i, l in enumerate(lines) if "def problem(self, x, cache=None):" in l)
    init_add = [" " + l[1:] for l in lines[:cut] if l.startswith("+")]
    prob_add = [" " + l[1:] for l in lines[cut:] if l.startswith("+")]
    assert any("_lib.Problem(" in l for l in init_add) and any("self._dev.gradient()" in l for l in prob_add)
    src = "\n".join([
        "import numpy as np",
        "class Cache:",
        "    Q = None",
        "class MAC:",
        "    def __init__(self, fixed_edges, candidate_edges, num_nodes, fiedler_method='tracemin_lu', fiedler_tol=1e-8,",
        "                 min_selection_weight_tol=1e-10):",
        "        self.num_nodes = num_nodes",
        "        self.weights = np.array([e.weight for e in candidate_edges], dtype=float)",
        "        self.edge_list = np.array([(e.i, e.j) for e in candidate_edges], dtype=np.int64)",
        *init_add,
        "    def problem(self, x, cache=None):",
        *prob_add,
    ])
    mod = types.ModuleType("stub_mac")
    exec(compile(src, "INTEGRATION.md#2b", "exec"), mod.__dict__)
    return mod


def k5_laplacian():
    A = np.ones((5, 5)) - np.eye(5)
    return sps.csr_matrix(np.diag(A.sum(axis=1)) - A)


def petersen_edges():
    g = load_golden("petersen_solve_k3")
    fixed = [Edge(int(a), int(b), float(w)) for a, b, w in zip(g["fi"], g["fj"], g["fw"])]
    cand = [Edge(int(a), int(b), float(w)) for a, b, w in zip(g["ci"], g["cj"], g["cw"])]
    return g, fixed, cand


def test_patch_text_compiles_and_reaches_the_library_without_fallback():
    if _lib.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu-marked test")
    mod = stub_fiedler_module()
    with pytest.raises(_lib.MachipError) as e:
        mod.find_fiedler_pair(k5_laplacian(), method="hip")
    assert e.value.status == _lib.NO_DEVICE
    g, fixed, cand = petersen_edges()
    with pytest.raises(_lib.MachipError) as e:
        stub_mac_class().MAC(fixed, cand, 10)
    assert e.value.status == _lib.NO_DEVICE


@pytest.mark.gpu
def test_patched_reference_plugin_points_give_the_reference_answers():
    mod = stub_fiedler_module()
    lam, v, X = mod.find_fiedler_pair(k5_laplacian(), method="hip")
    assert np.isclose(lam, 5.0) and X.shape == (5, 4) and abs(v.sum()) < 1e-9      # tests/utils/test_fiedler.py:26-33
    gx = load_golden("er300_x0")
    L = sps.csr_matrix((gx["L_data"], gx["L_indices"], gx["L_indptr"]), shape=(int(gx["n"]),) * 2)
    lam, v, X = mod.find_fiedler_pair(L, method="hip")
    assert abs(lam - float(gx["lam"])) <= 1e-8 * float(gx["lam"]) and X.shape == (int(gx["n"]), 4)
    assert np.abs(L @ v - lam * v).sum() / abs(L).sum(axis=1).max() < 1e-8          # the reference's stop rule (nx:246)
    g, fixed, cand = petersen_edges()
    mac = stub_mac_class().MAC(fixed, cand, 10)
    f, grad = mac.problem(g["x_init"])
    assert abs(f - g["f_traj"][0]) <= 1e-8 * g["f_traj"][0]
    assert np.allclose(grad, g["g_traj"][0], rtol=1e-5, atol=1e-7)
    cache = types.SimpleNamespace(Q=None)
    f2, _ = mac.problem(g["x_init"], cache=cache)                                    # the cache slot is written like mac.py:126-127
    assert cache.Q is not None and abs(f2 - f) <= 1e-10 * f
