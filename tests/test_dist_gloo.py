"""World-size-2 CPU test (gloo) of the multi-GPU host logic: shard bounds / padding exactly as
libmachip computes them, the unique-id rendezvous, and that the sharded-gradient protocol
(each rank evaluates its candidate range, all-gather, everything else replicated) reproduces the
single-process Frank-Wolfe trajectory bit for bit.  The arithmetic here is the CPU oracle; on GPUs
the same protocol runs inside machip_fw_step with ncclAllGather."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from mac_amd.dist import shard_bounds


def test_shard_bounds_cover_and_pad():
    for m in [0, 1, 7, 64, 500534, 2001737]:
        for R in [1, 2, 3, 8]:
            got = []
            for r in range(R):
                lo, hi, shard = shard_bounds(m, r, R)
                assert 0 <= lo <= hi <= m and hi - lo <= shard
                got.extend(range(lo, hi)) if m < 100 else got.append((lo, hi))
                assert shard * R >= m and shard * R - m < R
            if m < 100:
                assert got == list(range(m))
            else:
                assert got[0][0] == 0 and got[-1][1] == m and all(a[1] == b[0] for a, b in zip(got, got[1:]))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle
    from mac_amd.dist import exchange_unique_id, shard_bounds as sb
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = exchange_unique_id(dist, rank, lambda: bytes(range(128)))
        g = load_golden("er300_solve")
        n, k = int(g["n"]), int(g["k"])
        mo = oracle.MacOracle(g["fi"], g["fj"], g["fw"], g["ci"], g["cj"], g["cw"], n)
        m = len(g["cw"])
        lo, hi, shard = sb(m, rank, world)
        x = g["x_init"].copy()
        fs = []
        for it in range(int(g["max_iters"])):
            f, v, _ = oracle.find_fiedler_pair(mo.laplacian(x))          # replicated eigen-solve
            mine = np.zeros(shard)
            mine[:hi - lo] = oracle.supergradient(v, mo.ci[lo:hi], mo.cj[lo:hi], mo.cw[lo:hi])
            parts = [torch.zeros(shard, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(mine))               # the one collective per iteration
            grad = torch.cat(parts).numpy()[:m]
            s = oracle.solve_subset_box_lp(grad, k)
            x = x + oracle.naive_stepsize(it) * (s - x)
            fs.append(float(f))
        dist.barrier()
        q.put((rank, uid == bytes(range(128)), fs, x))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_gradient_matches_single_process():
    import multiprocessing as mp      # plain spawn: torch is imported in the children only
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("er300_solve")
    res.sort(key=lambda t: t[0])
    for rank, uid_ok, fs, x in res:
        assert uid_ok
        assert np.allclose(fs, g["f_traj"], rtol=1e-9)
        assert np.allclose(x, g["unrounded"], atol=1e-12)
    assert np.array_equal(res[0][3], res[1][3])        # replicas stay bit-identical


def _fg_worker(rank, world, key, q):
    sys.path.insert(0, ROOT)
    from mac_amd.dist import FileGroup, exchange_unique_id
    g = FileGroup(rank, world, key=key, timeout=60)
    uid = exchange_unique_id(g, rank, lambda: bytes([7]) * 128)
    g.barrier()
    mx = g.max(10.0 + rank)
    got = g.all_gather_object(("r", rank))
    g.close()
    q.put((rank, uid == bytes([7]) * 128, mx, got))


def test_file_group_two_ranks():
    """The torch-free single-node process group bench.py uses under torch.distributed.run."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"test_{os.getpid()}"
    procs = [ctx.Process(target=_fg_worker, args=(r, 2, key, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok, mx, got in res:
        assert ok and mx == 11.0 and got == [("r", 0), ("r", 1)]
    assert not os.path.exists(os.path.join("/tmp", f"machip_rdzv_{key}"))


class _FakeProblem:
    """Stands in for mac_amd._lib.Problem on a machine without a GPU: records what attach_ipc / detach_ipc hand to the
    C ABI (machip_ipc_export -> machip_comm_init_ipc -> machip_comm_close_ipc)."""

    def __init__(self, rank, m):
        self.rank, self.m, self.calls = rank, m, []

    def ipc_export(self):
        self.calls.append("export")
        return bytes([self.rank]) * 16

    def comm_init_ipc(self, rank, nranks, blobs, timeout_s):
        self.calls.append(("init", rank, nranks, tuple(blobs), timeout_s))

    def comm_close_ipc(self):
        self.calls.append("close")


def _ipc_proto_worker(rank, world, key, q):
    sys.path.insert(0, ROOT)
    from mac_amd.dist import FileGroup, attach_ipc, detach_ipc
    g = FileGroup(rank, world, key=key)
    P = _FakeProblem(rank, 1001)
    lo, hi, shard = attach_ipc(P, g, rank, world, timeout_s=3.5)
    detach_ipc(P, g)
    g.close()
    q.put((rank, P.calls, (lo, hi, shard)))


def test_ipc_attach_protocol_three_ranks():
    """mac_amd.dist.attach_ipc / detach_ipc over the torch-free file group with three processes: every rank exports once,
    receives ALL blobs in rank order (its own included, at its own index), maps them with the limit it was given, gets the
    shard bounds libmachip computes, and says goodbye only after a barrier.  (The product side of the same calls --
    hipIpcOpenMemHandle, the device-ordered steps -- runs in tests/test_gpu_parity.py::test_ipc_* with real processes on a GPU.)"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    key = f"ipc_{os.getpid()}"
    world = 3
    procs = [ctx.Process(target=_ipc_proto_worker, args=(r, world, key, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    blobs = tuple(bytes([r]) * 16 for r in range(world))
    for rank, calls, bounds in res:
        assert calls == ["export", ("init", rank, world, blobs, 3.5), "close"]
        assert bounds == shard_bounds(1001, rank, world)
