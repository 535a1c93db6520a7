"""Multi-GPU host logic: one process per GPU, candidates sharded in contiguous
ranges, one RCCL all-gather of the gradient per Frank-Wolfe iteration inside libmachip
(``machip_comm_init``).  The process group passed in is only used for the rendezvous (unique-id
broadcast) and for timing barriers -- any object with torch.distributed's ``broadcast_object_list`` /
``barrier`` works (gloo in bench.py and in the CPU tests)."""
from __future__ import annotations

from typing import Tuple


def shard_bounds(m: int, rank: int, nranks: int) -> Tuple[int, int, int]:
    """(lo, hi, shard) of rank's candidate range; identical to compute_gradient() in
    mac_amd/csrc/machip.hip: shard = ceil(m / R), ranges are [r*shard, (r+1)*shard) clipped to m,
    the gathered vector is padded to R*shard entries."""
    assert 0 <= rank < nranks
    shard = (m + nranks - 1) // nranks
    lo = min(m, shard * rank)
    hi = min(m, lo + shard)
    return lo, hi, shard


def exchange_unique_id(dist, rank: int, make_id) -> bytes:
    """Rank 0 creates the 128-byte ncclUniqueId (``make_id()``), everyone receives it."""
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = box[0]
    assert isinstance(uid, (bytes, bytearray)) and len(uid) == 128
    return bytes(uid)


def attach(problem, dist, rank: int, nranks: int):
    """Join ``problem`` (a mac_amd._lib.Problem) to the RCCL communicator of the job."""
    from mac_amd import _lib
    uid = exchange_unique_id(dist, rank, _lib.comm_unique_id)
    problem.comm_init(rank, nranks, uid)
    return shard_bounds(problem.m, rank, nranks)
