"""Multi-GPU host logic: one process per GPU, candidates sharded in contiguous
ranges, one RCCL all-gather of the gradient per Frank-Wolfe iteration inside libmachip
(``machip_comm_init``).  The process group passed in is only used for the rendezvous (unique-id
broadcast) and for timing barriers -- any object with torch.distributed's ``broadcast_object_list`` /
``barrier`` works (gloo in bench.py and in the CPU tests)."""
from __future__ import annotations

import os
import pickle
import time
from typing import Tuple


def shard_bounds(m: int, rank: int, nranks: int) -> Tuple[int, int, int]:
    """(lo, hi, shard) of rank's candidate range, computed BY libmachip (machip_shard_plan: the very function
    compute_gradient() and machip_comm_init use): shard = ceil(m / R), ranges are [r*shard, (r+1)*shard) clipped
    to m, the gathered vector is padded to R*shard entries."""
    assert 0 <= rank < nranks
    from mac_amd import _lib
    return _lib.shard_plan(m, nranks, rank)


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


def _all_gather(dist, obj, world: int):
    """all_gather_object under either signature: FileGroup's ``(obj) -> list`` or torch.distributed's
    ``(object_list, obj)``."""
    import inspect
    try:
        npar = len([p for p in inspect.signature(dist.all_gather_object).parameters.values()
                    if p.default is inspect.Parameter.empty and p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)])
    except (TypeError, ValueError):
        npar = 1
    if npar >= 2:
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    return dist.all_gather_object(obj)


def attach_ipc(problem, dist, rank: int, nranks: int, timeout_s: float = 10.0):
    """Row-partition the eigen-solve of ``problem`` between the processes of the job (machip_comm_init_ipc): every rank
    exports IPC handles of its exchange buffers, the group carries the blobs, every rank maps its peers' buffers.  Call
    after ``attach`` when an RCCL communicator carries the gradient, or alone (ranks sharing one GPU: the gradient shards
    then travel through the mapped buffers too).  ``detach_ipc`` before closing the problem.

    Exception-safe and collective: every rank runs the same two exchanges whatever happens locally (an export or a mapping
    that fails is carried as a flag), so no rank is left one collective behind; when ANY rank failed, every rank drops the
    communicator again (the handle is a single-rank handle as before) and raises -- the caller falls back to a replicated
    solve on all ranks alike."""
    err = None
    blob = None
    try:
        blob = problem.ipc_export()
    except Exception as e:          # noqa: BLE001
        err = e
    blobs = _all_gather(dist, blob, nranks)
    if err is None:
        if any(b is None for b in blobs):
            err = RuntimeError(f"IPC export failed on rank(s) {[r for r, b in enumerate(blobs) if b is None]}")
        else:
            try:
                problem.comm_init_ipc(rank, nranks, blobs, timeout_s)
            except Exception as e:  # noqa: BLE001
                err = e
    oks = _all_gather(dist, err is None, nranks)      # agreement; doubles as the barrier "everybody has mapped everybody"
    if not all(oks):
        if err is None:
            err = RuntimeError(f"IPC attach failed on rank(s) {[r for r, o in enumerate(oks) if not o]}")
        try:
            problem.comm_drop_ipc()
        except Exception:           # noqa: BLE001  (nothing was attached on this rank)
            pass
        raise err
    return shard_bounds(problem.m, rank, nranks)


def detach_ipc(problem, dist):
    """Orderly exit: nobody unmaps / frees while a peer may still write."""
    dist.barrier()
    problem.comm_close_ipc()


def default_key() -> str:
    ppid = os.getppid()
    start = "0"
    try:
        with open(f"/proc/{ppid}/stat") as fh:      # field 22 = start time of the launcher (clock ticks since boot)
            start = fh.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        pass
    return f"{os.environ.get('MASTER_PORT', '0')}_{ppid}_{start}"


class FileGroup:
    """Single-node process group over a shared temp directory: rendezvous / barrier / max only.

    bench.py uses it instead of torch.distributed so that no second HIP runtime (the one bundled
    with the PyTorch wheel) is ever loaded next to libmachip's -- the data path is RCCL inside
    libmachip either way.  The directory key is ``MACHIP_RDZV_KEY`` when the launcher sets one (bench.py's own
    launcher passes a fresh uuid), else MASTER_PORT + the launcher's PID + the launcher's start time from
    /proc (all ranks of one ``torch.distributed.run`` share their parent; a crashed earlier job that happened to
    share port and launcher PID has a different start time, so its leftovers are never read).
    Offers the subset of torch.distributed's interface that mac_amd.dist needs."""

    def __init__(self, rank: int, world: int, key: str = None, root: str = "/tmp", timeout: float = 600.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            key = os.environ.get("MACHIP_RDZV_KEY") or default_key()
        self.dir = os.path.join(root, f"machip_rdzv_{key}")
        os.makedirs(self.dir, exist_ok=True)
        self._seq = 0

    def _path(self, tag, r):
        return os.path.join(self.dir, f"{tag}.{r}")

    def _put(self, tag, obj):
        tmp = self._path(tag, self.rank) + ".tmp"
        with open(tmp, "wb") as fh:
            pickle.dump(obj, fh)
        os.replace(tmp, self._path(tag, self.rank))      # atomic publish

    def _get(self, tag, r):
        p = self._path(tag, r)
        t0 = time.monotonic()
        while not os.path.exists(p):
            if time.monotonic() - t0 > self.timeout:
                raise TimeoutError(f"rank {self.rank}: waited {self.timeout}s for rank {r} at '{tag}'")
            time.sleep(0.0005)
        with open(p, "rb") as fh:
            return pickle.load(fh)

    def all_gather_object(self, obj):
        self._seq += 1
        tag = f"s{self._seq}"
        self._put(tag, obj)
        return [self._get(tag, r) for r in range(self.world)]

    def barrier(self):
        self.all_gather_object(None)

    def broadcast_object_list(self, box, src=0):
        vals = self.all_gather_object(box[0] if self.rank == src else None)
        box[0] = vals[src]

    def max(self, value: float) -> float:
        return max(self.all_gather_object(float(value)))

    def close(self):
        self.barrier()
        self._put("bye", None)          # nobody reads anything of mine after this
        if self.rank == 0:
            for r in range(self.world):  # only now is it safe to remove the directory
                self._get("bye", r)
            for f in os.listdir(self.dir):
                try:
                    os.remove(os.path.join(self.dir, f))
                except OSError:
                    pass
            try:
                os.rmdir(self.dir)
            except OSError:
                pass
