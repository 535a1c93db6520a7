"""ctypes binding of libmachip.so (C ABI: include/machip.h).  No torch, no fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MACHIP_LIB") or os.path.join(_HERE, "libmachip.so")   # MACHIP_LIB: developer override (sanitizer builds)

OK, NOT_CONVERGED, DISCONNECTED, BAD_ARG, HIP_ERROR, RCCL_ERROR, NO_DEVICE = range(7)
STATUS_NAMES = ["OK", "NOT_CONVERGED", "DISCONNECTED", "BAD_ARG", "HIP_ERROR", "RCCL_ERROR", "NO_DEVICE"]


class MachipError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"libmachip: {STATUS_NAMES[status] if 0 <= status < 7 else status}: {msg}")
        self.status = status


class NotConverged(MachipError):
    pass


class Disconnected(MachipError):
    """lambda_2 ~ 0 (the reference fails with SuperLU 'Factor is exactly singular')."""


class SolveStats(C.Structure):
    _fields_ = [("lanczos_steps", C.c_int64), ("spmv_total", C.c_int64), ("vec_passes", C.c_int64),
                ("restarts", C.c_int64), ("nnz", C.c_int64), ("support", C.c_int64),
                ("residual", C.c_double), ("lnorm", C.c_double), ("gpu_ms", C.c_double),
                ("step_ms", C.c_double), ("steps_timed", C.c_int64), ("steps_lowp", C.c_int64), ("drift", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_lib = None

# name -> (restype, argtypes); every symbol declared in include/machip.h
SIGNATURES = {
    "machip_version": (C.c_int, []),
    "machip_sizeof_stats": (C.c_int, []),
    "machip_device_count": (C.c_int, []),
    "machip_last_error": (C.c_char_p, []),
    "machip_create": (C.c_int, [C.c_int, C.c_int64, C.c_int64, _i32p, _i32p, _f64p, C.c_int64, _i32p, _i32p,
                                _f64p, C.c_double, C.POINTER(C.c_void_p)]),
    "machip_destroy": (None, [C.c_void_p]),
    "machip_set_x": (C.c_int, [C.c_void_p, _f64p]),
    "machip_get_x": (C.c_int, [C.c_void_p, _f64p]),
    "machip_assemble": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "machip_get_laplacian": (C.c_int, [C.c_void_p, _i32p, _i32p, _f64p]),
    "machip_fiedler": (C.c_int, [C.c_void_p, C.c_double, C.c_int, _f64p, C.c_int, _f64p, _f64p, _f64p, C.c_int,
                                 C.POINTER(SolveStats)]),
    "machip_set_start": (C.c_int, [C.c_void_p, _f64p]),
    "machip_gradient": (C.c_int, [C.c_void_p, _f64p]),
    "machip_lp_topk": (C.c_int, [C.c_void_p, C.c_int64, _f64p]),
    "machip_fw_step": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_int, C.c_int, _f64p, _f64p,
                                 _f64p, C.POINTER(SolveStats)]),
    "machip_fw_commit": (C.c_int, [C.c_void_p]),
    "machip_fw_run": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _f64p, _f64p,
                                _f64p, _f64p, C.POINTER(SolveStats), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "machip_round_nearest": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, _f64p]),
    "machip_fiedler_csr": (C.c_int, [C.c_int, C.c_int64, _i32p, _i32p, _f64p, C.c_double, C.c_int, _f64p, _f64p,
                                     _f64p, _f64p, C.c_int, C.POINTER(SolveStats)]),
    "machip_spmv": (C.c_int, [C.c_void_p, _f64p, _f64p, C.c_int]),
    "machip_landscape": (C.c_int, [C.c_void_p, C.c_int, _f64p]),
    "machip_profile_spmv": (C.c_int, [C.c_void_p, C.c_int, _f64p, _f64p]),
    "machip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "machip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "machip_comm_init_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "machip_selftest_watchdog": (C.c_int, [C.c_int, C.c_int]),
    "machip_peer_access": (C.c_int, [C.c_int, C.c_int]),
    "machip_ipc_blob_bytes": (C.c_int, []),
    "machip_ipc_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "machip_comm_init_ipc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double]),
    "machip_comm_close_ipc": (C.c_int, [C.c_void_p]),
    "machip_comm_mode": (C.c_int, [C.c_void_p]),
    "machip_shard_plan": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "machip_eval_batch": (C.c_int, [C.c_void_p, C.c_int, _f64p, C.c_double, C.c_int, _f64p, C.POINTER(C.c_int)]),
    "machip_fw_sweep": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), _f64p, C.c_int, C.c_double, C.c_double, C.c_double,
                                  C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "machip_set_solver": (C.c_int, [C.c_void_p, C.c_int]),
    "machip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "machip_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "machip_option_name": (C.c_char_p, [C.c_int]),
    "machip_comm_drop_ipc": (C.c_int, [C.c_void_p]),
    "machip_comm_drop": (C.c_int, [C.c_void_p]),
    "machip_comm_timing": (C.c_int, [C.c_void_p, _f64p, _f64p]),
    "machip_solve_mode": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "machip_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "machip_synchronize": (C.c_int, [C.c_void_p]),
    "machip_membench": (C.c_int, [C.c_int, C.c_int64, C.c_int, _f64p, _f64p]),
    "machip_host_tridiag_smallest": (C.c_int, [_f64p, _f64p, C.c_int, _f64p, _f64p]),
    "machip_host_follow_records": (C.c_int, [_f64p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), C.c_int,
                                             C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "machip_release_cache": (None, []),
    "machip_panel_plan": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int)]),
}


def load():
    """Load libmachip.so (built by __graft_entry__.build() / mac_amd/csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                          "(hipcc --offload-arch=gfx950); mac_amd has no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    import atexit
    atexit.register(lib.machip_release_cache)     # the handle machip_fiedler_csr keeps between calls
    return lib


OPTION_AUTO = -(1 << 63)      # MACHIP_OPTION_AUTO: "the measured default"
CREATION_OPTIONS = frozenset({"asm_g", "asm_maxgrid", "vbudget_mb", "vcap"})      # read by machip_create only (options.h, last group)


def option_names():
    """Names of the option table (mac_amd/csrc/options.h)."""
    lib, out, i = load(), [], 0
    while True:
        nm = lib.machip_option_name(i)
        if nm is None:
            return out
        out.append(nm.decode())
        i += 1


def set_default_option(name, value=None):
    """Process default of an option: what handles created AFTERWARDS start from (value None = automatic).  The options a
    handle consumes at creation ("asm_g", "vbudget_mb", "vcap", "lane_queues" ...) can only be set this way."""
    check(load().machip_set_option(None, name.encode(), OPTION_AUTO if value is None else int(value)))


class default_options:
    """Context manager: process defaults for the handles created inside the block, restored afterwards (tests)."""

    def __init__(self, **opts):
        self.opts = opts

    def __enter__(self):
        lib = load()
        self.old = {}
        for k, v in self.opts.items():
            cur = C.c_int64(0)
            check(lib.machip_get_option(None, k.encode(), C.byref(cur)))
            self.old[k] = cur.value
            set_default_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            check(load().machip_set_option(None, k.encode(), v))
        return False


def membench(nbytes=1 << 30, reps=10, device=0):
    """(read GB/s, STREAM-triad GB/s) measured on the device (machip_membench)."""
    r, t = C.c_double(0.0), C.c_double(0.0)
    check(load().machip_membench(int(device), int(nbytes), int(reps), C.byref(r), C.byref(t)))
    return r.value, t.value


def device_count():
    return int(load().machip_device_count())


def require_device():
    n = device_count()
    if n <= 0:
        raise MachipError(NO_DEVICE, "no MI355X / HIP device visible; mac_amd has no CPU fallback")
    return n


def last_error():
    return (load().machip_last_error() or b"").decode("utf-8", "replace")


def check(status, allow=()):
    if status == OK or status in allow:
        return status
    msg = last_error()
    if status == NOT_CONVERGED:
        raise NotConverged(status, msg)
    if status == DISCONNECTED:
        raise Disconnected(status, msg)
    if status == BAD_ARG:
        raise AssertionError(f"libmachip: BAD_ARG: {msg}")     # the reference asserts (mac.py:47,52,183)
    raise MachipError(status, msg)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def p_f64(a):
    return None if a is None else a.ctypes.data_as(_f64p)


def p_i32(a):
    return None if a is None else a.ctypes.data_as(_i32p)


class Problem:
    """Owns one ``machip_problem`` handle (one MAC instance resident on one GPU)."""

    def __init__(self, n, fi, fj, fw, ci, cj, cw, min_selection_weight_tol=1e-10, device=0):
        lib = load()
        require_device()
        self.n = int(n)
        fi, fj, fw = i32(fi), i32(fj), f64(fw)
        ci, cj, cw = i32(ci), i32(cj), f64(cw)
        self.m = int(len(cw))
        h = C.c_void_p()
        check(lib.machip_create(int(device), self.n, len(fw), p_i32(fi), p_i32(fj), p_f64(fw), self.m,
                                p_i32(ci), p_i32(cj), p_f64(cw), float(min_selection_weight_tol), C.byref(h)))
        self._h = h
        self._lib = lib
        self.stats = SolveStats()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.machip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- x ----
    def set_x(self, x):
        x = f64(x)
        assert x.shape == (self.m,)
        check(self._lib.machip_set_x(self._h, p_f64(x)))

    def get_x(self):
        x = np.empty(self.m)
        check(self._lib.machip_get_x(self._h, p_f64(x)))
        return x

    # ---- Laplacian ----
    def assemble(self):
        nnz = C.c_int64()
        check(self._lib.machip_assemble(self._h, C.byref(nnz)))
        return int(nnz.value)

    def laplacian_csr(self):
        nnz = self.assemble()
        indptr = np.empty(self.n + 1, dtype=np.int32)
        indices = np.empty(nnz, dtype=np.int32)
        data = np.empty(nnz)
        check(self._lib.machip_get_laplacian(self._h, p_i32(indptr), p_i32(indices), p_f64(data)))
        return indptr, indices, data

    def spmv(self, v, variant=0):
        v = f64(v)
        y = np.empty(self.n)
        check(self._lib.machip_spmv(self._h, p_f64(v), p_f64(y), int(variant)))
        return y

    def landscape(self, sweeps=3):
        """u after `sweeps` Jacobi sweeps on L(x) u = 1 from u = 1/diag (machip_landscape): what a cold Lanczos start is
        weighted by (options start_land / start_pow)."""
        u = np.empty(self.n)
        check(self._lib.machip_landscape(self._h, int(sweeps), p_f64(u)))
        return u

    # ---- eigen-solve / gradient / LP ----
    def fiedler(self, tol=1e-8, max_steps=0, x0=None, warm_start=False, want_vec=True, q=0):
        lam = C.c_double()
        v = np.empty(self.n) if want_vec else None
        X = np.empty((q, self.n)) if q else None      # row q = column q of the n x q block
        x0a = f64(x0) if x0 is not None else None
        st = self._lib.machip_fiedler(self._h, float(tol), int(max_steps), p_f64(x0a), int(bool(warm_start)),
                                      C.byref(lam), p_f64(v), p_f64(X), int(q), C.byref(self.stats))
        check(st)
        return lam.value, v, (X.T if X is not None else None)

    def set_start(self, x0):
        x0 = f64(x0)
        assert x0.shape == (self.n,)
        check(self._lib.machip_set_start(self._h, p_f64(x0)))

    def gradient(self, want=True):
        g = np.empty(self.m) if want else None
        check(self._lib.machip_gradient(self._h, p_f64(g)))
        return g

    def lp_topk(self, k, want=True):
        s = np.empty(self.m) if want else None
        check(self._lib.machip_lp_topk(self._h, int(k), p_f64(s)))
        return s

    def fw_step(self, k, it, tol=1e-8, max_steps=0, warm_start=False):
        f, d, gn = C.c_double(), C.c_double(), C.c_double()
        check(self._lib.machip_fw_step(self._h, int(k), int(it), float(tol), int(max_steps),
                                       int(bool(warm_start)), C.byref(f), C.byref(d), C.byref(gn),
                                       C.byref(self.stats)))
        return f.value, d.value, gn.value

    def fw_commit(self):
        check(self._lib.machip_fw_commit(self._h))

    def fw_run(self, k, max_iters, first_iter=0, gap_tol=0.0, grad_tol=0.0, tol=1e-8, max_steps=0, warm_start=False, upper=float("inf")):
        """The Frank-Wolfe loop on the C side (machip_fw_run): up to max_iters iterations from the resident x with the
        reference's stop tests (0.0 disables them), no return to Python in between.  Returns a dict: f, dual, gnorm (arrays of
        the iterations done), upper, iters, stats (list of SolveStats), modes (list of (mode, closures))."""
        n = max(1, int(max_iters))
        f = np.zeros(n); d = np.zeros(n); gn = np.zeros(n)
        st = (SolveStats * n)()
        modes = (C.c_int * (2 * n))()
        up = C.c_double(float(upper)); done = C.c_int(0)
        status = self._lib.machip_fw_run(self._h, int(k), int(first_iter), int(max_iters), float(gap_tol), float(grad_tol), float(tol),
                                         int(max_steps), int(bool(warm_start)), C.byref(up), p_f64(f), p_f64(d), p_f64(gn), st, modes, C.byref(done))
        it = int(done.value)
        if it:
            C.memmove(C.byref(self.stats), C.byref(st[it - 1]), C.sizeof(SolveStats))
        check(status)
        return dict(f=f[:it], dual=d[:it], gnorm=gn[:it], upper=up.value, iters=it, stats=[st[i] for i in range(it)],
                    modes=[(int(modes[2 * i]), int(modes[2 * i + 1])) for i in range(it)])

    def round_nearest(self, k, decimals=10):
        """Device round_nearest of the resident x; decimals=None -> plain top-k."""
        out = np.empty(self.m)
        check(self._lib.machip_round_nearest(self._h, int(k), -1 if decimals is None else int(decimals), p_f64(out)))
        return out

    def profile_spmv(self, reps=200):
        us, by = C.c_double(), C.c_double()
        check(self._lib.machip_profile_spmv(self._h, int(reps), C.byref(us), C.byref(by)))
        return us.value, by.value

    def comm_init(self, rank, nranks, unique_id: bytes):
        _single_node_bootstrap()
        buf = C.create_string_buffer(unique_id, 128)
        with _stdout_to_stderr():
            st = self._lib.machip_comm_init(self._h, int(rank), int(nranks), buf)
        check(st)

    def ipc_export(self) -> bytes:
        """IPC handles of this handle's exchange buffers (machip_ipc_export): hand the blob to every peer process."""
        nb = self._lib.machip_ipc_blob_bytes()
        buf = C.create_string_buffer(nb)
        check(self._lib.machip_ipc_export(self._h, buf, nb))
        return buf.raw

    def comm_init_ipc(self, rank, nranks, blobs, timeout_s=10.0):
        """Map the peers' buffers (blobs: one per rank, in rank order) and row-partition the eigen-solve between the processes."""
        nb = self._lib.machip_ipc_blob_bytes()
        assert len(blobs) == nranks and all(len(b) == nb for b in blobs)
        buf = C.create_string_buffer(b"".join(blobs), nb * nranks)
        check(self._lib.machip_comm_init_ipc(self._h, int(rank), int(nranks), buf, float(timeout_s)))

    def comm_close_ipc(self):
        check(self._lib.machip_comm_close_ipc(self._h))

    def comm_drop_ipc(self):
        """Leave the inter-process communicator again (back to a single-rank handle)."""
        check(self._lib.machip_comm_drop_ipc(self._h))

    def comm_drop(self):
        """Leave any inter-process communicator (RCCL and / or IPC): single-rank handle again."""
        check(self._lib.machip_comm_drop(self._h))

    def comm_timing(self):
        """(gradient kernel us, exchange us) of the last sharded gradient (machip_comm_timing)."""
        a, b = C.c_double(0.0), C.c_double(0.0)
        check(self._lib.machip_comm_timing(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def solve_mode(self):
        """(mode, closures) of the last eigen-solve (machip_solve_mode)."""
        c = C.c_int64(0)
        return int(self._lib.machip_solve_mode(self._h, C.byref(c))), int(c.value)

    def set_option(self, name, value=None):
        """Entry `name` of this handle's option table (mac_amd/csrc/options.h; value None = the measured default).  Options the library
        reads when a handle is CREATED (CREATION_OPTIONS) are refused here -- on an existing handle they would silently do nothing: pass
        them through `with default_options(...)` around the constructor (MAC(options=...) does)."""
        if value is not None and name in CREATION_OPTIONS:
            raise ValueError(f"option {name!r} is read when the handle is created: set it with mac_amd._lib.default_options({name}=...) around the constructor")
        check(self._lib.machip_set_option(self._h, name.encode(), OPTION_AUTO if value is None else int(value)))

    def set_options(self, **opts):
        for k, v in opts.items():
            self.set_option(k, v)

    def get_option(self, name):
        v = C.c_int64(0)
        check(self._lib.machip_get_option(self._h, name.encode(), C.byref(v)))
        return None if v.value == OPTION_AUTO else v.value

    def set_solver(self, mode):
        """0 = automatic, 1 = Lanczos, 2 = preconditioned (LOBPCG + tridiagonal chain solve)."""
        check(self._lib.machip_set_solver(self._h, int(mode)))

    def eval_batch(self, X, tol=1e-8, max_steps=0):
        """lambda_2(L(x_b)) for every row of X (B x m), solved concurrently on the handle's evaluation lanes.
        Returns (lambda2[B], status[B])."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.ndim == 2 and X.shape[1] == self.m
        B = X.shape[0]
        lam = np.zeros(B)
        st = np.zeros(B, dtype=np.int32)
        check(self._lib.machip_eval_batch(self._h, B, p_f64(X), float(tol), int(max_steps), p_f64(lam),
                                          st.ctypes.data_as(C.POINTER(C.c_int))))
        return lam, st

    def fw_sweep(self, ks, X0, max_iters=5, gap_tol=1e-4, grad_tol=1e-8, tol=1e-8, max_steps=0, warm_start=False,
                 round_decimals=10, want_rounded=True):
        """Frank-Wolfe on B budgets of this graph at once (machip_fw_sweep).  Returns a dict of arrays: x (B x m),
        rounded (B x m or None), upper, f_traj (B x max_iters, NaN where a problem stopped early), iters, status."""
        X0 = np.ascontiguousarray(X0, dtype=np.float64)
        ks = np.ascontiguousarray(ks, dtype=np.int64)
        assert X0.ndim == 2 and X0.shape == (len(ks), self.m)
        B = len(ks)
        X = np.empty((B, self.m)); R = np.empty((B, self.m)) if want_rounded else None
        up = np.zeros(B); ft = np.full((B, max(1, max_iters)), np.nan)
        it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
        check(self._lib.machip_fw_sweep(self._h, B, ks.ctypes.data_as(C.POINTER(C.c_int64)), p_f64(X0), int(max_iters),
                                        float(gap_tol), float(grad_tol), float(tol), int(max_steps), int(bool(warm_start)),
                                        int(round_decimals), p_f64(X), p_f64(R), p_f64(up), p_f64(ft),
                                        it.ctypes.data_as(C.POINTER(C.c_int)), st.ctypes.data_as(C.POINTER(C.c_int))))
        return dict(x=X, rounded=R, upper=up, f_traj=ft[:, :max_iters], iters=it, status=st)

    def set_precision(self, precision):
        """0 = fp64 throughout, 1 = fp32 Krylov iterate + fp64 Rayleigh/residual refinement."""
        check(self._lib.machip_set_precision(self._h, int(precision)))

    def synchronize(self):
        check(self._lib.machip_synchronize(self._h))


class _stdout_to_stderr:
    """RCCL prints a version banner on stdout the first time it is touched; keep stdout clean for
    callers that emit machine-readable output (bench.py's single JSON line)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        os.dup2(self._saved, 1)
        os.close(self._saved)


def _single_node_bootstrap():
    """The ranks of one job share a node (one process per GPU, xGMI between them): keep RCCL's socket
    bootstrap on the loopback interface unless the caller chose one.  (On the test boxes the only other
    interface is a container veth, over which ncclCommInitRank was once seen to hang.)"""
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")


def comm_init_local(problems):
    """Join the given Problems (same edge lists, one per GPU or several on one GPU) as ranks 0..R-1 of an
    in-process communicator; afterwards each must be driven by its own host thread."""
    lib = load()
    arr = (C.c_void_p * len(problems))(*[p._h for p in problems])
    check(lib.machip_comm_init_local(arr, len(problems)))


def shard_plan(m, nranks, rank):
    """(lo, hi, shard) exactly as libmachip computes them (host arithmetic, no GPU needed)."""
    lo, hi, sh = C.c_int64(), C.c_int64(), C.c_int64()
    check(load().machip_shard_plan(int(m), int(nranks), int(rank), C.byref(lo), C.byref(hi), C.byref(sh)))
    return int(lo.value), int(hi.value), int(sh.value)


def comm_unique_id() -> bytes:
    _single_node_bootstrap()
    buf = C.create_string_buffer(128)
    lib = load()
    with _stdout_to_stderr():
        st = lib.machip_comm_unique_id(buf)
    check(st)
    return buf.raw


def fiedler_csr(indptr, indices, data, n, tol=1e-8, max_steps=0, x0=None, q=0, device=0):
    lib = load()           # (no device test here: the C entry point validates the matrix first and reports NO_DEVICE itself)
    indptr, indices, data = i32(indptr), i32(indices), f64(data)
    lam = C.c_double()
    v = np.empty(n)
    X = np.empty((q, n)) if q else None
    x0a = f64(x0) if x0 is not None else None
    stats = SolveStats()
    check(lib.machip_fiedler_csr(int(device), int(n), p_i32(indptr), p_i32(indices), p_f64(data), float(tol),
                                 int(max_steps), p_f64(x0a), C.byref(lam), p_f64(v), p_f64(X), int(q),
                                 C.byref(stats)))
    return lam.value, v, (X.T if X is not None else None), stats


def host_tridiag_smallest(a, b):
    lib = load()
    a, b = f64(a), f64(b)
    J = len(a)
    th = C.c_double()
    s = np.empty(J)
    check(lib.machip_host_tridiag_smallest(p_f64(a), p_f64(b), J, C.byref(th), p_f64(s)))
    return th.value, s


def host_follow_records(alpha, beta, l1, n, e_target, tiny_l=1.0, jcap=0, J=None):
    """The rule a Lanczos solve ends by (mac_amd/csrc/follow.h), applied on the host to a finished record array:
    alpha[0..J), beta[0..J], l1[0..J) -> (analysis points visited, order of the final tridiagonal or -1, estimate there)."""
    lib = load()
    alpha, beta, l1 = f64(alpha), f64(beta), f64(l1)
    J = int(len(alpha) if J is None else J)
    tri3 = np.zeros(3 * (J + 1))
    tri3[0:3 * J:3] = alpha[:J]; tri3[1:3 * (J + 1):3] = beta[:J + 1]; tri3[2:3 * J:3] = l1[:J]
    pts = (C.c_int * (J + 2))()
    npts, jeff, est = C.c_int(), C.c_int(), C.c_double()
    check(lib.machip_host_follow_records(p_f64(tri3), J, int(n), float(e_target), float(tiny_l), int(jcap), pts, J + 2,
                                         C.byref(npts), C.byref(jeff), C.byref(est)))
    return list(pts[:npts.value]), jeff.value, est.value
