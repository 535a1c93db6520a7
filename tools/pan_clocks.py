#!/usr/bin/env python3
"""Phase clocks of the column-panel matrix kernel (k_pan_mul / k_pan_mul8) inside a real solve: needs the developer build
   MACHIP_BUILD_FLAGS=-DPAN_CLOCKS MACHIP_BUILD_OUT=libmachip_clk.so bash mac_amd/csrc/build.sh;  MACHIP_LIB=mac_amd/libmachip_clk.so python tools/pan_clocks.py c4 19
Every workgroup leaves 16 stamps of the 100 MHz wall clock; the table is that of the LAST launch of the solve on iterate <it> of the Frank-Wolfe run.
usage: pan_clocks.py [c4] [iterate] [name=value options ...]"""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, bench
from mac_amd import _lib
from mac_amd.utils.fiedler import reference_start_block
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
it_at = int(sys.argv[2]) if len(sys.argv) > 2 else 19
sets = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in sys.argv[3:]] or [dict(panel=1)]
w = bench.make_workload(cfg)
P = _lib.Problem(w["n"], w["fi"], w["fj"], w["fw"], w["ci"], w["cj"], w["cw"])
P.set_start(reference_start_block(w["n"])[:, 0].copy())
P.set_x(w["x0"])
for it in range(it_at):
    P.fw_step(w["k"], it); P.fw_commit()
lib = _lib.load()
lib.machip_debug_pan_clocks.restype = C.c_int
lib.machip_debug_pan_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
names = {0: "workgroup entry (wave 0)", 1: "wave 1 entry", 2: "record form: all of wave 1's loads arrived / u form: operand in LDS, tile table known", 3: "prologue done (record form: wave 0)",
         4: "barrier 1 passed (record form)", 5: "operand in LDS, barrier passed", 6: "first chunks multiplied (wave 1)", 7: "tiles accumulated (wave 1)",
         8: "wave 1 done (stores issued)", 9: "wave 15 done",
         10: "ROW KERNEL (last launch; its own time base): workgroup entry", 11: "row kernel: row loads + coefficients arrived", 12: "row kernel: rows done, stores issued",
         13: "row kernel: six sums reduced and stored", 14: "row kernel: every store acknowledged"}
for o in sets:
    keys = list(o)
    for k_, v in o.items(): P.set_option(k_, v)
    P.assemble()
    for rep in range(2): lam, _, _ = P.fiedler(want_vec=False)
    st = P.stats
    out8 = (C.c_int * 12)()
    mode = P.solve_mode()
    g = 252
    buf = (C.c_longlong * (16 * 1024))()
    _lib.check(lib.machip_debug_pan_clocks(P._h, buf, 1024))      # (drop what earlier launches left)
    lam, _, _ = P.fiedler(want_vec=False)
    _lib.check(lib.machip_debug_pan_clocks(P._h, buf, 1024))
    c = np.frombuffer(buf, dtype=np.int64).reshape(1024, 16)
    live = c[:, 0] > 0
    c = c[live]
    t0 = c[:, 0].min()
    print(f"== {o}: nnz {int(st.nnz)} steps {int(st.lanczos_steps)} in-solve {1e3 * st.step_ms / max(1, st.steps_timed):.2f} us per step, {len(c)} workgroups")
    live2 = c[:, 10] > 0
    t1 = c[live2, 10].min() if live2.any() else 0
    if live2.any(): print(f"      (row kernel's first workgroup enters {(t1 - t0) * 0.01:.2f} us after the matrix kernel's first)")
    for i in range(15):
        v = (c[:, i] - (t0 if i < 10 else t1)) * 0.01
        if (c[:, i] > 0).any():
            v = v[c[:, i] > 0]
            print(f"      {names[i]:96s} min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f} us")
    for k_ in keys: P.set_option(k_, None)
