#!/usr/bin/env python3
"""Developer probe: eigen-solve cost on a synthetic stiff chain-like graph (long odometry chain, few
closures) -- wall time vs device time vs Lanczos steps, to see how much of a solve is host analysis."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from mac_amd import _lib

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 15000
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    rng = np.random.default_rng(1)
    fi = np.arange(n - 1, dtype=np.int32)
    fw = rng.uniform(100, 1000, n - 1)
    a = rng.integers(0, n, nc); b = np.clip(a + rng.integers(-3000, 3000, nc), 0, n - 1)
    keep = np.abs(a - b) > 1
    ci = np.minimum(a, b)[keep].astype(np.int32); cj = np.maximum(a, b)[keep].astype(np.int32)
    cw = rng.uniform(100, 300, len(ci))
    P = _lib.Problem(n, fi, fi + 1, fw, ci, cj, cw)
    P.set_x(np.ones(len(ci)))
    P.assemble()
    print('replay us/launch, bytes:', P.profile_spmv(400), flush=True)
    for env in ({}, {"MACHIP_SCHED": "0"}, {"MACHIP_DEBUG": "1"}):
        os.environ.pop("MACHIP_SCHED", None); os.environ.pop("MACHIP_DEBUG", None); os.environ.update(env)
        for rep in range(2):
            t0 = time.perf_counter(); lam, _, _ = P.fiedler(want_vec=False); dt = time.perf_counter() - t0
            s = P.stats
            print(f"{env} lam={lam:.9e} steps={s.lanczos_steps} restarts={s.restarts} wall={dt*1e3:.1f} ms gpu={s.gpu_ms:.1f} ms "
                  f"-> {dt*1e6/s.lanczos_steps:.2f} us/step res={s.residual:.2e}", flush=True)

if __name__ == "__main__":
    main()
