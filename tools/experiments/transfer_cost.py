"""Host <-> device costs of the drop-in calls at 50^3 Octet: set_bc, solve with / without download, reactions."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd import _capi, lattice_arrays as LA
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed, f, _ = bench.cantilever_bc(lat.node_xyz, float(n))
with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, bench.E, bench.NU, precond=3, palette=1) as dev:
    def T(fn, reps=5):
        out = []
        for _ in range(reps):
            t = time.perf_counter(); r = fn(); out.append(round(1e3 * (time.perf_counter() - t), 2))
        return out, r
    ubar = np.zeros_like(f)
    print("set_bc (fixed, ubar, f)", T(lambda: dev.set_bc(fixed, ubar, f))[0])
    print("set_bc (fixed, None, f)", T(lambda: dev.set_bc(fixed, None, f))[0])
    dev.assemble()
    t, r = T(lambda: dev.solve(rtol=1e-8, max_iter=5000, download=False))
    print("solve, no download", t, "device ms", dev.last_stats["ms_solve"])
    t, (u, st) = T(lambda: dev.solve(rtol=1e-8, max_iter=5000))
    print("solve + download", t, "device ms", st["ms_solve"])
    print("reactions", T(lambda: dev.reactions(u))[0])
    print("spmv", T(lambda: dev.spmv(u))[0])
