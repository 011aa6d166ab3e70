import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pylatticedso_amd import _capi, lattice_arrays as LA
import bench
n = 12
lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed, f, _ = bench.cantilever_bc(lat.node_xyz, float(n))
for cond in (0, 1):
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, bench.E, bench.NU,
                          precond=3, palette=1, warm_start=4, condense=cond, tile_nodes=64, coarse_max_dofs=600) as dev:
        dev.set_bc(fixed, None, f)
        its = []
        for k in range(9):
            if k == 5:
                dev.set_bc(fixed, None, 2.0 * f)          # another right-hand side: twice the load
            if k == 7:
                g = np.zeros_like(f); g[lat.node_xyz[:, 0] == float(n), 1] = 1e-3
                dev.set_bc(fixed, None, g)                # an unrelated one
            dev.assemble()
            u, st = dev.solve(rtol=1e-9, max_iter=20000)
            assert st["converged"] == 1 and np.isfinite(u).all()
            its.append(int(st["iterations"]))
        print("condense", cond, "iterations", its)
