"""Iteration counts of the preconditioner variants on a small cantilever (GPU)."""
import sys, numpy as np
sys.path.insert(0, ".")
from pylatticedso_amd import lattice_arrays as LA, _capi
geom, n, r = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
tn = int(sys.argv[4]) if len(sys.argv) > 4 else 0
plane = float(sys.argv[5]) if len(sys.argv) > 5 else None      # pretend y = plane is a slab interface (RCCL, world 1)
cmax = int(sys.argv[6]) if len(sys.argv) > 6 else 0      # bound on the dofs of the global dense level
cmaxL = int(sys.argv[7]) if len(sys.argv) > 7 else 0     # precond 4: bound on the rank-local dense level
lat = LA.generate((1, 1, 1), (n, n, n), [geom], [r])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed = np.zeros((lat.n_nodes, 6), np.uint8); fixed[lat.node_xyz[:, 0] == 0.0] = 1
tgt = lat.node_xyz[:, 0] == float(n)
f = np.zeros((lat.n_nodes, 6)); f[tgt, 2] = -0.1 / tgt.sum()
u0 = None
for pc in (1, 2, 3, 4):
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                          precond=pc, tile_nodes=tn, coarse_max_dofs=cmax, local_max_dofs=cmaxL) as dev:
        if plane is not None:
            sh = np.flatnonzero(lat.node_xyz[:, 1] == plane)
            dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), sh, np.arange(len(sh)), len(sh))
        dev.set_bc(fixed, None, f); dev.assemble()
        u, st = dev.solve(rtol=1e-8, max_iter=20000)
        if u0 is None: u0 = u
        print(geom, n, "precond", pc, "its", st["iterations"], "conv", st["converged"], "rel diff", np.linalg.norm(u - u0) / np.linalg.norm(u0), flush=True)
