"""Iteration counts of the weak-scaling workloads (50 x 50N x 50 Octet) solved on ONE GPU with the settings each rank
would use (global brick grid, coarse_max_dofs 3072)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylatticedso_amd import lattice_arrays as LA, _capi
N = int(sys.argv[1]); pc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.time()
lat = LA.generate((1, 1, 1), (50, 50 * N, 50), ["Octet"], [0.03])
pen = LA.penalize(lat, LA.compute_lzone(lat))
print(f"N={N}: {lat.n_beams} struts generated in {time.time()-t0:.0f} s", flush=True)
fixed = np.zeros((lat.n_nodes, 6), np.uint8); fixed[lat.node_xyz[:, 0] == 0.0] = 1
tgt = lat.node_xyz[:, 0] == 50.0
f = np.zeros((lat.n_nodes, 6)); f[tgt, 2] = -0.1 / tgt.sum()
with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                      precond=pc, palette=1) as dev:
    dev.set_bc(fixed, None, f); dev.assemble()
    st = dev.solve(rtol=1e-8, max_iter=20000, download=False)
    st = st[-1] if isinstance(st, tuple) else st
    print(f"N={N} precond {pc}: iterations {st['iterations']} converged {st['converged']}; PCG iteration {dev.time_kernel(3, 20)*1e3:.1f} us", flush=True)
