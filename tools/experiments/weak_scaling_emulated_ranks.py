"""Iteration count the N-rank weak-scaling run would see with precond 3 / 4, emulated on ONE GPU: single-rank RCCL
communicator, the slab interface planes declared as shared nodes (tile level and local level leave them out, so the
local level decouples into N independent slab blocks exactly as on N ranks)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylatticedso_amd import lattice_arrays as LA, _capi
N = int(sys.argv[1])
lat = LA.generate((1, 1, 1), (50, 50 * N, 50), ["Octet"], [0.03])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed = np.zeros((lat.n_nodes, 6), np.uint8); fixed[lat.node_xyz[:, 0] == 0.0] = 1
tgt = lat.node_xyz[:, 0] == 50.0
f = np.zeros((lat.n_nodes, 6)); f[tgt, 2] = -0.1 / tgt.sum()
y = lat.node_xyz[:, 1]
shared = np.flatnonzero((np.abs(y / 50.0 - np.round(y / 50.0)) < 1e-9) & (y > 0) & (y < 50.0 * N))
print(f"N={N}: {lat.n_beams} struts, {len(shared)} shared nodes", flush=True)
for pc, lmax in ((3, 0), (4, 2200 * N)):
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                          precond=pc, palette=1, local_max_dofs=lmax, coarse_modes=int(os.environ.get("COARSE_MODES", "0")), coarse_max_dofs=int(os.environ.get("COARSE_DOFS", "3072")),
                          grid=((0.0, 0.0, 0.0), (50.0, 50.0 * N, 50.0), lat.n_nodes)) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
        dev.set_bc(fixed, None, f)
        t0 = time.time(); dev.assemble(); ta = time.time() - t0
        st = dev.solve(rtol=1e-8, max_iter=20000, download=False)
        st = st[-1] if isinstance(st, tuple) else st
        print(f"N={N} precond {pc}: iterations {st['iterations']} converged {st['converged']} "
              f"(device assembly {st['ms_assembly']:.2f} ms, solve {st['ms_solve']:.1f} ms)", flush=True)
