"""Cold-start cost of the two-level preconditioner's rocSOLVER factorisation on a fresh box (no torch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pylatticedso_amd import _capi, lattice_arrays as LA
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed = np.zeros((lat.n_nodes, 6), np.uint8); fixed[lat.node_xyz[:, 0] == 0.0] = 1
f = np.zeros((lat.n_nodes, 6)); f[lat.node_xyz[:, 0] == float(n), 2] = -0.1
t0 = time.time()
d = _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3, precond=2)
print("create", round(time.time() - t0, 2), flush=True)
d.set_bc(fixed, None, f)
for rep in range(3):
    t0 = time.time(); d.assemble(); print(f"assemble #{rep}", round(time.time() - t0, 3), flush=True)
t0 = time.time(); st = d.solve(rtol=1e-8, download=False); print("solve", round(time.time() - t0, 3), st["iterations"], flush=True)
