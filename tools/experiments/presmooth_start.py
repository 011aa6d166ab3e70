"""Would PCG iterations run WITHOUT the dense level (while its factorisation is still in flight on another stream) pay?
m Jacobi-PCG iterations first, then the multi-level PCG on the remaining residual to the same absolute tolerance:
how many multi-level iterations are saved?  50^3 Octet cantilever."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylatticedso_amd import _capi, lattice_arrays as LA   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
tgt = lat.node_xyz[:, 0] == float(n)
f = np.zeros((lat.n_nodes, 6))
f[tgt, 2] = -0.1 / tgt.sum()
args = (lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3)
rtol = 1e-8
with _capi.HipLattice(*args, precond=3, palette=1) as full, _capi.HipLattice(*args, precond=1, palette=1) as jac:
    full.set_bc(fixed, None, f)
    full.assemble()
    u, st = full.solve(rtol=rtol, max_iter=5000)
    print(f"multi-level from zero: {st['iterations']} iterations", flush=True)
    jac.set_bc(fixed, None, f)
    jac.assemble()
    for m in (5, 10, 20, 40):
        um, _ = jac.solve(rtol=1e-30, max_iter=m, raise_on_noconv=False)
        r = np.where(fixed != 0, 0.0, f - full.spmv(um))
        full.set_bc(fixed, None, r)
        rt = rtol * np.linalg.norm(f) / np.linalg.norm(r)
        d, st2 = full.solve(rtol=rt, max_iter=5000)
        res = np.where(fixed != 0, 0.0, f - full.spmv(um + d))
        print(f"{m:3d} Jacobi iterations first (|r| = {np.linalg.norm(r) / np.linalg.norm(f):.3f} |f|): then "
              f"{st2['iterations']} multi-level iterations, final residual {np.linalg.norm(res) / np.linalg.norm(f):.2e}",
              flush=True)
