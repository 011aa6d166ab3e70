"""BASELINE.json configs[4]-like workload (BCC + Octet superposed per cell, r = [0.04, 0.03]) on ONE GPU in fp64:
how large a hybrid plate the single-GPU path takes.  Usage: hybrid_capacity.py nx ny nz"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylatticedso_amd import lattice_arrays as LA, _capi
nx, ny, nz = (int(v) for v in sys.argv[1:4])
t0 = time.time()
lat = LA.generate((1, 1, 1), (nx, ny, nz), ["BCC", "Octet"], [0.04, 0.03])
pen = LA.penalize(lat, LA.compute_lzone(lat))
print(f"{nx}x{ny}x{nz} BCC+Octet: {lat.n_beams} struts, {lat.n_nodes} nodes generated in {time.time()-t0:.0f} s", flush=True)
fixed = np.zeros((lat.n_nodes, 6), np.uint8); fixed[lat.node_xyz[:, 0] == 0.0] = 1
tgt = lat.node_xyz[:, 0] == float(nx)
f = np.zeros((lat.n_nodes, 6)); f[tgt, 2] = -0.1 / tgt.sum()
t0 = time.time()
with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                      precond=3, palette=1) as dev:
    print(f"handle created in {time.time()-t0:.1f} s", flush=True)
    dev.set_bc(fixed, None, f); dev.assemble()
    st = dev.solve(rtol=1e-8, max_iter=20000, download=False)
    st = st[-1] if isinstance(st, tuple) else st
    print(f"assembly {st['ms_assembly']:.1f} ms, solve {st['ms_solve']:.1f} ms, {st['iterations']} iterations, converged "
          f"{st['converged']}: {lat.n_beams / (st['ms_assembly'] + st['ms_solve']) / 1e3:.1f} M beams/s; "
          f"K*p {dev.time_kernel(0, 10)*1e3:.0f} us, PCG iteration {dev.time_kernel(3, 10)*1e3:.0f} us", flush=True)
