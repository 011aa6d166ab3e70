"""BASELINE.json configs[4] workload (BCC + Octet superposed per cell, r = [0.04, 0.03], cantilever) on ONE GPU:
fp64 and the fp32 solver modes (opts.precision).  Usage: hybrid_capacity.py nx ny nz [precision ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylatticedso_amd import lattice_arrays as LA, _capi   # noqa: E402

nx, ny, nz = (int(v) for v in sys.argv[1:4])
modes = [int(v) for v in sys.argv[4:]] or [0, 1]
t0 = time.time()
lat = LA.generate((1, 1, 1), (nx, ny, nz), ["BCC", "Octet"], [0.04, 0.03])
pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
t_host = time.time() - t0
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
tgt = lat.node_xyz[:, 0] == float(nx)
f = np.zeros((lat.n_nodes, 6))
f[tgt, 2] = -0.1 / tgt.sum()
out = {"workload": f"{nx}x{ny}x{nz} BCC+Octet r=[0.04,0.03] cantilever", "struts": lat.n_beams, "nodes": lat.n_nodes,
       "host_build_s": t_host, "runs": []}
for precision in modes:
    t0 = time.time()
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                          precond=3, palette=1, precision=precision,
                          tile_modes=int(os.environ.get("TILE_MODES", "0")),
                          coarse_modes=int(os.environ.get("COARSE_MODES", "0")),
                          coarse_max_dofs=int(os.environ.get("COARSE_DOFS", "0"))) as dev:
        t_create = time.time() - t0
        dev.set_bc(fixed, None, f)
        dev.assemble()
        st = dev.solve(rtol=1e-8, max_iter=20000, download=False)
        dev.assemble()
        st = dev.solve(rtol=1e-8, max_iter=20000, download=False)
        ms = st["ms_assembly"] + st["ms_solve"]
        out["runs"].append({"precision": precision, "pl_create_s": t_create, "assembly_ms": st["ms_assembly"],
                            "solve_ms": st["ms_solve"], "iterations": st["iterations"], "inner_solves": st["restarts"],
                            "converged": st["converged"], "rel_residual": st["rel_residual"],
                            "beams_per_s": lat.n_beams / ms * 1e3, "spmv_us": dev.time_kernel(0, 10) * 1e3,
                            "pcg_iteration_us": dev.time_kernel({0: 3, 1: 8, 2: 9}[precision], 10) * 1e3})
    print(json.dumps(out["runs"][-1]), flush=True)
print(json.dumps(out))
