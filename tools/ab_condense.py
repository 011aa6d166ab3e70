#!/usr/bin/env python3
"""A/B of opts.condense on cantilevers: iterations, solve time, agreement of the two solutions."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylatticedso_amd import _capi, lattice_arrays as LA  # noqa: E402

cases = [("BCC", 16, 0.05), ("BCC", 50, 0.05), ("Octet", 24, 0.03)] if len(sys.argv) < 2 else \
    [(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]))]
for geom, n, r in cases:
    lat = LA.generate((1, 1, 1), (n, n, n), [geom], [r])
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == float(n)
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    out = {}
    for cond in (-1, 0 if os.environ.get('AUTO') else 1):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                              precond=3, palette=1, condense=cond,
                              tile_nodes=int(os.environ.get('TILE', '0'))) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-8, max_iter=50000)
            u, st = dev.solve(rtol=1e-8, max_iter=50000)
            res = np.where(fixed != 0, 0.0, f - dev.spmv(u))
            out[cond] = (u, st, np.linalg.norm(res) / np.linalg.norm(f), dev.time_kernel(3, 20))
    (u0, s0, r0, t0), (u1, s1, r1, t1) = out[-1], out[0 if os.environ.get('AUTO') else 1]
    print(f"{geom} {n}^3: plain {s0['iterations']} its {s0['ms_solve']:.1f} ms ({t0 * 1e3:.0f} us/it, true res {r0:.1e}) | "
          f"condensed ({int(s1['condensed_nodes'])} of {lat.n_nodes} nodes) {s1['iterations']} its {s1['ms_solve']:.1f} ms "
          f"({t1 * 1e3:.0f} us/it, true res {r1:.1e}) | rel diff {np.linalg.norm(u1 - u0) / np.linalg.norm(u0):.1e}",
          flush=True)
