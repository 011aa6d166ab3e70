#!/usr/bin/env python3
"""Sweep of opts.tile_nodes (target nodes per K*p tile = block of the preconditioner's tile level) over cantilevers:
iterations, iteration time, assembly and solve times.  Usage: sweep_tiles.py [tile sizes, comma separated] [cases...]
with a case = GEOM:n:radius[:g] (g = graded radii per cell), GEOM may be BCC+Octet."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylatticedso_amd import _capi, lattice_arrays as LA  # noqa: E402

tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "256,192").split(",")]
cases = sys.argv[2:] or ["Octet:50:0.03", "Octet:32:0.03", "BCC:50:0.05", "BCC+Octet:40:0.04", "Octet:50:0.03:g"]
for case in cases:
    parts = case.split(":")
    geoms, n, r = parts[0].split("+"), int(parts[1]), float(parts[2])
    graded = len(parts) > 3
    radii = [r, 0.75 * r][:len(geoms)]
    lat = LA.generate((1, 1, 1), (n, n, n), geoms, radii)
    rad = lat.beam_radius.copy()
    if graded:   # smooth radius field, one value per cell (as bench.py's graded streaming workload)
        c = lat.node_xyz[lat.beam_conn].mean(axis=1)
        rad *= 0.8 + 0.4 * (np.floor(c[:, 0]) * 0.37 + np.floor(c[:, 1]) * 0.21 + np.floor(c[:, 2]) * 0.11) % 1.0
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, rad))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == float(n)
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    for tn in tiles:
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, rad, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                              precond=3, palette=1, tile_nodes=tn,
                              tile_modes=int(os.environ.get("TILE_MODES", "0")),
                              coarse_modes=int(os.environ.get("COARSE_MODES", "0"))) as dev:
            dev.set_bc(fixed, None, f)
            best = None
            for rep in range(3):
                dev.assemble()
                u, st = dev.solve(rtol=1e-6, max_iter=50000)
                tot = st["ms_assembly"] + st["ms_solve"]
                if best is None or tot < best[0]:
                    best = (tot, st)
            tot, st = best
            print(f"{case:22s} tile {tn:3d}: {st['iterations']:4d} iterations, assembly {st['ms_assembly']:6.2f} ms, solve "
                  f"{st['ms_solve']:7.2f} ms, iteration {1e3 * st['ms_solve'] / st['iterations']:6.1f} us, "
                  f"{lat.n_beams / tot / 1e3:6.1f} M beams/s", flush=True)
