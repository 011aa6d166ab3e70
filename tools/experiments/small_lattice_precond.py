"""Which solver serves the SMALL lattices of the reference's own presets (6x3x3 ... 16x8x8 cells) best: Jacobi PCG (what
LatticeSim.device_model picked below 20 000 nodes until round 5) or the multi-level PCG with the short iteration?
Prints assembly + solve time and iterations per size.   python tools/experiments/small_lattice_precond.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pylatticedso_amd import _capi, lattice_arrays as LA  # noqa: E402

E, NU = 1013.0, 0.3
for geom, cells in [("BCC", (6, 3, 3)), ("BCC", (8, 4, 4)), ("BCC", (10, 5, 5)), ("BCC", (12, 6, 6)), ("BCC", (16, 8, 8)), ("Octet", (6, 3, 3)), ("Octet", (10, 5, 5)),
                    ("Octet", (12, 12, 12)), ("BCC", (16, 16, 16))]:
    lat = LA.generate((1, 1, 1), cells, [geom], [0.05 if geom == "BCC" else 0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    tgt = (lat.node_xyz[:, 0] == float(cells[0])) | (lat.node_xyz[:, 2] == float(cells[2]))
    f[tgt, 2] = -0.1 / tgt.sum()
    row = f"{geom:5s} {cells[0]:2d}x{cells[1]:2d}x{cells[2]:2d} {lat.n_beams:6d} struts {lat.n_nodes:6d} nodes |"
    variants = [("Jacobi", dict(precond=1)), ("multi-level", dict(precond=3, palette=1)),
                ("ml tiles 16", dict(precond=3, palette=1, tile_nodes=16)),
                ("ml tiles 32", dict(precond=3, palette=1, tile_nodes=32)),
                ("ml tiles 64", dict(precond=3, palette=1, tile_nodes=64)),
                ("ml tiles 32, 12 dense modes", dict(precond=3, palette=1, tile_nodes=32, coarse_modes=12))]
    if 6 * lat.n_nodes <= 16384:
        variants.append(("dense factor (precond 5)", dict(precond=5)))
    for name, kw in variants:
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, **kw) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            dev.solve(rtol=1e-8, max_iter=100000, download=False)
            t0 = time.perf_counter()
            for _ in range(5):
                dev.assemble()
                st = dev.solve(rtol=1e-8, max_iter=100000, download=False)
            dt = (time.perf_counter() - t0) / 5
            row += f" {name}: {dt * 1e3:6.2f} ms ({st['iterations']:4d} its, asm {st['ms_assembly']:.2f} + solve {st['ms_solve']:.2f}) |"
    print(row, flush=True)
