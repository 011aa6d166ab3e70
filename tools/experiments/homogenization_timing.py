"""Time of the six periodic solves of get_homogenized_properties per unit cell (device path) and their PCG iteration counts."""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_sim import LatticeSim
from pylatticedso_amd.utils_simulation import get_homogenized_properties
for geoms, radii in ((["BCC"], [0.05]), (["Octet"], [0.05]), (["Kelvin"], [0.05]), (["BCC", "Hybrid1", "Hybrid4"], [0.05, 0.04, 0.03])):
    preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                           "radii": radii, "geom_types": geoms},
              "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": True}, "boundary_conditions": {}}
    ts = []
    for rep in range(3):
        L = LatticeSim(preset)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            S, an = get_homogenized_properties(L)
        ts.append(time.perf_counter() - t0)
    print(geoms, "nodes", L.lattice.n_nodes, "ms", [round(1e3 * t, 1) for t in ts], "pcg iterations", an.pcg_iterations)
