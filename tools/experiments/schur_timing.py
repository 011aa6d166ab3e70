"""Time of one exact cell condensation (get_schur_complement / calculate_schur_complement_cells) per geometry."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_sim import LatticeSim
from pylatticedso_amd.utils_schur import get_schur_complement
g = np.load(os.path.join(ROOT, "tests", "golden", "Schur_complement_BCC.npz"))
for geoms, radii in ((["BCC"], [0.05]), (["Hybrid1"], [0.05]), (["BCC", "Hybrid1", "Hybrid4"], [0.05, 0.04, 0.03])):
    preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                           "radii": radii, "geom_types": geoms},
              "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": True}, "boundary_conditions": {}}
    ts = []
    for rep in range(3):
        L = LatticeSim(preset)
        t0 = time.perf_counter(); S = get_schur_complement(L); ts.append(time.perf_counter() - t0)
        L._device.close() if getattr(L, "_device", None) is not None else None
    err = None
    if geoms == ["BCC"]:
        ref = g["schur_matrices"][4]
        err = float(np.abs(S - ref).max() / np.abs(ref).max())
    print(geoms, "S", S.shape, "ms", [round(1e3 * t, 1) for t in ts], "rel err vs reference dataset", err, "asym", float(np.abs(S - S.T).max() / np.abs(S).max()))
