"""BASELINE.json configs[3] / SURVEY.md 8(d) item 4: 24^3 BCC lattice with a "gyroid-like" graded radius field
(one radius per cell, unit_cell parameterisation), 50 objective + gradient evaluations with adjoint sensitivities on
one GPU, driven by a projected-gradient loop (the SciPy SLSQP driver of the reference also works, but with 13 824
parameters its own linear algebra dominates)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))
from pyLatticeOpti.lattice_opti import LatticeOpti      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
preset = {
    "geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n},
                 "radii": [0.05], "geom_types": ["BCC"]},
    "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
    "boundary_conditions": {
        "Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"],
                                   "Value": [0, 0, 0, 0, 0, 0]}},
        "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}},
    "optimization_informations": {
        "objective_function": "min", "objective_type": "compliance", "max_iterations": iters,
        "optimization_parameters": {"type": "unit_cell"}, "enable_parameter_normalization": True,
        "enable_gradient_computing": True, "simulation_type": "FEM"}}
t0 = time.perf_counter()
L = LatticeOpti(preset)
L._device = L.device_model(precond=3, palette=1)      # (LatticeOpti asks for warm_start = 4: the Galerkin start)
c = L._cell_center
r = np.clip(0.05 + 0.03 * (np.sin(2 * np.pi * c[:, 0] / 8) * np.cos(2 * np.pi * c[:, 1] / 8)
                           + np.sin(2 * np.pi * c[:, 1] / 8) * np.cos(2 * np.pi * c[:, 2] / 8)
                           + np.sin(2 * np.pi * c[:, 2] / 8) * np.cos(2 * np.pi * c[:, 0] / 8)) / 1.5, 0.01, 0.1)
theta = np.asarray(L.normalize_optimization_parameters(list(r)))
t_setup = time.perf_counter() - t0
vol0 = float((r ** 2).sum())
hist, t_loop = [], time.perf_counter()
step = 0.05
for it in range(iters):
    f = L.objective(list(theta))
    g = L.gradient(list(theta))
    hist.append(L.denorm_objective)
    # projected gradient at (approximately) constant strut volume
    theta = np.clip(theta - step * g / max(np.abs(g).max(), 1e-30), 0.0, 1.0)
    rr = np.asarray(L.denormalize_optimization_parameters(list(theta)))
    rr *= np.sqrt(vol0 / float((rr ** 2).sum()))
    theta = np.clip((np.clip(rr, 0.01, 0.1) - 0.01) / 0.09, 0.0, 1.0)
t_loop = time.perf_counter() - t_loop
print(json.dumps({"workload": f"{n}^3 BCC graded radius, unit_cell parameterisation ({L.number_parameters} parameters)",
                  "struts": L.lattice.n_beams, "iterations": iters, "setup_s": t_setup, "loop_s": t_loop,
                  "s_per_objective_plus_gradient": t_loop / iters, "compliance_first": hist[0],
                  "compliance_last": hist[-1], "last_pcg_iterations": L._model.stats["iterations"]}))
