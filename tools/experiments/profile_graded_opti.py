"""cProfile of LatticeOpti.objective + gradient on the graded 24^3 BCC lattice (configs[3] through the drop-in layer)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_opti import LatticeOpti
from pylatticedso_amd.timing import timing
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
preset = {
    "geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n},
                 "radii": [0.05], "geom_types": ["BCC"]},
    "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
    "boundary_conditions": {
        "Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"], "Value": [0] * 6}},
        "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}},
    "optimization_informations": {
        "objective_function": "min", "objective_type": "compliance", "max_iterations": 10,
        "optimization_parameters": {"type": "unit_cell"}, "enable_parameter_normalization": True,
        "enable_gradient_computing": True, "simulation_type": "FEM"}}
L = LatticeOpti(preset)
rng = np.random.default_rng(0)
theta = 0.3 + 0.3 * rng.random(L.number_parameters)
for _ in range(3):
    L.objective(list(theta)); L.gradient(list(theta)); theta = np.clip(theta + 0.01 * rng.standard_normal(len(theta)), 0, 1)
timing.reset()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(10):
    L.objective(list(theta)); L.gradient(list(theta)); theta = np.clip(theta + 0.01 * rng.standard_normal(len(theta)), 0, 1)
dt = time.perf_counter() - t0
pr.disable()
print("ms per objective + gradient: %.2f" % (1e3 * dt / 10))
top = sorted(((sum(v), k, len(v)) for k, v in timing.timings.items()), reverse=True)[:10]
for t, k, c in top:
    print("    %8.1f ms  %5d x  %s" % (1e3 * t, c, k))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue())
