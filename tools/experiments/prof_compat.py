import cProfile, pstats, sys, time
sys.path.insert(0, '.')
from pylatticedso_amd.lattice_sim import LatticeSim
from pylatticedso_amd.utils_simulation import solve_FEM_FenicsX
n = 50
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n}, "radii": [0.03], "geom_types": ["Octet"]},
          "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
          "boundary_conditions": {"Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"], "Value": [0, 0, 0, 0, 0, 0]}},
                                  "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
L0 = LatticeSim(preset); del L0
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
L = LatticeSim(preset, reference_compat=True)
t1 = time.perf_counter()
xsol, model = solve_FEM_FenicsX(L)
t2 = time.perf_counter()
pr.disable()
print("LatticeSim", t1 - t0, "solve", t2 - t1)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
