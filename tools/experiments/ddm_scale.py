import os, sys, time, json, numpy as np
PC = len(sys.argv) > 3
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from pylatticedso_amd.lattice_sim import LatticeSim
n = int(sys.argv[1])
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n}, "radii": [0.05], "geom_types": ["BCC"]},
 "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False,
   "DDM": {"enable_preconditioner": PC, "preconditioner_type": "mean", "max_iterations": 20000, "schur_complement_computation": {"type": "RBF", "precision_greedy": 1e-6}}},
 "boundary_conditions": {"Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X","Y","Z","RX","RY","RZ"], "Value": [0]*6}},
                         "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
t0 = time.time()
L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")])
t1 = time.time()
cb = L.cell_boundary_nodes()
t2 = time.time()
print(f"n={n}: LatticeSim {t1-t0:.1f}s, cell_boundary_nodes {t2-t1:.1f}s, cells {L.lattice.n_cells}", flush=True)
if len(sys.argv) > 2:
    xsol, info, idx, b = L.solve_DDM()
    t3 = time.time()
    dev = L.ddm_model()
    print(f"solve_DDM {t3-t2:.2f}s its {L.iteration} info {info}; operator apply {dev.time_kernel(0, 50)*1e3:.1f} us; CG iteration {dev.time_kernel(3, 50)*1e3:.1f} us")
