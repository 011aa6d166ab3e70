"""What one column of pl_schur costs: the same three calls from Python, timed."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_sim import LatticeSim
from pylatticedso_amd.utils_schur import node_order_to_simulate
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                       "radii": [0.05], "geom_types": ["BCC"]},
          "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": True}, "boundary_conditions": {}}
L = LatticeSim(preset)
order = node_order_to_simulate(L, 0)
dev = L.device_model()
dev.assemble()
N = dev.n_nodes
fixed = np.zeros((N, 6), np.uint8); fixed[order] = 1
ubar = np.zeros((N, 6))
t = {"set_bc": 0.0, "solve": 0.0, "reactions": 0.0}
its = []
for rep in range(2):
    for j in range(48):
        ubar[:] = 0; ubar[order[j // 6], j % 6] = 1.0
        a = time.perf_counter(); dev.set_bc(fixed, ubar, None); b = time.perf_counter()
        u, st = dev.solve(rtol=1e-13, max_iter=1000); c = time.perf_counter()
        R = dev.reactions(u); d = time.perf_counter()
        if rep:
            t["set_bc"] += b - a; t["solve"] += c - b; t["reactions"] += d - c; its.append(int(st["iterations"]))
print({k: round(1e3 * v / 48, 3) for k, v in t.items()}, "ms per column; iterations", sorted(set(its)), "nodes", N, "precond_used", st["precond_used"], "device solve ms", st["ms_solve"], "assembly", st["ms_assembly"])
