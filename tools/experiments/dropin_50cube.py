import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))
from pyLatticeSim.lattice_sim import LatticeSim
from pyLatticeSim.utils_simulation import solve_FEM_FenicsX
n = int(sys.argv[1])
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n}, "radii": [0.03], "geom_types": ["Octet"]},
 "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
 "boundary_conditions": {"Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X","Y","Z","RX","RY","RZ"], "Value": [0]*6}},
                         "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
t0 = time.time(); L = LatticeSim(preset); t1 = time.time()
xsol, model = solve_FEM_FenicsX(L); t2 = time.time()
print(f"{n}^3 Octet through the drop-in API: LatticeSim {t1-t0:.1f} s, solve_FEM_FenicsX {t2-t1:.2f} s ({model.stats['iterations']} PCG iterations, device solve {model.stats['ms_solve']:.1f} ms), {L.lattice.n_beams} struts, len(xsol) {len(xsol)}")
