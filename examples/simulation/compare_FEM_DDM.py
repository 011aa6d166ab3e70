"""FEM vs domain-decomposition solve of the reference's L-shaped cantilever preset (cf. the reference's
examples/simulation/compare_FEM_DDM.py; same preset file, same calls)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeSim.lattice_sim import LatticeSim                                   # noqa: E402
from pyLatticeSim.utils_simulation import solve_FEM_FenicsX                       # noqa: E402

path = "simulation/"
name_file = "Cantilever_L_beam"

start_time_FEM = time.time()
lattice_Sim_object = LatticeSim(path + name_file, verbose=1)
print("Lattice generation time --- %s seconds ---" % (time.time() - start_time_FEM))
sol_FEM = solve_FEM_FenicsX(lattice_Sim_object)[0]
print("FEM simulation time --- %s seconds ---" % (time.time() - start_time_FEM))

start_time_DDM = time.time()
lattice_object = LatticeSim(path + name_file, enable_domain_decomposition_solver=True, verbose=1)
print("Lattice generation time --- %s seconds ---" % (time.time() - start_time_DDM))
sol_DDM = lattice_object.solve_DDM()[0]
print("DDM simulation time --- %s seconds ---" % (time.time() - start_time_DDM))

relative_error = np.linalg.norm(sol_FEM - sol_DDM) / np.linalg.norm(sol_FEM)
print("Relative error between FEM and DDM", relative_error)
