"""Domain-decomposition solves of the reference's own DDM presets (cf. its examples/simulation/
domain_decomposition_example.py and domain_decomposition_surrogate_example.py): surrogate cell Schur complements,
the assembled-Schur CG preconditioner factorised on the device, solve_DDM on the device.

Both presets use BCC + Hybrid1 + Hybrid4 cells, whose reduced basis is one of the large files the reference does not
ship: examples/optimization/optimization_DDM_surrogate_chain.py builds it (dataset -> greedy basis) - run that first,
or point $PYLATTICE_DATA_ROOT at a reference checkout that has the file.

Usage: python domain_decomposition_example.py [simulation/Three_point_bending | simulation/simulation_DDM_surrogate]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeDesign.plotting_lattice import LatticePlotting                      # noqa: E402
from pyLatticeSim.lattice_sim import LatticeSim                                   # noqa: E402

name_file = sys.argv[1] if len(sys.argv) > 1 else "simulation/Three_point_bending"

t0 = time.time()
try:
    solver_DDM = LatticeSim(name_file, verbose=1, enable_domain_decomposition_solver=True)
except FileNotFoundError as err:
    sys.exit(f"{err}\nrun examples/optimization/optimization_DDM_surrogate_chain.py first (it builds the reduced basis).")
t1 = time.time()
xsol, info, _, b = solver_DDM.solve_DDM()
stats = solver_DDM.ddm_model().last_stats
print(f"{solver_DDM.get_number_cells()} cells, {len(b)} free boundary dofs: lattice + Schur complements {t1 - t0:.2f} s, "
      f"solve_DDM {time.time() - t1:.2f} s ({solver_DDM.iteration} CG iterations, preconditioner "
      f"{int(stats['precond_used'])}, info {info}), max |u| = {np.abs(xsol).max():.4e}")

vizualizer = LatticePlotting()
print("plot:", vizualizer.visualize_lattice(solver_DDM, beam_color_type="radii", deformed_form=True,
                                            enable_boundary_conditions=True,
                                            domain_decomposition_simulation_plotting=True))
