"""The whole surrogate tool chain of the reference on the GPU build, for its own preset
``optimization/optimization_DDM_surrogate`` (5 x 1 x 1 cells of BCC + Hybrid1 + Hybrid4, RBF surrogate, unit_cell
parameterisation, compliance) - whose reduced basis is one of the large files the reference does not ship:

  1. dataset of exact cell Schur complements over a grid of the three radii   (construct_schur_complement_dataset)
  2. greedy reduced basis of that dataset                                     (reduce_basis_schur_with_greedy)
  3. ``LatticeOpti`` with ``simulation_type: "DDM"``: device solve_DDM, gradients from the spline's dS/dr, SLSQP.

Usage: python optimization_DDM_surrogate_chain.py [samples per radius = 4] [SLSQP iterations = 10]
"""
import os
import sys
import time
from itertools import product

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeOpti.lattice_opti import LatticeOpti                                                  # noqa: E402
from pyLatticeSim.greedy_algorithm import find_name_file_reduced_basis, reduce_basis_greedy        # noqa: E402
from pyLatticeSim.lattice_sim import LatticeSim, open_lattice_parameters                           # noqa: E402
from pyLatticeSim.utils_schur import get_schur_complement                                          # noqa: E402

n_samples = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 10
name_file = "optimization/optimization_DDM_surrogate"
preset = open_lattice_parameters(name_file)
tol = preset["simulation_parameters"]["DDM"]["schur_complement_computation"]["precision_greedy"]
# The shipped preset enables the preconditioner without naming its type, which LatticeSim rejects - reference and
# mirror alike (lattice_sim.py:219-221).  "exact" = the assembled matrix of the cells' own Schur complements, factorised
# on the device: one CG step while that matrix is positive definite; where the spline surrogate leaves its training
# range and turns indefinite the device falls back to Jacobi CG and lifts the preset's cap of 10 iterations.
preset["simulation_parameters"]["DDM"].setdefault("preconditioner_type", "exact")

# 1. one periodic cell of the same geometry, exact Schur complements on the GPU
cell = {"geometry": dict(preset["geometry"], number_of_cells={"x": 1, "y": 1, "z": 1}),
        "simulation_parameters": {"enable": True, "material": preset["simulation_parameters"]["material"],
                                  "periodicity": True}}
t0 = time.time()
one = LatticeSim(cell)
grid = np.round(np.linspace(0.01, 0.1, n_samples), 4)
data = {}
for radii in product(grid, repeat=len(one.geom_types)):
    one.reset_cell_with_new_radii(list(radii))
    data[tuple(float(r) for r in radii)] = get_schur_complement(one)
print(f"dataset: {len(data)} Schur complements of {next(iter(data.values())).shape} in {time.time() - t0:.1f} s")

# 2. greedy reduced basis, stored where the surrogate modes look for it
t0 = time.time()
out = reduce_basis_greedy(data, tol, find_name_file_reduced_basis(one, tol), verbose=0)
print(f"reduced basis: {out[3].shape[1]} vectors in {time.time() - t0:.1f} s")

# 3. the optimisation itself
t0 = time.time()
lattice_object = LatticeOpti(preset, verbose=0)
lattice_object.redefine_optim_parameters(max_iteration=n_iter, disp=False)
sol = lattice_object.optimize_lattice()
hist = lattice_object._history["objective"]
print(f"optimisation: {sol.nit} SLSQP iterations, {sol.nfev} objective + {sol.njev} gradient evaluations in "
      f"{time.time() - t0:.1f} s; compliance {hist[0]:.4e} -> {hist[-1]:.4e}; relative density "
      f"{lattice_object.relative_density():.3f}; last solve_DDM: {lattice_object.iteration} CG iterations")
