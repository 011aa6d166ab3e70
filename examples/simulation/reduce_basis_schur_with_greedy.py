"""Reduced basis of a Schur-complement dataset by the greedy algorithm (cf. the reference's
examples/simulation/reduce_basis_schur_with_greedy.py).  Run construct_schur_complement_dataset.py first."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeSim.greedy_algorithm import find_name_file_reduced_basis, reduce_basis_greedy     # noqa: E402
from pyLatticeSim.lattice_sim import LatticeSim                                                   # noqa: E402
from pyLatticeSim.utils_schur import load_schur_complement_dataset                               # noqa: E402

name_file = "simulation/hybrid_cell_simulation"
tolerance_greedy = 1e-3

lattice_Sim_object = LatticeSim(name_file)
try:
    schur_data = load_schur_complement_dataset(lattice_Sim_object)
except FileNotFoundError as err:
    sys.exit(f"{err}\nNo Schur-complement dataset for this preset yet: run "
             "examples/simulation/construct_schur_complement_dataset.py first.")
file_name = find_name_file_reduced_basis(lattice_Sim_object, tol_greedy=tolerance_greedy)
reduce_basis_greedy(schur_data, tolerance_greedy, file_name)
