"""Dataset of exact Schur complements of a parametrised cell, condensed on the GPU (cf. the reference's
examples/simulation/construct_schur_complement_dataset.py, which needs dolfinx for every matrix): the input of the
reduced-basis / surrogate DDM modes.  Usage: python construct_schur_complement_dataset.py [preset] [step]"""
import os
import sys
from itertools import product

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeSim.lattice_sim import LatticeSim                                                   # noqa: E402
from pyLatticeSim.utils_schur import get_schur_complement, save_schur_complement_npz              # noqa: E402

name_file = sys.argv[1] if len(sys.argv) > 1 else "simulation/hybrid_cell_simulation"
step_radius = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
lattice_object = LatticeSim(name_file)
radius_range = np.round(np.arange(0.01, 0.11, step_radius), 3)

radius_values_batch, schur_matrix_batch = [], []
for i, radius_combinations in enumerate(product(radius_range, repeat=len(lattice_object.geom_types)), start=1):
    if sum(radius_combinations) <= 0.003:
        continue
    lattice_object.reset_cell_with_new_radii(list(radius_combinations))
    schur_complement = get_schur_complement(lattice_object)
    radius_values_batch.append(list(radius_combinations))
    schur_matrix_batch.append(schur_complement)
    print(f"Combination {i}: {radius_combinations}  |S| = {np.linalg.norm(schur_complement):.4e}")
save_schur_complement_npz(lattice_object, radius_values_batch, schur_matrix_batch)
