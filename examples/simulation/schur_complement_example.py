"""Schur complement of a hybrid cell (cf. the reference's examples/simulation/schur_complement_example.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeSim.lattice_sim import LatticeSim                   # noqa: E402
from pyLatticeSim.utils_schur import get_schur_complement         # noqa: E402

name_file = "simulation/hybrid_cell_simulation"
lattice_object = LatticeSim(name_file)
schur_complement = get_schur_complement(lattice_object)
print("Schur complement matrix:\n", schur_complement)
