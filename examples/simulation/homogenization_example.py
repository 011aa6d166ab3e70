"""Homogenised elastic constants of one periodic cell (cf. the reference's examples/simulation/homogenization_example.py,
which needs dolfinx + dolfinx_mpc): strut records, global K and all residual evaluations on the GPU, the six
constrained solves on the periodic master nodes on the host."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeSim.export_simulation_results import exportSimulationResults                       # noqa: E402
from pyLatticeSim.homogenization_cell import directional_modulus                                 # noqa: E402
from pyLatticeSim.lattice_sim import LatticeSim                                                   # noqa: E402
from pyLatticeSim.utils import create_homogenization_figure                                      # noqa: E402
from pyLatticeSim.utils_simulation import get_homogenized_properties                             # noqa: E402

name_file = sys.argv[1] if len(sys.argv) > 1 else "simulation/hybrid_cell_simulation"

lattice_object = LatticeSim(name_file)
mat_S_orthotropic, homogenization_analysis = get_homogenized_properties(lattice_object)
homogenization_analysis.print_orthotropic_form()
for name, (theta, phi) in {"[100]": (90, 0), "[110]": (90, 45), "[111]": (np.degrees(np.arccos(3 ** -0.5)), 45)}.items():
    print(f"directional modulus {name}: {np.linalg.norm(directional_modulus(mat_S_orthotropic, theta, phi)):.4f}")

print("figure:", create_homogenization_figure(mat_S_orthotropic, save=True, name_file=name_file, plot=False))

exportData = exportSimulationResults(homogenization_analysis, name_file)
print("written:", *exportData.export_data_homogenization())
