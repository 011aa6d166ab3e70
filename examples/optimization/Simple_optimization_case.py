"""Constant-radius compliance minimisation under a density constraint through the reference's API names
(cf. the reference's examples/optimization/Simple_optimization_case.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeOpti.lattice_opti import LatticeOpti      # noqa: E402

name_file = "optimization/optimization_beam_flexion"
lattice_object = LatticeOpti(name_file, verbose=1, convergence_plotting=False)
lattice_object.optimize_lattice()
print("solution:", lattice_object.solution.x, "compliance:", lattice_object.denorm_objective,
      "relative density:", lattice_object.relative_density())
