"""Beam-in-flexion simulation through the reference's API names (cf. the reference's
examples/simulation/simulation_lattice.py): LatticeSim -> solve_FEM_FenicsX -> plot -> export."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "src"))

from pyLatticeDesign.plotting_lattice import LatticePlotting                      # noqa: E402
from pyLatticeSim.export_simulation_results import exportSimulationResults       # noqa: E402
from pyLatticeSim.lattice_sim import LatticeSim                                   # noqa: E402
from pyLatticeSim.utils_simulation import solve_FEM_FenicsX                       # noqa: E402

name_file = "simulation/simulation_beam_flexion"
lattice_Sim_object = LatticeSim(name_file)
sol, simulation_lattice = solve_FEM_FenicsX(lattice_Sim_object)
print(f"{lattice_Sim_object.lattice.n_beams} struts, {len(sol)} free boundary dofs, "
      f"{simulation_lattice.stats['iterations']} PCG iterations, max |u| = {abs(sol).max():.4e}")

vizualizer = LatticePlotting()
print("plot:", vizualizer.visualize_lattice(lattice_Sim_object, beam_color_type="radii", deformed_form=True,
                                            enable_boundary_conditions=True))
export_results = exportSimulationResults(simulation_lattice, name_file)
export_results.export_displacement_rotation()
print("vtk :", export_results.export_finalize())
