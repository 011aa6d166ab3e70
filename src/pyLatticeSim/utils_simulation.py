"""``from pyLatticeSim.utils_simulation import solve_FEM_FenicsX, get_homogenized_properties`` (reference:
src/pyLatticeSim/utils_simulation.py:21,83)."""
from pylatticedso_amd.utils_simulation import (FullScaleLatticeSimulation, get_homogenized_properties,  # noqa: F401
                                               solve_FEM_FenicsX)
