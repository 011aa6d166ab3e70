"""``from pyLatticeSim.utils_simulation import solve_FEM_FenicsX`` (reference: src/pyLatticeSim/utils_simulation.py:21)."""
from pylatticedso_amd.utils_simulation import FullScaleLatticeSimulation, solve_FEM_FenicsX  # noqa: F401
