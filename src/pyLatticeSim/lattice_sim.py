"""``from pyLatticeSim.lattice_sim import LatticeSim`` (reference: src/pyLatticeSim/lattice_sim.py:83)."""
from pylatticedso_amd.lattice_sim import LatticeSim, open_lattice_parameters  # noqa: F401
