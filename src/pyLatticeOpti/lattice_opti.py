"""``from pyLatticeOpti.lattice_opti import LatticeOpti`` (reference: src/pyLatticeOpti/lattice_opti.py:59)."""
from pylatticedso_amd.lattice_opti import LatticeOpti  # noqa: F401
