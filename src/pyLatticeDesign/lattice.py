"""``from pyLatticeDesign.lattice import Lattice`` (reference: src/pyLatticeDesign/lattice.py:36)."""
from pylatticedso_amd.lattice import Lattice  # noqa: F401
