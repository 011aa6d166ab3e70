"""``from pyLatticeSim.homogenization_cell import HomogenizedCell`` (reference: src/pyLatticeSim/homogenization_cell.py:60)."""
from pylatticedso_amd.homogenization_cell import HomogenizedCell, directional_modulus  # noqa: F401
