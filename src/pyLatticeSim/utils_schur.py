"""``from pyLatticeSim.utils_schur import get_schur_complement`` (reference: src/pyLatticeSim/utils_schur.py:22)."""
from pylatticedso_amd.utils_schur import get_schur_complement, node_order_to_simulate  # noqa: F401
