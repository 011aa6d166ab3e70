"""``from pyLatticeSim.utils_schur import get_schur_complement`` (reference: src/pyLatticeSim/utils_schur.py:22)."""
from pylatticedso_amd.utils_schur import (define_path_schur_complement, get_schur_complement,  # noqa: F401
                                          load_schur_complement_dataset, node_order_to_simulate,
                                          save_schur_complement_npz)
