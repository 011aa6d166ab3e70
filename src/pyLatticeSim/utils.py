"""``from pyLatticeSim.utils import create_homogenization_figure`` (reference: src/pyLatticeSim/utils.py:19-148)."""
from pylatticedso_amd.sim_utils import (clear_directory, create_homogenization_figure,  # noqa: F401
                                        directional_modulus, directional_modulus_grid)
