"""``from pyLatticeSim.greedy_algorithm import ...`` (reference: src/pyLatticeSim/greedy_algorithm.py)."""
from pylatticedso_amd.greedy_algorithm import (find_name_file_reduced_basis, reduce_basis_greedy,  # noqa: F401
                                               save_reduced_basis)
