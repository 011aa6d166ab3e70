"""``from pyLatticeDesign.utils import ...`` (reference: src/pyLatticeDesign/utils.py:111-453)."""
from pylatticedso_amd.design_utils import save_JSON_to_Grasshopper, save_lattice_object  # noqa: F401
from pylatticedso_amd.lattice_sim import open_lattice_parameters  # noqa: F401


def function_penalization_Lzone(radius: float, angle: float) -> float:
    """utils.py:432-453: penalisation length r / tan(angle / 2) (1e-7 beyond 170 degrees, 0 at 0)."""
    import math
    if angle > 170:
        return 0.0000001
    if angle == 0.0:
        return 0.0
    return radius / math.tan(math.radians(angle) / 2)
