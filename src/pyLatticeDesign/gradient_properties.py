"""``from pyLatticeDesign.gradient_properties import ...`` (reference: src/pyLatticeDesign/gradient_properties.py)."""
from pylatticedso_amd.gradient_properties import (get_grad_settings, grad_material_setting,  # noqa: F401
                                                  grad_settings_constant)
