"""``pyLatticeDesign.gradient_properties`` under its reference names (src/pyLatticeDesign/gradient_properties.py):
per-axis gradient factors of the cell size / strut radius, as plain lists.  The arithmetic is
``lattice_arrays.gradient_table``, which the array-backed lattice generation uses directly."""
import random

from .lattice_arrays import gradient_table


def grad_settings_constant(num_cells_x, num_cells_y, num_cells_z, material_gradient: bool = False):
    """All factors 1 (gradient_properties.py:12-40): one [1, 1, 1] row per cell, or an nz x ny x nx block of 1 for
    the material table."""
    if material_gradient:
        return [[[1] * num_cells_x for _ in range(num_cells_y)] for _ in range(num_cells_z)]
    return [[1.0, 1.0, 1.0] for _ in range(num_cells_x * num_cells_y * num_cells_z)]


def get_grad_settings(num_cells_x, num_cells_y, num_cells_z, grad_properties):
    """[rule, direction, parameters] -> rows [fx, fy, fz] for index 0 .. max(n) - 1 (gradient_properties.py:44-137)."""
    rule, direction, parameters = grad_properties
    rule = getattr(rule, "value", rule)
    return gradient_table(num_cells_x, num_cells_y, num_cells_z, rule, direction, parameters).tolist()


def grad_material_setting(numCellsX, numCellsY, numCellsZ, gradMatProperty):
    """Material index table nz x ny x nx (gradient_properties.py:140-183): -1 random in 1..3, 0 single material,
    1 graded along the given axis; anything else -> empty list."""
    multimat, direction = gradMatProperty
    if multimat not in (-1, 0, 1):
        return []

    def value(x, y, z):
        if multimat == -1:
            return random.randint(1, 3)
        if multimat == 0:
            return 1
        return (x, y, z)[direction if direction in (0, 1) else 2] + 1

    return [[[value(x, y, z) for x in range(numCellsX)] for y in range(numCellsY)] for z in range(numCellsZ)]
