"""Gradient settings under the reference's names (what the reference's Tests/Gradient_test.py exercises), plus the
five rules against their closed forms and against the factors the lattice generator applies to the cells."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "src"))
from pyLatticeDesign.gradient_properties import (get_grad_settings, grad_material_setting,  # noqa: E402
                                                 grad_settings_constant)
from pylatticedso_amd import lattice_arrays as LA  # noqa: E402


@pytest.mark.parametrize("n", [(2, 2, 2), (1, 1, 1), (3, 2, 1)])
def test_constant_settings(n):
    rows = grad_settings_constant(*n)
    assert isinstance(rows, list) and len(rows) == n[0] * n[1] * n[2]
    assert all(isinstance(r, list) and r == [1.0, 1.0, 1.0] and all(isinstance(v, float) for v in r) for r in rows)
    block = grad_settings_constant(*n, material_gradient=True)
    assert (len(block), len(block[0]), len(block[0][0])) == (n[2], n[1], n[0])
    assert all(v == 1 for plane in block for row in plane for v in row)


def test_rules_follow_their_closed_forms():
    n, p = 6, 0.3
    closed = {"constant": lambda i: 1.0, "linear": lambda i: 1.0 + i * p,
              "parabolic": lambda i: 1.0 + (i / 3.0) * p if i < 3 else 1.0 + ((n - i - 1) / 3.0) * p,
              "sinusoide": lambda i: 1.0 + p * math.sin(i / n * math.pi),
              "exponential": lambda i: 1.0 + math.exp(i * p)}
    for rule, f in closed.items():
        rows = get_grad_settings(n, 4, 2, [rule, [1, 0, 0], [p, 0.0, 0.0]])
        assert isinstance(rows, list) and len(rows) == n          # one row per index up to the longest axis
        assert [r[0] for r in rows] == pytest.approx([f(i) for i in range(n)])
        assert all(r[1] == 1.0 and r[2] == 1.0 for r in rows)
    with pytest.raises(ValueError):
        get_grad_settings(2, 2, 2, ["cubic", [1, 1, 1], [0.1, 0.1, 0.1]])


def test_short_axes_hold_their_last_factor():
    # the row index runs to max(n); an axis with fewer cells stays at its last value (gradient_properties.py:126-137)
    rows = get_grad_settings(4, 2, 1, ["linear", [1, 1, 1], [0.1, 0.5, 2.0]])
    assert [r[0] for r in rows] == pytest.approx([1.0, 1.1, 1.2, 1.3])
    assert [r[1] for r in rows] == pytest.approx([1.0, 1.5, 1.5, 1.5])
    assert [r[2] for r in rows] == pytest.approx([1.0, 1.0, 1.0, 1.0])


def test_generator_applies_the_table_to_the_cells():
    table = np.asarray(get_grad_settings(4, 2, 2, ["linear", [1, 0, 1], [0.25, 0.0, 0.5]]))
    lat = LA.generate((1, 1, 1), (4, 2, 2), ["BCC"], [0.04], grad_radius=table)
    pos = lat.cell_pos
    expect = 0.04 * table[pos[:, 0], 0] * table[pos[:, 1], 1] * table[pos[:, 2], 2]
    assert np.allclose(lat.cell_radii[:, 0], expect)


def test_material_table():
    assert grad_material_setting(3, 2, 2, [0, 0]) == [[[1, 1, 1], [1, 1, 1]], [[1, 1, 1], [1, 1, 1]]]
    g = grad_material_setting(3, 2, 2, [1, 0])
    assert g[0][0] == [1, 2, 3] and g[1][1] == [1, 2, 3]
    g = grad_material_setting(3, 2, 2, [1, 2])
    assert g[0][0] == [1, 1, 1] and g[1][0] == [2, 2, 2]
    r = grad_material_setting(2, 2, 2, [-1, 0])
    assert all(v in (1, 2, 3) for plane in r for row in plane for v in row)
    assert grad_material_setting(2, 2, 2, [7, 0]) == []
