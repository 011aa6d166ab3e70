"""Unit-cell strut tables (fractional end points inside the unit cube).

Same schema as the reference's ``src/pyLatticeDesign/geometries/*.json`` (``get_beam_structure``,
geometries_utils.py:41-89): a list of ``[x1, y1, z1, x2, y2, z2]`` rows.  The built-in cells are
generated from their crystallographic description rather than tabulated; user cells can be dropped as
JSON files (same schema, optional ``parameters`` block with arithmetic expressions) into a directory
passed through ``PYLATTICE_GEOMETRY_PATH`` or ``register_geometry``.
"""
from __future__ import annotations

import ast
import itertools
import json
import math
import operator
import os

import numpy as np

_C = (0.5, 0.5, 0.5)
_CORNERS = list(itertools.product((0.0, 1.0), repeat=3))
_FACE_CENTRES = [(0.5, 0.5, 0.0), (0.5, 0.5, 1.0), (0.5, 0.0, 0.5), (0.5, 1.0, 0.5), (0.0, 0.5, 0.5), (1.0, 0.5, 0.5)]
_EDGE_MIDS = [p for p in itertools.product((0.0, 0.5, 1.0), repeat=3) if sorted(p).count(0.5) == 1
              and all(v in (0.0, 0.5, 1.0) for v in p)]
_OCTANTS = list(itertools.product((0.25, 0.75), repeat=3))


def _dist2(p, q):
    return sum((a - b) ** 2 for a, b in zip(p, q))


def _bcc():
    return [(*_C, *c) for c in _CORNERS]


def _cubic():
    return [(*p, *q) for p, q in itertools.combinations(_CORNERS, 2) if abs(_dist2(p, q) - 1.0) < 1e-12]


def _octet():
    rows = []
    for f in _FACE_CENTRES:                      # face diagonals: face centre -> its 4 corners
        rows += [(*c, *f) for c in _CORNERS if abs(_dist2(c, f) - 0.5) < 1e-12]
    rows += [(*p, *q) for p, q in itertools.combinations(_FACE_CENTRES, 2) if abs(_dist2(p, q) - 0.5) < 1e-12]
    return rows


def _hybrid1():
    # every octant point (1/4,3/4)^3 to the three nearest edge mid-points
    return [(*o, *e) for o in _OCTANTS for e in _EDGE_MIDS if abs(_dist2(o, e) - 3 * 0.0625) < 1e-12]


def _hybrid4():
    rows = [(*o, *f) for o in _OCTANTS for f in _FACE_CENTRES if abs(_dist2(o, f) - 3 * 0.0625) < 1e-12]
    rows += [(*f, *_C) for f in _FACE_CENTRES]
    return rows


def _bccz():
    return _bcc() + [(x, y, 0.0, x, y, 1.0) for x in (0.0, 1.0) for y in (0.0, 1.0)]


def _fcc():
    rows = []
    for f in _FACE_CENTRES:
        rows += [(*c, *f) for c in _CORNERS if abs(_dist2(c, f) - 0.5) < 1e-12]
    return rows


_BUILTIN = {"BCC": _bcc, "Cubic": _cubic, "Octet": _octet, "Hybrid1": _hybrid1, "Hybrid4": _hybrid4,
            "BCCZ": _bccz, "FCC": _fcc}
_REGISTERED: dict[str, list] = {}

_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.Pow: operator.pow, ast.USub: operator.neg, ast.UAdd: operator.pos}
_FUNCS = {"tan": math.tan, "sin": math.sin, "cos": math.cos, "sqrt": math.sqrt, "pi": math.pi}


def _eval_expr(expr, names):
    """Arithmetic-only evaluator for parametric geometry files (the reference uses sympy.sympify)."""
    if isinstance(expr, (int, float)):
        return float(expr)

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return float(n.value)
        if isinstance(n, ast.BinOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.left), ev(n.right))
        if isinstance(n, ast.UnaryOp) and type(n.op) in _OPS:
            return _OPS[type(n.op)](ev(n.operand))
        if isinstance(n, ast.Name) and (n.id in names or n.id in _FUNCS):
            return names.get(n.id, _FUNCS.get(n.id))
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in _FUNCS:
            return _FUNCS[n.func.id](*[ev(a) for a in n.args])
        raise ValueError(f"Failed to evaluate expression '{expr}'")

    return float(ev(ast.parse(str(expr), mode="eval")))


def register_geometry(name: str, beams) -> None:
    _REGISTERED[name] = [tuple(map(float, b)) for b in beams]


def _from_json(path):
    with open(path, "r") as fh:
        geometry = json.load(fh)
    params = {}
    for key, val in geometry.get("parameters", {}).items():
        params[key] = _eval_expr(val, params)
    return [tuple(_eval_expr(c, params) for c in beam) for beam in geometry["beams"]]


def get_beam_structure(lattice_type: str) -> np.ndarray:
    """(nb, 6) float64 array of fractional strut end points of one unit cell."""
    if lattice_type in _REGISTERED:
        rows = _REGISTERED[lattice_type]
    elif lattice_type in _BUILTIN:
        rows = _BUILTIN[lattice_type]()
    else:
        rows = None
        for d in filter(None, os.environ.get("PYLATTICE_GEOMETRY_PATH", "").split(os.pathsep)):
            p = os.path.join(d, f"{lattice_type}.json")
            if os.path.exists(p):
                rows = _from_json(p)
                break
        if rows is None:
            raise FileNotFoundError(f"Geometry file '{lattice_type}.json' not found.")
    return np.asarray(rows, dtype=np.float64).reshape(-1, 6)
