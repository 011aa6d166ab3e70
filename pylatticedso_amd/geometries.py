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
    # BCC plus a strut along z through the cell centre, split there (bottom face centre - centre - top face centre)
    return _bcc() + [(0.5, 0.5, 0.0, *_C), (*_C, 0.5, 0.5, 1.0)]


def _fcc():
    rows = []
    for f in _FACE_CENTRES:
        rows += [(*c, *f) for c in _CORNERS if abs(_dist2(c, f) - 0.5) < 1e-12]
    return rows


def _octant_of(c):
    """The point a quarter of the body diagonal inside the cube from corner c."""
    return tuple(0.25 if v == 0.0 else 0.75 for v in c)


def _diamond():
    # diamond cubic: every corner to its octant point, the octant point to the centres of the three faces that meet
    # at that corner
    rows = []
    for c in _CORNERS:
        o = _octant_of(c)
        rows += [(*c, *o), (*o, 0.5, 0.5, c[2]), (*o, c[0], 0.5, 0.5), (*o, 0.5, c[1], 0.5)]
    return rows


def _original():
    # like the diamond cell, but the octant points reach the mid-points of the three EDGES that meet at the corner
    rows = []
    for c in _CORNERS:
        o = _octant_of(c)
        rows += [(*c, *o), (*o, 0.5, c[1], c[2]), (*o, c[0], c[1], 0.5), (*o, c[0], 0.5, c[2])]
    return rows


def _hybrid2():
    return [(*e, *_C) for e in _EDGE_MIDS]                    # the twelve edge mid-points to the cell centre


def _hybrid3():
    # every octant point to the centres of its three nearest faces (Hybrid4 without the face-centre -> centre struts)
    return [(*o, *f) for o in _OCTANTS for f in _FACE_CENTRES if abs(_dist2(o, f) - 3 * 0.0625) < 1e-12]


def _hybrid5():
    # six points half-way between the cell centre and the face centres: each to the centre and to the four corners of
    # its face
    rows = []
    for f in _FACE_CENTRES:
        p = tuple(0.5 * (a + b) for a, b in zip(f, _C))
        rows.append((*p, *_C))
        rows += [(*p, *c) for c in _CORNERS if abs(_dist2(c, f) - 0.5) < 1e-12]
    return rows


def _kelvin():
    # truncated octahedron: on every face a square of four points a quarter edge from the face centre, plus the struts
    # between squares of adjacent faces (nearest points, 1/8 apart in squared distance)
    pts, rows = [], []
    for ax in range(3):
        a, b = [k for k in range(3) if k != ax]
        for side in (0.0, 1.0):
            sq = []
            for k, off in ((a, 0.25), (a, 0.75), (b, 0.25), (b, 0.75)):
                q = [0.5, 0.5, 0.5]
                q[ax] = side
                q[k] = off
                sq.append(tuple(q))
            rows += [(*sq[i], *sq[j]) for i in (0, 1) for j in (2, 3)]
            pts += [(q, (ax, side)) for q in sq]
    for (p, fp), (q, fq) in itertools.combinations(pts, 2):
        if fp != fq and abs(_dist2(p, q) - 0.125) < 1e-12:
            rows.append((*p, *q))
    return rows


def _octahedron():
    return [(*p, *q) for p, q in itertools.combinations(_FACE_CENTRES, 2) if abs(_dist2(p, q) - 0.5) < 1e-12]


def _octahedron_z():
    return _octahedron() + [(0.5, 0.5, 0.0, 0.5, 0.5, 1.0)]          # one strut through the cell along z


def _octahedron_yz():
    return _octahedron() + [(*f, *_C) for f in _FACE_CENTRES]           # (all six face centres to the cell centre)


def _original2():
    # BCC diagonals plus, on every face, the four corners and the four edge mid-points to the face centre
    rows = _bcc()
    for f in _FACE_CENTRES:
        ax = [k for k in range(3) if f[k] != 0.5][0]
        ring = [p for p in itertools.product((0.0, 0.5, 1.0), repeat=3) if p[ax] == f[ax] and p != f]
        rows += [(*p, *f) for p in ring]
    return rows


def _auxetic(hgeom=0.35, angle_deg=20.0):
    # re-entrant ("bow-tie") honeycomb on the four vertical faces of the cell; h = height of the re-entrant node above the
    # face's bottom edge, v = height where the inclined struts leave the vertical edges (parametric in the reference:
    # hgeom = 0.35, angleGeom = 20 degrees, valGeom = hgeom - tan(angleGeom) / 2)
    h = hgeom
    v = _eval_expr("hgeom - tan(angleGeom * pi / 180) / 2", {"hgeom": hgeom, "angleGeom": angle_deg})
    rows = []

    def face(point, upper_from_node):            # point(s, z) -> xyz of the face's in-plane coordinates
        lower = [(point(0.5, 0.0), point(0.5, h)), (point(0.0, v), point(0.5, h)), (point(1.0, v), point(0.5, h))]
        upper = [(point(0.5, 1.0), point(0.5, 1.0 - h)), (point(0.0, 1.0 - v), point(0.5, 1.0 - h)),
                 (point(1.0, 1.0 - v), point(0.5, 1.0 - h))]
        if upper_from_node:     # (strut direction decides from which end the penalisation points are laid off: with these
            upper = [(b, a) for a, b in upper]      # irrational coordinates that is visible in the last bit)
        return [(*a, *b) for a, b in lower + upper]
    for y in (0.0, 1.0):
        rows += face(lambda s, z, y=y: (s, y, z), False)
        rows += [(x, y, v, x, y, 1.0 - v) for x in (0.0, 1.0)]          # the four vertical edges, once
    for x in (1.0, 0.0):
        rows += face(lambda s, z, x=x: (x, s, z), True)
    return rows


_BUILTIN = {"BCC": _bcc, "Cubic": _cubic, "Octet": _octet, "Hybrid1": _hybrid1, "Hybrid4": _hybrid4,
            "BCCZ": _bccz, "FCC": _fcc, "OctetExt": _fcc, "Diamond": _diamond, "Original": _original,
            "Original2": _original2, "Hybrid2": _hybrid2, "Hybrid3": _hybrid3, "Hybrid5": _hybrid5, "Kelvin": _kelvin,
            "Octahedron": _octahedron, "OctahedronZ": _octahedron_z, "OctahedronYZ": _octahedron_yz,
            "Auxetic": _auxetic}
_REGISTERED: dict[str, list] = {}

_OPS = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.Pow: operator.pow, ast.USub: operator.neg, ast.UAdd: operator.pos}
_FUNCS = {"tan": math.tan, "sin": math.sin, "cos": math.cos, "sqrt": math.sqrt, "pi": math.pi}


def _eval_expr(expr, names):
    """Evaluator for parametric geometry files.  The reference evaluates these expressions with
    sympy.sympify(expr, locals).evalf() (geometries_utils.py:20-38), parameters entering as Python floats: pi and tan()
    stay symbolic until the end while float coefficients fold at double precision - the result is neither plain double
    arithmetic nor the correctly rounded value (the Auxetic cell's valGeom differs from both by one ulp), and node
    coordinates must match bit for bit.  So sympy - which the reference requires anyway - is used when it is importable;
    without it an arithmetic-only evaluator in double precision serves (parametric cells are then good to 1 ulp)."""
    if isinstance(expr, (int, float)):
        return float(expr)
    try:
        import sympy
    except ImportError:                                    # pragma: no cover - sympy is a dependency of the reference
        sympy = None
    if sympy is not None:
        ctx = {k: getattr(sympy, k) for k in ("sin", "cos", "tan", "asin", "acos", "atan", "exp", "log", "sqrt", "pi")}
        ctx.update({k: float(v) for k, v in names.items()})
        try:
            res = sympy.sympify(str(expr), locals=ctx)
            return float(res.evalf()) if hasattr(res, "evalf") else float(res)
        except Exception as e:      # noqa: BLE001 - same message as the reference
            raise ValueError(f"Failed to evaluate expression '{expr}': {e}") from e

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
