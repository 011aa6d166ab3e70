"""Lazy ``Cell`` / ``Beam`` / ``Point`` views over the array-backed lattice.

The reference keeps the lattice as a Python object graph (``lattice.cells[i].beams_cell``, ``beam.point1.x``,
``point.displacement_vector`` ... - src/pyLatticeDesign/{cell,beam,point}.py); at 10^6 cells that is minutes of
construction and tens of GB, so here the arrays of ``lattice_arrays`` / ``LatticeSim`` are the truth and these small
objects are created on demand, each holding only (simulation, index).  Attribute names, units and index conventions
are the reference's:

* node indices: design nodes sorted by coordinates (lattice.py:687-696), then - once the joints are penalised - the
  penalisation points (``node_mod``), again sorted by coordinates;
* beam indices: design struts sorted by (lower end, upper end, radius) (lattice.py:675-685); after
  ``set_penalized_beams`` (lattice_sim.py:245-308) every strut with a penalised end is REPLACED by its up-to-three
  segments, which take the indices after the design struts in the same sort order; penalised segments carry
  ``beam_mod = True`` and the radius ``1.5 r`` (beam.py:405-411).

Vector attributes (``displacement_vector``, ``fixed_DOF`` ...) are live rows of the simulation's (N, 6) arrays: reading
costs nothing, writing through them updates the simulation state like assigning to the reference's lists does.
"""
from __future__ import annotations

import math

import numpy as np

from . import lattice_arrays as LA


# ------------------------------------------------------------------------------------------------------------------
# tables
# ------------------------------------------------------------------------------------------------------------------
class _Tables:
    """Index tables of one LatticeSim state (rebuilt whenever the lattice arrays are regenerated)."""

    def __init__(self, sim):
        lat, pen = sim.lattice, sim.penalized
        N, B = lat.n_nodes, lat.n_beams
        self.n_design_nodes, self.n_design_beams = N, B
        penalised = bool(sim.is_penalized) and pen is not None
        has = np.zeros((B, 2), bool)
        if penalised:
            has[:, 0], has[:, 1] = pen.seg_len[:, 0] > 0, pen.seg_len[:, 2] > 0
        # ---- penalisation points: indices N.. in coordinate order
        sb, se = np.nonzero(has)
        pxyz = pen.pen_xyz[sb, se] if len(sb) else np.zeros((0, 3))
        order = np.lexsort((pxyz[:, 2], pxyz[:, 1], pxyz[:, 0])) if len(sb) else np.zeros(0, np.int64)
        self.pen_strut, self.pen_end = sb[order], se[order]
        self.pen_xyz = pxyz[order]
        pid = np.full((B, 2), -1, np.int64)
        pid[self.pen_strut, self.pen_end] = N + np.arange(len(order))
        self.pen_id = pid
        self.node_xyz = np.concatenate([lat.node_xyz, self.pen_xyz]) if len(order) else lat.node_xyz
        self.n_nodes = len(self.node_xyz)
        self._sim_ref = (sim, has, pid, lat.beam_conn[:, 0].astype(np.int64), lat.beam_conn[:, 1].astype(np.int64))
        # node -> cells (design nodes); penalisation points belong to the cells of their strut
        cn_ptr, cn_idx = lat.cell_node_ptr, lat.cell_node_idx
        cell_of = np.repeat(np.arange(lat.n_cells), np.diff(cn_ptr))
        o3 = np.argsort(cn_idx, kind="stable")
        ptr3 = np.concatenate([[0], np.bincount(cn_idx, minlength=N)]).astype(np.int64)
        self.node_cell_ptr, self.node_cell_idx = np.cumsum(ptr3), cell_of[o3]
        cb_ptr, cb_idx = lat.cell_beam_ptr, lat.cell_beam_idx
        cell_of_b = np.repeat(np.arange(lat.n_cells), np.diff(cb_ptr))
        o4 = np.argsort(cb_idx, kind="stable")
        ptr4 = np.concatenate([[0], np.bincount(cb_idx, minlength=B)]).astype(np.int64)
        self.strut_cell_ptr, self.strut_cell_idx = np.cumsum(ptr4), cell_of_b[o4]

    # The beam (segment) tables are only built when somebody looks at beams: at 50^3 Octet they are 9 M segments sorted on
    # seven keys (3 s), and the reference_compat bookkeeping needs the NODE tables only.
    _BEAM_ATTRS = ("beam_conn", "beam_parent", "beam_part", "beam_mod", "beam_radius", "beam_index", "beam_copy", "n_beams",
                   "has_copies", "strut_beam_ptr", "strut_beam_idx", "node_beam_ptr", "node_beam_idx")

    def __getattr__(self, name):
        if name in _Tables._BEAM_ATTRS and "_sim_ref" in self.__dict__:
            self._build_beams()
            return self.__dict__[name]
        raise AttributeError(name)

    def _build_beams(self):
        sim, has, pid, a, b = self.__dict__.pop("_sim_ref")
        lat, pen = sim.lattice, sim.penalized
        N, B = lat.n_nodes, lat.n_beams
        # ---- beams: untouched design struts keep their index, the segments of the others follow
        split = has.any(axis=1)
        keep = np.flatnonzero(~split)
        s1 = np.flatnonzero(has[:, 0])
        s3 = np.flatnonzero(has[:, 1])
        sm = np.flatnonzero(split)
        start = np.where(has[:, 0], pid[:, 0], a)
        stop = np.where(has[:, 1], pid[:, 1], b)
        seg_conn = np.concatenate([np.c_[a[s1], pid[s1, 0]], np.c_[start[sm], stop[sm]], np.c_[pid[s3, 1], b[s3]]])
        seg_parent = np.concatenate([s1, sm, s3])
        seg_part = np.concatenate([np.zeros(len(s1), np.int8), np.ones(len(sm), np.int8), np.full(len(s3), 2, np.int8)])
        # reference_compat: a split strut shared by k cells is k copies of every segment, copy j living in the beams_cell
        # of its j-th owner cell only (lattice_sim.py:250-303)
        mult = getattr(sim, "beam_mult", None) if getattr(sim, "_compat_rows", False) else None
        seg_copy = np.zeros(len(seg_parent), np.int64)
        if mult is not None and len(seg_parent) and (mult[seg_parent] > 1).any():
            rep = mult[seg_parent]
            seg_copy = np.arange(rep.sum()) - np.repeat(np.cumsum(rep) - rep, rep)
            seg_conn, seg_parent, seg_part = (np.repeat(seg_conn, rep, axis=0), np.repeat(seg_parent, rep),
                                              np.repeat(seg_part, rep))
        seg_mod = seg_part != 1
        pc = LA.PENALIZATION_COEFFICIENT
        seg_rad = lat.beam_radius[seg_parent] * np.where(seg_mod, pc, 1.0)
        if len(seg_conn):
            p1, p2 = self.node_xyz[seg_conn[:, 0]], self.node_xyz[seg_conn[:, 1]]
            d = p1 - p2
            first = np.argmax(d != 0, axis=1)
            less = d[np.arange(len(d)), first] < 0                      # p1 < p2 lexicographically
            lo, hi = np.where(less[:, None], p1, p2), np.where(less[:, None], p2, p1)
            so = np.lexsort((seg_rad, hi[:, 2], hi[:, 1], hi[:, 0], lo[:, 2], lo[:, 1], lo[:, 0]))
        else:
            so = np.zeros(0, np.int64)
        self.beam_conn = np.concatenate([np.c_[a[keep], b[keep]], seg_conn[so]]) if B else np.zeros((0, 2), np.int64)
        self.beam_parent = np.concatenate([keep, seg_parent[so]])
        self.beam_part = np.concatenate([np.full(len(keep), -1, np.int8), seg_part[so]])     # -1: whole design strut
        self.beam_mod = np.concatenate([np.zeros(len(keep), bool), seg_mod[so]])
        self.beam_radius = np.concatenate([lat.beam_radius[keep], seg_rad[so]])
        self.beam_index = np.concatenate([keep, B + np.arange(len(so))])                    # reference's Beam.index
        self.beam_copy = np.concatenate([np.full(len(keep), -1, np.int64), seg_copy[so]])   # -1: one object for all cells
        self.n_beams = len(self.beam_conn)
        self.has_copies = bool((self.beam_copy > 0).any())
        # strut -> its beams (views): CSR
        o = np.argsort(self.beam_parent, kind="stable")
        ptr = np.concatenate([[0], np.bincount(self.beam_parent, minlength=B)]).astype(np.int64)
        self.strut_beam_ptr, self.strut_beam_idx = np.cumsum(ptr), o
        # node -> beams
        ends = self.beam_conn.ravel()
        o2 = np.argsort(ends, kind="stable")
        ptr2 = np.concatenate([[0], np.bincount(ends, minlength=self.n_nodes)]).astype(np.int64)
        self.node_beam_ptr, self.node_beam_idx = np.cumsum(ptr2), o2 // 2


def _tables(sim) -> _Tables:
    t = getattr(sim, "_view_tables", None)
    if t is None or t[0] is not sim.lattice or t[1] is not sim.penalized:
        t = (sim.lattice, sim.penalized, _Tables(sim))
        sim._view_tables = t
    return t[2]


class _LazySeq:
    """len / index / iterate without materialising the objects (the reference hands out lists and sets)."""

    def __init__(self, n, make):
        self._n, self._make = int(n), make

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._make(k) for k in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return self._make(i)

    def __iter__(self):
        return (self._make(k) for k in range(self._n))

    def __contains__(self, item):
        return getattr(item, "_sim", None) is not None and 0 <= getattr(item, "_i", -1) < self._n


# ------------------------------------------------------------------------------------------------------------------
# Point
# ------------------------------------------------------------------------------------------------------------------
class PointView:
    __slots__ = ("_sim", "_i", "magnification_factor")

    def __init__(self, sim, i):
        self._sim, self._i, self.magnification_factor = sim, int(i), 5.0

    # -- geometry
    def _xyz(self):
        return _tables(self._sim).node_xyz[self._i]

    x = property(lambda self: float(self._xyz()[0]))
    y = property(lambda self: float(self._xyz()[1]))
    z = property(lambda self: float(self._xyz()[2]))
    index = property(lambda self: self._i)
    coordinates = property(lambda self: (self.x, self.y, self.z))
    data = property(lambda self: [self._i, self.x, self.y, self.z])

    @property
    def node_mod(self):
        return self._i >= _tables(self._sim).n_design_nodes

    # -- simulation rows (point.py:68-72)
    def _row(self, name):
        t = _tables(self._sim)
        if self._i < t.n_design_nodes or getattr(self._sim, "_compat_rows", False):
            return getattr(self._sim, name)[self._i]
        return self._sim._node_mod_rows(name)[self._i - t.n_design_nodes]

    displacement_vector = property(lambda self: self._row("displacement_vector"))
    reaction_force_vector = property(lambda self: self._row("reaction_force_vector"))
    applied_force = property(lambda self: self._row("applied_force"))
    fixed_DOF = property(lambda self: self._row("fixed_DOF"))

    @property
    def index_boundary(self):
        if self.node_mod and not getattr(self._sim, "_compat_rows", False):
            return None
        v = int(self._sim.index_boundary[self._i])
        return None if v < 0 else v

    @property
    def tag(self):
        """Lattice.apply_tag_all_point (lattice.py:981-1000): tag relative to the lattice box - or, once blocks are erased,
        relative to the 'continuity box' of the LAST cell that holds the point (get_relative_boundary_box, :1046-1086: per
        axis the extent of all cells with the same index along that axis)."""
        sim, box = self._sim, self._sim.lattice.bbox
        if getattr(sim, "eraser_blocks", None) is not None and not self.node_mod:
            t, lat = _tables(sim), sim.lattice
            c = t.node_cell_idx[t.node_cell_ptr[self._i + 1] - 1]
            box = []
            for ax in range(3):
                same = lat.cell_pos[:, ax] == lat.cell_pos[c, ax]
                box += [lat.cell_coord[same, ax].min(), (lat.cell_coord[same, ax] + lat.cell_size[same, ax]).max()]
            box = np.asarray(box)
        v = int(LA.point_tags(self._xyz()[None, :], box)[0])
        return None if v < 0 else v

    @property
    def deformed_coordinates(self):
        u, m = self.displacement_vector, self.magnification_factor
        return self.x + u[0] * m, self.y + u[1] * m, self.z + u[2] * m

    @property
    def cell_belongings(self):
        t = _tables(self._sim)
        if self.node_mod:
            k = self._i - t.n_design_nodes
            s = t.pen_strut[k]
            cells = t.strut_cell_idx[t.strut_cell_ptr[s]:t.strut_cell_ptr[s + 1]]
        else:
            cells = t.node_cell_idx[t.node_cell_ptr[self._i]:t.node_cell_ptr[self._i + 1]]
        return [CellView(self._sim, c) for c in cells]

    @property
    def connected_beams(self):
        t = _tables(self._sim)
        return [BeamView(self._sim, b) for b in t.node_beam_idx[t.node_beam_ptr[self._i]:t.node_beam_ptr[self._i + 1]]]

    # -- behaviour of point.py
    def __eq__(self, other):
        return isinstance(other, PointView) and self.coordinates == other.coordinates

    def __hash__(self):
        return hash(self.coordinates)

    def __sub__(self, other):
        return [self.x - other.x, self.y - other.y, self.z - other.z]

    def __repr__(self):
        return f"point({self.x}, {self.y}, {self.z}, Index:{self._i})"

    def distance_to(self, other):
        return math.sqrt(sum(d * d for d in (self - other)))

    def is_on_boundary(self, boundary_box_lattice):
        b = boundary_box_lattice
        return (self.x in (b[0], b[1])) or (self.y in (b[2], b[3])) or (self.z in (b[4], b[5]))

    def tag_point(self, boundary_box_domain):
        v = int(LA.point_tags(self._xyz()[None, :], np.asarray(boundary_box_domain, float))[0])
        return None if v < 0 else v

    def initialize_reaction_force(self):
        self.reaction_force_vector[:] = 0.0

    def initialize_displacement(self):
        self.displacement_vector[:] = 0.0

    def set_applied_force(self, appliedForce, DOF):
        if len(DOF) != len(appliedForce):
            raise ValueError("Length of DOF and applied_force must be equal.")
        for d, v in zip(DOF, appliedForce):
            self.applied_force[d] = v

    def set_reaction_force(self, reactionForce):
        if len(reactionForce) != 6:
            raise ValueError("Reaction force must have exactly 6 values.")
        self.reaction_force_vector[:] += np.asarray(reactionForce, float)      # accumulates (point.py:368-380)

    def fix_DOF(self, DOF):
        for d in DOF:
            self.fixed_DOF[d] = True

    def calculate_point_energy(self):
        return float(((self.reaction_force_vector + self.applied_force) * self.displacement_vector).sum())


# ------------------------------------------------------------------------------------------------------------------
# Beam
# ------------------------------------------------------------------------------------------------------------------
class BeamView:
    __slots__ = ("_sim", "_i")

    def __init__(self, sim, i):
        self._sim, self._i = sim, int(i)

    def _t(self):
        return _tables(self._sim)

    point1 = property(lambda self: PointView(self._sim, self._t().beam_conn[self._i, 0]))
    point2 = property(lambda self: PointView(self._sim, self._t().beam_conn[self._i, 1]))
    radius = property(lambda self: float(self._t().beam_radius[self._i]))
    beam_mod = property(lambda self: bool(self._t().beam_mod[self._i]))
    index = property(lambda self: int(self._t().beam_index[self._i]))
    type_beam = property(lambda self: int(self._sim.lattice.beam_type[self._t().beam_parent[self._i]]))
    material = property(lambda self: 0)
    penalization_coefficient = property(lambda self: LA.PENALIZATION_COEFFICIENT)

    @property
    def length(self):                                   # Beam.get_length rounds to 4 decimals (beam.py:125-135)
        return round(self.point1.distance_to(self.point2), 4)

    def get_length(self):
        return self.length

    @property
    def volume(self):
        return self.get_volume()

    def get_volume(self, section_type="circular"):
        if section_type != "circular":
            raise NotImplementedError("Only circular sections are supported.")
        return math.pi * self.radius ** 2 * self.length

    @property
    def data(self):
        c = self._t().beam_conn[self._i]
        return [self.index, int(c[0]), int(c[1]), self.type_beam]

    @property
    def cell_belongings(self):
        t = self._t()
        s = t.beam_parent[self._i]
        return [CellView(self._sim, c) for c in t.strut_cell_idx[t.strut_cell_ptr[s]:t.strut_cell_ptr[s + 1]]]

    def _angle(self, end):
        lz = getattr(self._sim, "lzone", None)
        s = self._t().beam_parent[self._i]
        return {"radius": None, "angle": None, "L_zone": None if lz is None else float(lz[s, end])}

    angle_point_1 = property(lambda self: self._angle(0))
    angle_point_2 = property(lambda self: self._angle(1))

    def get_length_mod(self):
        lz = self._sim.lzone[self._t().beam_parent[self._i]]
        return float(lz[0]), float(lz[1])

    def is_point_on_beam(self, node):
        """beam.py:332-362: exact collinearity (cross product == 0) and the projection inside the segment."""
        p1, p2 = self.point1, self.point2
        v1 = (p2.x - p1.x, p2.y - p1.y, p2.z - p1.z)
        v2 = (node.x - p1.x, node.y - p1.y, node.z - p1.z)
        if node.coordinates in (p1.coordinates, p2.coordinates):
            return False
        cr = (v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0])
        if cr != (0, 0, 0):
            return False
        dotp = v2[0] * v1[0] + v2[1] * v1[1] + v2[2] * v1[2]
        return 0 <= dotp <= v1[0] ** 2 + v1[1] ** 2 + v1[2] ** 2

    def __eq__(self, other):
        return isinstance(other, BeamView) and other._sim is self._sim and other._i == self._i

    def __hash__(self):
        return hash((id(self._sim), self._i))

    def __repr__(self):
        return f"Beam({self.point1}, {self.point2}, radii:{self.radius}, Type:{self.type_beam}, Index:{self.index})"


# ------------------------------------------------------------------------------------------------------------------
# Cell
# ------------------------------------------------------------------------------------------------------------------
class CellView:
    __slots__ = ("_sim", "_i")

    def __init__(self, sim, i):
        self._sim, self._i = sim, int(i)

    def _lat(self):
        return self._sim.lattice

    index = property(lambda self: self._i)
    pos = property(lambda self: [int(v) for v in self._lat().cell_pos[self._i]])
    coordinate = property(lambda self: [float(v) for v in self._lat().cell_coord[self._i]])
    size = property(lambda self: [float(v) for v in self._lat().cell_size[self._i]])
    geom_types = property(lambda self: list(self._sim.geom_types))
    radii = property(lambda self: [float(v) for v in self._sim._cell_parameter_radii()[self._i]])
    center_point = property(lambda self: [c + 0.5 * s for c, s in zip(self.coordinate, self.size)])

    @property
    def beams_cell(self):
        lat, t = self._lat(), _tables(self._sim)
        struts = lat.cell_beam_idx[lat.cell_beam_ptr[self._i]:lat.cell_beam_ptr[self._i + 1]]
        ids = np.concatenate([t.strut_beam_idx[t.strut_beam_ptr[s]:t.strut_beam_ptr[s + 1]] for s in struts]) \
            if len(struts) else np.zeros(0, np.int64)
        if t.has_copies and len(ids):
            # copy j of a split strut belongs to the (j mod owners)-th owner cell of the strut (owner cells in ascending
            # order; more copies than owners: struts cut by check_hybrid_collision, every owner penalises every copy)
            par = t.beam_parent[ids]
            own = t.strut_cell_ptr[par + 1] - t.strut_cell_ptr[par]
            rank = np.array([np.searchsorted(t.strut_cell_idx[t.strut_cell_ptr[s]:t.strut_cell_ptr[s + 1]], self._i)
                             for s in par])
            ids = ids[(t.beam_copy[ids] < 0) | (t.beam_copy[ids] % np.maximum(1, own) == rank)]
        return [BeamView(self._sim, b) for b in np.sort(ids)]

    @property
    def points_cell(self):
        lat, t = self._lat(), _tables(self._sim)
        nodes = lat.cell_node_idx[lat.cell_node_ptr[self._i]:lat.cell_node_ptr[self._i + 1]]
        struts = lat.cell_beam_idx[lat.cell_beam_ptr[self._i]:lat.cell_beam_ptr[self._i + 1]]
        pens = t.pen_id[struts].ravel()
        return [PointView(self._sim, n) for n in np.concatenate([nodes, np.sort(pens[pens >= 0])])]

    @property
    def boundary_box(self):
        c, s = self.coordinate, self.size
        return [c[0], c[0] + s[0], c[1], c[1] + s[1], c[2], c[2] + s[2]]

    @property
    def corner_coordinates(self):
        b = self.boundary_box
        return [(x, y, z) for x in b[0:2] for y in b[2:4] for z in b[4:6]]

    @property
    def volume(self):
        s = self.size
        return s[0] * s[1] * s[2]

    @property
    def volume_each_geom(self):
        out = np.zeros(len(self._sim.geom_types))
        for b in self.beams_cell:
            out[b.type_beam] += b.volume
        return out

    @property
    def relative_density(self):
        return float(self.volume_each_geom.sum() / self.volume)

    @property
    def schur_complement(self):
        S, idx = self._sim.schur_complements, self._sim.cell_schur_index
        return None if S is None or idx is None else S[idx[self._i]]

    @property
    def node_in_order_simulation(self):
        from .utils_schur import node_order_to_simulate
        return [PointView(self._sim, n) for n in node_order_to_simulate(self._sim, self._i)]

    def define_node_order_to_simulate(self, face_priority=None, tol=1e-9):
        return None                                       # the order is a pure function of the arrays here

    def get_number_boundary_nodes(self):
        return len(self.node_in_order_simulation)

    def get_number_nodes_at_boundary(self):
        return self.get_number_boundary_nodes()

    def get_point_on_surface(self, surfaceName):
        ax = "XYZ".index(surfaceName[0].upper())
        val = self.coordinate[ax] + (self.size[ax] if surfaceName.lower().endswith("max") else 0.0)
        return [p for p in self.points_cell if p.coordinates[ax] == val]

    def get_displacement_data(self):
        return [list(p.displacement_vector) for p in self.node_in_order_simulation]

    def get_internal_energy(self):
        return float(sum(p.calculate_point_energy() for p in self.points_cell if p.index_boundary is not None))

    def change_beam_radius(self, new_radius):
        """Cell.change_beam_radius (cell.py:896-917): new radius per geometry for the struts of this cell."""
        if len(new_radius) != len(self._sim.geom_types):
            raise ValueError("Invalid hybrid radii data.")
        radii = self._sim._cell_parameter_radii().copy()
        radii[self._i] = new_radius
        self._sim.set_cell_radii(radii)

    def __eq__(self, other):
        return isinstance(other, CellView) and other._sim is self._sim and other._i == self._i

    def __hash__(self):
        return hash((id(self._sim), "cell", self._i))

    def __repr__(self):
        return f"Cell(Coordinates:{self.coordinate}, Size: {self.size}, Index:{self._i})"


class LatticeViews:
    """Mixin of LatticeSim: the reference's ``lattice.cells`` / ``.beams`` / ``.nodes`` containers, created lazily."""

    @property
    def cells(self):
        return _LazySeq(self.lattice.n_cells, lambda i: CellView(self, i))

    @property
    def beams(self):
        return _LazySeq(_tables(self).n_beams, lambda i: BeamView(self, i))

    @property
    def nodes(self):
        return _LazySeq(_tables(self).n_nodes, lambda i: PointView(self, i))

    def _node_mod_rows(self, name):
        """(P, 6) rows of the penalisation points: displacements come from the closed-form back-substitution of the
        last FEM solve (pl_node_mod), the other vectors are zero / free there."""
        t = _tables(self)
        if getattr(self, "_compat_rows", False):
            return getattr(self, name)[t.n_design_nodes:]
        store = getattr(self, "_node_mod_store", None)
        if store is None or store.get("tables") is not t:
            P = t.n_nodes - t.n_design_nodes
            store = {"tables": t, "displacement_vector": np.zeros((P, 6)), "reaction_force_vector": np.zeros((P, 6)),
                     "applied_force": np.zeros((P, 6)), "fixed_DOF": np.zeros((P, 6), bool)}
            self._node_mod_store = store
            if getattr(self, "_node_mod_pending", None) is None:
                self._node_mod_stale = False        # a regenerated lattice has no results yet: zeros, as after __init__
        pending = getattr(self, "_node_mod_pending", None)
        if pending is None and name == "displacement_vector" and getattr(self, "_node_mod_stale", False):
            raise RuntimeError("displacements of the penalisation points (node_mod) of the last solve were not fetched before "
                               f"the device state changed ({self._node_mod_stale}); look at one of them (or call "
                               "lattice.fetch_node_mod()) right after the solve, or solve again")
        if pending is not None and name == "displacement_vector":
            # first look at a penalisation point after a solve: back-substitute on the device now (B x 12 doubles
            # come back over PCIe - not something every solve should pay for)
            self._node_mod_pending = None
            store[name][:] = pending()[t.pen_strut, t.pen_end]
        return store[name]

    def fetch_node_mod(self):
        """Evaluate the pending back-substitution of the penalisation points now (optimisation loops that change the
        radii afterwards and still want the node_mod displacements of this solve)."""
        return self._node_mod_rows("displacement_vector")

    def set_node_mod_displacement(self, per_strut):
        """Store (B, 2, 6) penalisation-point displacements (``HipLattice.node_mod``) in node order."""
        t = _tables(self)
        self._node_mod_rows("displacement_vector")[:] = np.asarray(per_strut)[t.pen_strut, t.pen_end]

    def get_number_beams(self):
        """len(lattice.beams) as in the reference (lattice.py:202-204): the SEGMENTS once the joints are penalised;
        ``self.lattice.n_beams`` is the number of design struts."""
        return _tables(self).n_beams

    def get_number_nodes(self):
        return _tables(self).n_nodes
