"""Array-backed (SoA) construction of the lattice the hot path consumes.

This is the host-side restatement of what the reference builds as a Python object graph:

* ``generate``      <- Lattice.generate_lattice (lattice.py:421-483) + Cell.generate_beams (cell.py:293-382)
                       + Lattice.define_beam_node_index (lattice.py:665-698)
* ``gradient_table``<- gradient_properties.get_grad_settings (gradient_properties.py:44-137)
* ``compute_lzone`` <- define_connected_beams_for_all_nodes / define_angles_between_beams (lattice.py:805-904),
                       Beam.get_angle_between_beams (beam.py:204-277), function_penalization_Lzone (utils.py:432-453)
* ``penalize``      <- LatticeSim.set_penalized_beams (lattice_sim.py:245-308) + Beam.get_point_on_beam_at_distance
                       (beam.py:279-326) + gmsh subdivision of every segment (lattice_generation.py:50-101)

Everything is numpy; nothing here does stiffness arithmetic (that lives in the HIP library).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from .geometries import get_beam_structure

PENALIZATION_COEFFICIENT = 1.5   # lattice_sim.py:112, beam.py:71
MESH_ELEMENT_LENGTH = 0.05       # lattice_generation.py:50 (fraction of cell_size_x)


def gradient_table(ncx, ncy, ncz, rule, direction, parameters):
    """Rows [fx, fy, fz] for cell index 0..max(n)-1 (gradient_properties.py:44-137)."""
    n = [ncx, ncy, ncz]

    def factor(i, total, p):
        mid = total / 2
        if rule == "constant":
            return 1.0
        if rule == "linear":
            return 1.0 + i * p
        if rule == "parabolic":
            return 1.0 + (i / mid) * p if i < mid else 1.0 + ((total - i - 1) / mid) * p
        if rule == "sinusoide":
            return 1.0 + p * math.sin((i / total) * math.pi)
        if rule == "exponential":
            return 1.0 + math.exp(i * p)
        raise ValueError(f"Unknown gradient rule: {rule}")

    idx = [0, 0, 0]
    rows = []
    for _ in range(max(n)):
        rows.append([factor(idx[d], n[d], parameters[d]) if direction[d] == 1 else 1.0 for d in range(3)])
        for d in range(3):
            if direction[d] == 1 and idx[d] < n[d] - 1:
                idx[d] += 1
    return np.asarray(rows, dtype=np.float64)


@dataclass
class LatticeArrays:
    """Design lattice (before joint penalisation)."""
    node_xyz: np.ndarray            # (N,3) f64, sorted by (x,y,z)  -> node index
    beam_conn: np.ndarray           # (B,2) i32, point1 -> point2 as first created
    beam_radius: np.ndarray         # (B,)  f64
    beam_type: np.ndarray           # (B,)  i32 geometry index
    beam_cell0: np.ndarray          # (B,)  i32 primary cell (cell_belongings[0])
    cell_pos: np.ndarray            # (C,3) i32
    cell_coord: np.ndarray          # (C,3) f64 min corner
    cell_size: np.ndarray           # (C,3) f64
    cell_radii: np.ndarray          # (C,G) f64 radius per geometry (after gradient)
    cell_beam_ptr: np.ndarray       # (C+1,) CSR cell -> beams
    cell_beam_idx: np.ndarray
    cell_node_ptr: np.ndarray       # (C+1,) CSR cell -> nodes
    cell_node_idx: np.ndarray
    bbox: np.ndarray                # (6,) xmin,xmax,ymin,ymax,zmin,zmax
    cell_size_nominal: tuple = (1.0, 1.0, 1.0)
    extras: dict = field(default_factory=dict)

    @property
    def n_nodes(self):
        return len(self.node_xyz)

    @property
    def n_beams(self):
        return len(self.beam_conn)

    @property
    def n_cells(self):
        return len(self.cell_pos)


def _csr_from_pairs(rows, cols, nrows):
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    keep = np.ones(len(rows), bool)
    keep[1:] = (rows[1:] != rows[:-1]) | (cols[1:] != cols[:-1])
    rows, cols = rows[keep], cols[keep]
    ptr = np.concatenate([[0], np.bincount(rows, minlength=nrows)]).astype(np.int64)
    return np.cumsum(ptr), cols.astype(np.int64)


def generate(cell_size, num_cells, geom_types, radii, grad_radius=None, grad_dim=None, erased_blocks=None,
             cell_radii_override=None, cell_range=None, backend="auto", want_creator=False) -> LatticeArrays:
    """Lattice.generate_lattice on arrays: cells in i,j,k order, struts of every geometry, nodes and struts
    de-duplicated through coordinates rounded to 9 decimals (first creator wins, cell.py:312-368).

    ``cell_range`` = ((i0,i1),(j0,j1),(k0,k1)) restricts generation to a box of cells of the SAME global lattice
    (coordinates, gradients and creation order are those of the full lattice) - used by the slab partition.

    ``backend``: "native" = the multi-threaded generator of libpylattice_hip (pl_generate_lattice, host code),
    "numpy" = the vectorised restatement below, "auto" = native when the library is there and accepts the lattice.
    Both give the same arrays bit for bit (tests/test_host_lattice.py).

    ``want_creator``: also return ``extras["node_creator"]``, the position (in the cell list) of the cell that first
    created every node - the reference constructs a ``Point`` exactly there (cell.py:341-362)."""
    nx, ny, nz = num_cells
    csx, csy, csz = cell_size
    if grad_dim is None:
        grad_dim = np.ones((max(nx, ny, nz), 3))
    if grad_radius is None:
        grad_radius = np.ones((max(nx, ny, nz), 3))

    def starts(n, cs, ax):
        s = np.zeros(n)
        for i in range(1, n):
            s[i] = s[i - 1] + cs * grad_dim[i - 1][ax]
        return s

    xs, ys, zs = starts(nx, csx, 0), starts(ny, csy, 1), starts(nz, csz, 2)
    rng = cell_range if cell_range is not None else ((0, nx), (0, ny), (0, nz))
    I, J, K = np.meshgrid(np.arange(*rng[0]), np.arange(*rng[1]), np.arange(*rng[2]), indexing="ij")
    pos = np.stack([I.ravel(), J.ravel(), K.ravel()], axis=1).astype(np.int32)
    coord = np.stack([xs[pos[:, 0]], ys[pos[:, 1]], zs[pos[:, 2]]], axis=1)
    if erased_blocks:
        keep = np.ones(len(pos), bool)
        for blk in erased_blocks:
            inside = np.ones(len(pos), bool)
            for d in range(3):
                inside &= (blk[d] <= coord[:, d]) & (coord[:, d] <= blk[d] + blk[d + 3])
            keep &= ~inside
        pos, coord = pos[keep], coord[keep]
    C = len(pos)
    size = np.stack([csx * grad_dim[pos[:, 0], 0], csy * grad_dim[pos[:, 1], 1], csz * grad_dim[pos[:, 2], 2]], axis=1)
    gfac = grad_radius[pos[:, 0], 0] * grad_radius[pos[:, 1], 1] * grad_radius[pos[:, 2], 2]
    radii = np.asarray(radii, dtype=np.float64)
    cell_radii = radii[None, :] * gfac[:, None]
    if cell_radii_override is not None:
        cell_radii = np.asarray(cell_radii_override, dtype=np.float64).reshape(C, len(radii))

    # template of one cell: all geometries with radius > 0, in geom order (cell.py:272-288)
    tmpl, ttype = [], []
    for g, (name, r) in enumerate(zip(geom_types, radii)):
        if r > 0.0:
            fr = get_beam_structure(name)
            tmpl.append(fr)
            ttype.append(np.full(len(fr), g, np.int32))
    tmpl = np.concatenate(tmpl)
    ttype = np.concatenate(ttype)
    nb = len(tmpl)

    native = None
    if backend in ("auto", "native"):
        try:
            from ._capi import generate_lattice
            native = generate_lattice(coord, size, cell_radii, tmpl, ttype,
                                      want_created=len(geom_types) > 1 or want_creator)
        except (OSError, FileNotFoundError, AttributeError):
            native = None
        if native is None and backend == "native":
            raise RuntimeError("libpylattice_hip's host generator is not available for this lattice")
    extras = {}
    if native is not None:
        node_xyz, beam_conn = native["node_xyz"], native["beam_conn"]
        beam_radius, beam_type, beam_cell0 = native["beam_radius"], native["beam_type"], native["beam_cell0"]
        cb_ptr, cb_idx = native["cell_beam_ptr"], native["cell_beam_idx"]
        cn_ptr, cn_idx = native["cell_node_ptr"], native["cell_node_idx"]
        if want_creator:
            creator = np.full(len(node_xyz), C, np.int64)
            np.minimum.at(creator, native["pid"].reshape(-1).astype(np.int64), np.repeat(np.arange(C), nb * 2))
            extras["node_creator"] = creator
        if len(geom_types) > 1:                                  # lattice.py:482-483
            cell_of = np.repeat(np.arange(C), nb)
            split = _hybrid_collision_split(node_xyz, tmpl, native["pid"].astype(np.int64), native["bid"].astype(np.int64),
                                            beam_conn, beam_radius, beam_type, beam_cell0)
            if split is not None:
                beam_conn, beam_radius, beam_type, beam_cell0, new_of_old_ptr, new_of_old_idx, dmult = split
                extras["design_mult"] = dmult
                pair_beam = native["bid"].astype(np.int64)
                cnt = np.diff(new_of_old_ptr)[pair_beam]
                start = new_of_old_ptr[pair_beam]
                pair_cell = np.repeat(cell_of, cnt)
                within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
                cb_ptr, cb_idx = _csr_from_pairs(pair_cell, new_of_old_idx[np.repeat(start, cnt) + within], C)
        bbox = np.array([node_xyz[:, 0].min(), node_xyz[:, 0].max(), node_xyz[:, 1].min(), node_xyz[:, 1].max(),
                         node_xyz[:, 2].min(), node_xyz[:, 2].max()])
        return LatticeArrays(node_xyz=node_xyz, beam_conn=beam_conn, beam_radius=beam_radius, beam_type=beam_type,
                             beam_cell0=beam_cell0, cell_pos=pos, cell_coord=coord, cell_size=size,
                             cell_radii=cell_radii, cell_beam_ptr=cb_ptr, cell_beam_idx=cb_idx, cell_node_ptr=cn_ptr,
                             cell_node_idx=cn_idx, bbox=bbox, cell_size_nominal=(float(csx), float(csy), float(csz)),
                             extras=extras)

    # end points of every (cell, template strut): frac*size + coordinate  (cell.py:300-305)
    P1 = tmpl[None, :, 0:3] * size[:, None, :] + coord[:, None, :]
    P2 = tmpl[None, :, 3:6] * size[:, None, :] + coord[:, None, :]
    pts = np.stack([P1, P2], axis=2).reshape(-1, 3)             # creation order: cell, strut, end
    key = np.round(pts, 9) + 0.0                                 # +0.0 folds -0.0
    # rows -> one int64 per point through per-axis ranks; lexicographic (x,y,z) order == order of the combined key
    ranks, sizes = [], []
    for ax in range(3):
        uq, inv_ax = np.unique(key[:, ax], return_inverse=True)
        ranks.append(inv_ax.ravel().astype(np.int64))
        sizes.append(len(uq))
    comb = (ranks[0] * sizes[1] + ranks[1]) * sizes[2] + ranks[2]
    _, first, inv = np.unique(comb, return_index=True, return_inverse=True)
    inv = inv.ravel()
    node_xyz = pts[first]                                        # first creator's coordinates, sorted by (x,y,z)
    pid = inv.reshape(C, nb, 2)
    if want_creator:
        extras["node_creator"] = first // (2 * nb)

    N = len(node_xyz)
    lo = np.minimum(pid[..., 0], pid[..., 1]).ravel()
    hi = np.maximum(pid[..., 0], pid[..., 1]).ravel()
    bkey = lo * N + hi
    _, bfirst, binv = np.unique(bkey, return_index=True, return_inverse=True)
    binv = binv.ravel()
    conn_u = pid.reshape(-1, 2)[bfirst]
    cell_of = np.repeat(np.arange(C), nb)
    t_of = np.tile(ttype, C)
    rad_u = cell_radii[cell_of[bfirst], t_of[bfirst]]
    typ_u = t_of[bfirst]
    cell0_u = cell_of[bfirst]
    # beam index = sort by (min point, max point, radius)  (lattice.py:675-685); min point == lower node index
    border = np.lexsort((rad_u, np.maximum(conn_u[:, 0], conn_u[:, 1]), np.minimum(conn_u[:, 0], conn_u[:, 1])))
    brank = np.empty(len(border), np.int64)
    brank[border] = np.arange(len(border))
    beam_conn = conn_u[border].astype(np.int32)
    bid = brank[binv]                                           # (C*nb,) beam index of each created strut

    beam_radius, beam_type, beam_cell0 = rad_u[border], typ_u[border].astype(np.int32), cell0_u[border].astype(np.int32)
    pair_cell, pair_beam = cell_of, bid
    if len(geom_types) > 1:                                      # lattice.py:482-483
        split = _hybrid_collision_split(node_xyz, tmpl, pid, bid, beam_conn, beam_radius, beam_type, beam_cell0)
        if split is not None:
            beam_conn, beam_radius, beam_type, beam_cell0, new_of_old_ptr, new_of_old_idx, dmult = split
            extras["design_mult"] = dmult
            # every (cell, strut) membership goes to all segments of that strut (per_cell_add, lattice.py:1188-1195)
            cnt = np.diff(new_of_old_ptr)[pair_beam]
            start = new_of_old_ptr[pair_beam]
            pair_cell = np.repeat(pair_cell, cnt)
            within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            pair_beam = new_of_old_idx[np.repeat(start, cnt) + within]

    cb_ptr, cb_idx = _csr_from_pairs(pair_cell, pair_beam, C)
    cn_ptr, cn_idx = _csr_from_pairs(np.repeat(np.arange(C), nb * 2), pid.ravel(), C)
    bbox = np.array([node_xyz[:, 0].min(), node_xyz[:, 0].max(), node_xyz[:, 1].min(), node_xyz[:, 1].max(),
                     node_xyz[:, 2].min(), node_xyz[:, 2].max()])
    return LatticeArrays(node_xyz=node_xyz, beam_conn=beam_conn, beam_radius=beam_radius,
                         beam_type=beam_type, beam_cell0=beam_cell0,
                         cell_pos=pos, cell_coord=coord, cell_size=size, cell_radii=cell_radii,
                         cell_beam_ptr=cb_ptr, cell_beam_idx=cb_idx, cell_node_ptr=cn_ptr, cell_node_idx=cn_idx,
                         bbox=bbox, cell_size_nominal=(float(csx), float(csy), float(csz)), extras=extras)


def apply_symmetry(lat: LatticeArrays, geom_types, plane, reference_point) -> LatticeArrays:
    """Lattice.apply_symmetry (lattice.py:497-580) on arrays: every cell gets a twin at the mirrored position - its box
    reflected across the plane through ``reference_point`` (``x'min = 2 x_ref - (x_min + dx)`` for plane "YZ" / "X", and so
    on), same grid position ``pos``, same size and radii, the SAME unit cell (translated, not reflected).

    The reference builds the twins without the lattice's de-duplication tables: each creates its own Point and Beam objects.
    Points hash by exact coordinates, so the node SET merges coincident ones (first object kept, re-indexed by coordinates);
    Beam objects hash by identity, so a strut lying between two twins (or between a twin and its original, on the symmetry
    plane) exists once per creating cell.  Here every strut exists once; ``extras["design_mult"]`` = number of Beam objects
    the reference holds of it, ``extras["design_mult_kind"] = "per_cell"`` (each copy sits in ONE cell's beams_cell, unlike
    the copies of check_hybrid_collision which every owner holds)."""
    plane = plane.upper()
    if plane not in {"XY", "XZ", "YZ", "X", "Y", "Z"}:
        raise ValueError("Invalid symmetry plane. Choose from 'XY', 'XZ', 'YZ', 'X', 'Y', or 'Z'.")
    ax = {"YZ": 0, "X": 0, "XZ": 1, "Y": 1, "XY": 2, "Z": 2}[plane]
    if (np.asarray(lat.extras.get("design_mult", [1])) > 1).any():
        raise NotImplementedError("symmetry of a lattice with struts cut by check_hybrid_collision in several cells")
    C, G = lat.n_cells, lat.cell_radii.shape[1]
    coord = lat.cell_coord.copy()
    coord[:, ax] = 2 * float(reference_point[ax]) - (lat.cell_coord[:, ax] + lat.cell_size[:, ax])
    tmpl, ttype = [], []
    for g, name in enumerate(geom_types):
        fr = get_beam_structure(name)
        tmpl.append(fr)
        ttype.append(np.full(len(fr), g, np.int32))
    tmpl, ttype = np.concatenate(tmpl), np.concatenate(ttype)
    use = lat.cell_radii[:, ttype] > 0.0                                    # (C, nb): geometries with radius > 0 only
    P1 = tmpl[None, :, 0:3] * lat.cell_size[:, None, :] + coord[:, None, :]
    P2 = tmpl[None, :, 3:6] * lat.cell_size[:, None, :] + coord[:, None, :]
    ci, si = np.nonzero(use)
    p1, p2 = P1[ci, si], P2[ci, si]
    rad, typ = lat.cell_radii[ci, ttype[si]], ttype[si]
    # node set: exact coordinates, originals and twins together, sorted by (x, y, z)  (lattice.py:687-696)
    allpts = np.concatenate([lat.node_xyz, p1, p2]) + 0.0
    node_xyz, inv = np.unique(allpts, axis=0, return_inverse=True)
    inv = inv.ravel()
    N0, M = lat.n_nodes, len(ci)
    if len(geom_types) > 1:
        # hybrid lattices were indexed once already (check_hybrid_collision ends with define_beam_node_index,
        # lattice.py:1213) and indices are only handed to objects that have none (lattice.py:665-698): the original
        # nodes and struts keep theirs, the new ones follow in sorted order
        is_old = np.zeros(len(node_xyz), bool)
        is_old[inv[:N0]] = True
        rank = np.empty(len(node_xyz), np.int64)
        rank[inv[:N0]] = np.arange(N0)
        rank[~is_old] = N0 + np.arange((~is_old).sum())
        perm = np.argsort(rank)
        node_xyz, inv = node_xyz[perm], rank[inv]
    old = inv[:N0]
    a_new, b_new = inv[N0:N0 + M], inv[N0 + M:]
    conn = np.concatenate([old[lat.beam_conn.astype(np.int64)], np.stack([a_new, b_new], axis=1)])
    radius = np.concatenate([lat.beam_radius, rad])
    btype = np.concatenate([lat.beam_type, typ])
    cell0 = np.concatenate([lat.beam_cell0, C + ci])
    lo, hi = conn.min(axis=1), conn.max(axis=1)
    # struts that coincide (same two nodes, same radius, same geometry index): one entry, copies counted
    key = np.stack([lo, hi, btype], axis=1)
    if len(geom_types) > 1:
        # beam_key compares COORDINATES; node numbers are no longer in coordinate order here
        crank = np.empty(len(node_xyz), np.int64)
        crank[np.lexsort((node_xyz[:, 2], node_xyz[:, 1], node_xyz[:, 0]))] = np.arange(len(node_xyz))
        klo, khi = np.minimum(crank[lo], crank[hi]), np.maximum(crank[lo], crank[hi])
    else:
        klo, khi = lo, hi
    order = np.lexsort((np.arange(len(conn)), radius, khi, klo))             # beam_key order; first creator first
    key = np.stack([klo, khi, btype], axis=1)
    ks, rs = key[order], radius[order]
    newgrp = np.ones(len(order), bool)
    newgrp[1:] = (ks[1:] != ks[:-1]).any(axis=1) | (rs[1:] != rs[:-1])
    gid = np.cumsum(newgrp) - 1
    first = order[newgrp]
    dmult = np.bincount(gid)
    copy_cell0 = cell0[order]
    if len(geom_types) > 1:
        # struts that hold an original keep the original's index; the purely new ones follow in key order
        has_old = first < lat.n_beams
        newrank = np.empty(len(first), np.int64)
        newrank[has_old] = first[has_old]
        newrank[~has_old] = lat.n_beams + np.arange((~has_old).sum())
        gperm = np.argsort(newrank)
        copy_cell0 = np.concatenate([copy_cell0[gid == g_] for g_ in gperm]) if len(gperm) else copy_cell0
        first, dmult = first[gperm], dmult[gperm]
        gid = newrank[gid]
    uid = np.empty(len(conn), np.int64)
    uid[order] = gid
    B0 = lat.n_beams
    pair_cell = np.concatenate([np.repeat(np.arange(C), np.diff(lat.cell_beam_ptr)), C + ci])
    pair_beam = np.concatenate([uid[lat.cell_beam_idx], uid[B0:]])
    cb_ptr, cb_idx = _csr_from_pairs(pair_cell, pair_beam, 2 * C)
    pair_cn = np.concatenate([np.repeat(np.arange(C), np.diff(lat.cell_node_ptr)), C + ci, C + ci])
    pair_n = np.concatenate([old[lat.cell_node_idx], a_new, b_new])
    cn_ptr, cn_idx = _csr_from_pairs(pair_cn, pair_n, 2 * C)
    bbox = np.array([node_xyz[:, 0].min(), node_xyz[:, 0].max(), node_xyz[:, 1].min(), node_xyz[:, 1].max(),
                     node_xyz[:, 2].min(), node_xyz[:, 2].max()])
    extras = dict(lat.extras)
    # creating cell of every COPY, in strut order (copies of one strut next to each other, first creator first)
    extras.update(design_mult=dmult, design_mult_kind="per_cell", n_original_cells=C, design_copy_cell0=copy_cell0)
    return LatticeArrays(node_xyz=node_xyz, beam_conn=conn[first].astype(np.int32), beam_radius=radius[first],
                         beam_type=btype[first].astype(np.int32), beam_cell0=cell0[first].astype(np.int32),
                         cell_pos=np.concatenate([lat.cell_pos, lat.cell_pos]), cell_coord=np.concatenate([lat.cell_coord, coord]),
                         cell_size=np.concatenate([lat.cell_size, lat.cell_size]),
                         cell_radii=np.concatenate([lat.cell_radii, lat.cell_radii]), cell_beam_ptr=cb_ptr,
                         cell_beam_idx=cb_idx, cell_node_ptr=cn_ptr, cell_node_idx=cn_idx, bbox=bbox,
                         cell_size_nominal=lat.cell_size_nominal, extras=extras)


def random_cell_radii(node_creator, n_cells, n_geom, range_radius, hybrid, seed=44):
    """Per-cell radii of ``enable_randomness`` (lattice.py:426,458-465) from the reference's own random stream.

    generate_lattice seeds the MODULE-level generator (``random.seed(44)``) and draws, for every cell that is not erased,
    ``random.uniform(lo, hi)`` once (or once per geometry with ``randomness_hybrid``).  The same stream is also consumed by
    ``Point.__init__``, which adds ``random.gauss(0, node_uncertainty_SD)`` to each coordinate even when the deviation is
    0 (point.py:55-57): three gauss calls for every NEW point a cell creates, between that cell's draw and the next one.
    ``random.gauss`` produces values in pairs and caches the second, so the calls alternately consume two uniforms and
    none.  Replayed here with ``random.Random(seed)`` and the number of points every cell creates first."""
    import random
    rng = random.Random(seed)
    lo, hi = float(range_radius[0]), float(range_radius[1])
    new_points = np.bincount(node_creator, minlength=n_cells)
    out = np.empty((n_cells, n_geom))
    gauss = rng.gauss
    for c in range(n_cells):
        if hybrid:
            out[c] = [rng.uniform(lo, hi) for _ in range(n_geom)]
        else:
            out[c] = rng.uniform(lo, hi)
        for _ in range(3 * int(new_points[c])):
            gauss(0, 0.0)
    return out


def _hybrid_collision_split(node_xyz, tmpl, pid, bid, beam_conn, beam_radius, beam_type, beam_cell0):
    """Lattice.check_hybrid_collision (lattice.py:1111-1215): in a hybrid lattice a strut of one geometry that passes
    through a node of another geometry OF THE SAME CELL is replaced by the chain of segments between those nodes
    (same radius, type and owner cells; orientation of the original strut).  Returns None when nothing is cut, else the
    re-indexed strut arrays (define_beam_node_index order, lattice.py:665-685), a CSR map old strut -> new struts and the
    number of copies the reference holds of every new strut.

    Two passes: (1) on the unit-cell template, with a tolerance, the few (strut, template point) pairs that can
    collide at all; (2) for those pairs only, in every cell, the reference's exact floating-point tests on the actual
    node coordinates (AABB with 1e-12, Beam.is_point_on_beam beam.py:332-362 with its exact-zero cross product, and
    1e-12 < t < 1 - 1e-12)."""
    C, nb, _ = pid.shape
    # (1) template points and candidate pairs
    tpts, tinv = np.unique(np.round(tmpl.reshape(-1, 3), 9), axis=0, return_inverse=True)
    tp = tinv.ravel().reshape(nb, 2)                               # template point id of every template strut end
    a, b = tpts[tp[:, 0]], tpts[tp[:, 1]]
    v = b - a                                                      # (nb, 3)
    w = tpts[None, :, :] - a[:, None, :]                           # (nb, npt, 3)
    L2 = (v * v).sum(axis=1)
    cr = np.cross(np.broadcast_to(v[:, None, :], w.shape), w)
    t = (w * v[:, None, :]).sum(axis=2) / np.where(L2 > 0, L2, 1.0)[:, None]
    cand = (np.abs(cr).max(axis=2) <= 1e-9) & (t > 1e-9) & (t < 1 - 1e-9) & (L2 > 0)[:, None]
    cs, cq = np.nonzero(cand)
    if len(cs) == 0:
        return None
    # node id of template point q in every cell: any (strut, end) that carries it
    first_slot = np.full(len(tpts), -1, np.int64)
    flat_tp = tp.ravel()
    first_slot[flat_tp[::-1]] = np.arange(len(flat_tp))[::-1]
    pid_flat = pid.reshape(C, nb * 2)
    node_q = pid_flat[:, first_slot[cq]]                          # (C, ncand)
    beam_c = bid.reshape(C, nb)[:, cs]                            # (C, ncand) strut (index before the split)
    p1, p2 = beam_conn[beam_c, 0].astype(np.int64), beam_conn[beam_c, 1].astype(np.int64)
    # (2) exact tests
    A, B_, Nn = node_xyz[p1], node_xyz[p2], node_xyz[node_q]
    vx, vy, vz = B_[..., 0] - A[..., 0], B_[..., 1] - A[..., 1], B_[..., 2] - A[..., 2]
    wx, wy, wz = Nn[..., 0] - A[..., 0], Nn[..., 1] - A[..., 1], Nn[..., 2] - A[..., 2]
    L2 = vx * vx + vy * vy + vz * vz
    tol = 1e-12
    ok = (node_q != p1) & (node_q != p2) & (L2 > 0.0)
    for ax in range(3):
        lo, hi = np.minimum(A[..., ax], B_[..., ax]), np.maximum(A[..., ax], B_[..., ax])
        ok &= (lo - tol <= Nn[..., ax]) & (Nn[..., ax] <= hi + tol)
    ok &= ~((Nn == A).all(axis=-1) | (Nn == B_).all(axis=-1))
    ok &= (vy * wz - vz * wy == 0) & (vz * wx - vx * wz == 0) & (vx * wy - vy * wx == 0)
    dot = wx * vx + wy * vy + wz * vz
    ok &= (0 <= dot) & (dot <= vx ** 2 + vy ** 2 + vz ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        tt = dot / L2
    ok &= (tt > 1e-12) & (tt < 1.0 - 1e-12)
    if not ok.any():
        return None
    cut_beam, cut_node, cut_t = beam_c[ok], node_q[ok], tt[ok]
    # a strut shared by several cells is cut once per owner cell in the reference (the loop does not skip non-primary
    # owners), and every owner gets ALL the copies (per_cell_add, lattice.py:1188-1195): k identical copies of each
    # segment, k = number of cells in which the cut was found.  Here every segment exists once and carries that number
    # (``design_mult``; used by LatticeSim(reference_compat=True) only).  Refused: owner cells that see different cutting
    # nodes on the same strut (the copies would then be different chains).
    cut_cell = np.nonzero(ok)[0]
    per_cell = np.unique(np.stack([cut_cell, cut_beam]), axis=1)
    ncut_cells = np.bincount(per_cell[1], minlength=len(beam_conn))
    tri = np.unique(np.stack([cut_cell, cut_beam, cut_node]), axis=1)
    pairs, cnt = np.unique(tri[1:], axis=1, return_counts=True)
    if (cnt != ncut_cells[pairs[0]]).any():
        raise NotImplementedError(
            "hybrid collision: the owner cells of a shared strut see different cutting nodes (lattice.py:1146-1180 would "
            "leave different chains of segments side by side); this lattice is outside the accelerated path")
    key = np.unique(np.stack([cut_beam, cut_node]), axis=1, return_index=True)[1]
    cut_beam, cut_node, cut_t = cut_beam[key], cut_node[key], cut_t[key]
    order = np.lexsort((cut_t, cut_beam))                        # internal.sort(key=t), lattice.py:1176
    cut_beam, cut_node = cut_beam[order], cut_node[order]
    M = len(cut_beam)
    first = np.ones(M, bool)
    first[1:] = cut_beam[1:] != cut_beam[:-1]
    last = np.ones(M, bool)
    last[:-1] = cut_beam[:-1] != cut_beam[1:]
    seg_a = np.where(first, beam_conn[cut_beam, 0], np.concatenate([[0], cut_node[:-1]]))
    seg_b = cut_node
    tail_a, tail_b = cut_node[last], beam_conn[cut_beam[last], 1]
    parent = np.concatenate([cut_beam, cut_beam[last]])
    sconn = np.stack([np.concatenate([seg_a, tail_a]), np.concatenate([seg_b, tail_b])], axis=1)
    is_cut = np.zeros(len(beam_conn), bool)
    is_cut[cut_beam] = True
    keep = np.flatnonzero(~is_cut)
    conn = np.concatenate([beam_conn[keep].astype(np.int64), sconn])
    src = np.concatenate([keep, parent])                          # strut (old numbering) every new strut comes from
    rad, typ, cell0 = beam_radius[src], beam_type[src], beam_cell0[src]
    dmult = np.concatenate([np.ones(len(keep), np.int64), ncut_cells[parent]])
    lo, hi = conn.min(axis=1), conn.max(axis=1)
    pair = lo * len(node_xyz) + hi
    if len(np.unique(pair)) != len(pair):
        raise NotImplementedError("hybrid collision produces a segment that coincides with another strut (the reference "
                                  "keeps both); this lattice is outside the accelerated path")
    border = np.lexsort((rad, hi, lo))                            # beam_key, lattice.py:675-680
    rank = np.empty(len(border), np.int64)
    rank[border] = np.arange(len(border))
    o = np.argsort(src, kind="stable")
    ptr = np.zeros(len(beam_conn) + 1, np.int64)
    np.add.at(ptr, src + 1, 1)
    return (conn[border].astype(np.int32), rad[border], typ[border], cell0[border], np.cumsum(ptr), rank[o], dmult[border])


# ------------------------------------------------------------------------------------------------
# joint penalisation
# ------------------------------------------------------------------------------------------------
def _libm(fn, arr):
    """fn (math.acos, math.tan, ...) applied element-wise through the C library, as the reference's scalar code does:
    numpy's vectorised arccos / tan may differ from libm in the last bit, and L_zone decides where the penalisation points
    lie - bit-exact only with the same function.  Evaluated once per DISTINCT value (lattices repeat a handful of
    angles); arrays beyond a few million entries keep numpy's own loops (large lattices take the device kernel pl_lzone)."""
    arr = np.asarray(arr, dtype=np.float64)
    if arr.size > 4_000_000:
        return {math.acos: np.arccos, math.tan: np.tan}[fn](arr)
    uq, inv = np.unique(arr.ravel(), return_inverse=True)
    vals = np.array([fn(float(v)) if np.isfinite(v) else np.nan for v in uq], dtype=np.float64)
    return vals[inv].reshape(arr.shape)


def _lzone_of(radius, angle_deg):
    """function_penalization_Lzone (utils.py:432-453), vectorised."""
    with np.errstate(divide="ignore", invalid="ignore"):
        L = radius / _libm(math.tan, np.radians(angle_deg) / 2.0)
    L = np.where(angle_deg > 170.0, 0.0000001, L)
    return np.where(angle_deg == 0.0, 0.0, L)


def point_tags(xyz, box):
    """Point.tag_point (point.py:169-235) for many points; -1 where the reference returns None."""
    xmin, xmax, ymin, ymax, zmin, zmax = box
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    ex = [(x == xmin), (x > xmin) & (x < xmax), (x == xmax)]
    ey = [(y == ymin), (y > ymin) & (y < ymax), (y == ymax)]
    ez = [(z == zmin), (z > zmin) & (z < zmax), (z == zmax)]
    tag = np.full(len(xyz), -1, np.int64)
    table = {  # (ix, iy, iz) with 0=min 1=inside 2=max
        (0, 1, 1): 12, (2, 1, 1): 13, (1, 0, 1): 11, (1, 2, 1): 14, (1, 1, 0): 10, (1, 1, 2): 15,
        (0, 0, 1): 102, (1, 0, 0): 100, (2, 0, 1): 104, (1, 0, 2): 108, (0, 1, 0): 101, (2, 1, 0): 103,
        (0, 2, 1): 106, (1, 2, 0): 105, (2, 2, 1): 107, (1, 2, 2): 111, (0, 1, 2): 109, (2, 1, 2): 110,
        (0, 0, 0): 1000, (2, 0, 0): 1001, (0, 2, 0): 1002, (2, 2, 0): 1003, (0, 0, 2): 1004, (2, 0, 2): 1005,
        (0, 2, 2): 1006, (2, 2, 2): 1007}
    for (i, j, k), code in table.items():
        tag[ex[i] & ey[j] & ez[k]] = code
    return tag


_EDGE_GROUPS = [[102, 104, 106, 107], [100, 108, 105, 111], [101, 109, 103, 110]]
_FACE_GROUPS = [[10, 15], [11, 14], [12, 13]]


def _angle_periodic(xyz, tag, ba, bb):
    """Beam.get_angle_between_beams with periodicity=True, scalar restatement of beam.py:204-277
    (including its 'last matching candidate wins' loops).  ba/bb = (p1, p2) node ids."""
    p1 = p2 = None

    def scan(group):
        nonlocal p1, p2
        for i1, c1 in enumerate(ba):
            if tag[c1] > 0 and tag[c1] in group:
                p1 = i1
                for i2, c2 in enumerate(bb):
                    if tag[c2] > 0 and tag[c2] in group:
                        p2 = i2
                        break

    scan(range(1000, 1008))
    if p1 is None and p2 is None:
        for g in _EDGE_GROUPS:
            scan(g)
    if p1 is None and p2 is None:
        for g in _FACE_GROUPS:
            scan(g)
    a1, a2 = ba
    b1, b2 = bb
    if a1 == b1 or (p1 == 0 and p2 == 0):
        u, v = xyz[a2] - xyz[a1], xyz[b2] - xyz[b1]
    elif a1 == b2 or (p1 == 0 and p2 == 1):
        u, v = xyz[a2] - xyz[a1], xyz[b1] - xyz[b2]
    elif a2 == b1 or (p1 == 1 and p2 == 0):
        u, v = xyz[a1] - xyz[a2], xyz[b2] - xyz[b1]
    elif a2 == b2 or (p1 == 1 and p2 == 1):
        u, v = xyz[a1] - xyz[a2], xyz[b1] - xyz[b2]
    else:
        return None
    c = float(u @ v) / (math.sqrt(float(u @ u)) * math.sqrt(float(v @ v)))
    return math.degrees(math.acos(max(min(c, 1.0), -1.0)))


def compute_lzone(lat: LatticeArrays, periodicity: bool = False) -> np.ndarray:
    """(B,2) penalisation length at each strut end: over the other struts meeting at that node, the one that
    maximises L = r_other / tan(angle/2) (lattice.py:871-904)."""
    xyz, conn, rad = lat.node_xyz, lat.beam_conn.astype(np.int64), lat.beam_radius
    B, N = len(conn), len(xyz)
    if not periodicity:
        # half-edges sorted by node
        he_node = conn.ravel()                                   # (2B,) node of half-edge h = 2*b + end
        he_far = conn[:, ::-1].ravel()
        dirv = xyz[he_far] - xyz[he_node]
        dnorm = np.sqrt(dirv[:, 0] * dirv[:, 0] + dirv[:, 1] * dirv[:, 1] + dirv[:, 2] * dirv[:, 2])
        order = np.argsort(he_node, kind="stable")
        ptr = np.zeros(N + 1, np.int64)
        np.add.at(ptr, he_node + 1, 1)
        ptr = np.cumsum(ptr)
        deg = np.diff(ptr)
        lz = np.zeros(2 * B)
        # group nodes by valence so each group is a dense (n, v, v) problem
        for v in np.unique(deg):
            if v < 2:
                continue
            nodes = np.flatnonzero(deg == v)
            hes = order[ptr[nodes][:, None] + np.arange(v)[None, :]]          # (n, v) half-edge ids
            d = dirv[hes]                                                     # (n, v, 3)
            # same operation order as beam.py:269-273: ((x1 x2 + y1 y2) + z1 z2) / (|u| |v|)
            dot = (d[:, :, None, 0] * d[:, None, :, 0] + d[:, :, None, 1] * d[:, None, :, 1]
                   + d[:, :, None, 2] * d[:, None, :, 2])
            nn = dnorm[hes]
            cosang = np.clip(dot / (nn[:, :, None] * nn[:, None, :]), -1.0, 1.0)
            ang = np.degrees(_libm(math.acos, cosang))
            r_other = rad[hes // 2][:, None, :] * np.ones((1, v, 1))          # (n, i, j) radius of j
            L = _lzone_of(r_other, ang)
            valid = (ang > 1e-12) & ~np.eye(v, dtype=bool)[None]
            L = np.where(valid, L, -1.0)
            best = L.max(axis=2)
            lz[hes] = np.where(best < 0.0, 0.0, best)
        return lz.reshape(B, 2)

    # periodic lattices (small single-cell models used for the Schur datasets): scalar restatement
    tag = point_tags(xyz, lat.bbox)
    xmin, xmax, ymin, ymax, zmin, zmax = lat.bbox
    inc = [[] for _ in range(N)]
    for b, (a, c) in enumerate(conn):
        inc[a].append(b)
        inc[c].append(b)
    tol = 1e-9
    w = xyz.copy()
    w[np.abs(w[:, 0] - xmax) <= tol, 0] = xmin
    w[np.abs(w[:, 1] - ymax) <= tol, 1] = ymin
    w[np.abs(w[:, 2] - zmax) <= tol, 2] = zmin
    buckets = {}
    for i, k in enumerate(map(tuple, np.round(w / tol).astype(np.int64))):
        buckets.setdefault(k, []).append(i)
    merged = [None] * N
    for ids in buckets.values():
        s = sorted(set(b for i in ids for b in inc[i]))
        for i in ids:
            merged[i] = s
    lz = np.zeros((B, 2))
    for b in range(B):
        for e in range(2):
            best = -1.0
            for nb in merged[conn[b, e]]:
                if nb == b:
                    continue
                ang = _angle_periodic(xyz, tag, tuple(conn[b]), tuple(conn[nb]))
                if ang is None or not ang > 1e-12:
                    continue
                L = float(_lzone_of(np.float64(rad[nb]), np.float64(ang)))
                if L > best:
                    best = L
            lz[b, e] = 0.0 if best < 0 else best
    return lz


@dataclass
class PenalizedBeams:
    """Per design-lattice strut: the up-to-three colinear segments it is meshed as."""
    seg_len: np.ndarray      # (B,3) geometric length of [pen@point1, middle, pen@point2]; 0 where absent
    seg_nsub: np.ndarray     # (B,3) i32 number of equal P1 sub-elements gmsh puts on each segment
    pen_xyz: np.ndarray      # (B,2,3) coordinates of the two penalisation points (NaN where absent)
    lzone: np.ndarray        # (B,2)


def gmsh_subdivisions(length, h):
    """int(length/h + 0.99), >= 1 where length > 0 (gmsh 1-D mesher with a uniform size field)."""
    n = np.floor(np.asarray(length) / h + 0.99).astype(np.int32)
    return np.where(np.asarray(length) > 0, np.maximum(n, 1), 0).astype(np.int32)


def penalize(lat: LatticeArrays, lzone: np.ndarray | None, mesh_size: float | None = None,
             backend="auto") -> PenalizedBeams:
    """Split every strut into pen(L1) + middle + pen(L2) exactly where the reference puts the new points
    (start + (end-start)/round(length,4) * L, beam.py:300-312) and count gmsh sub-elements per segment.
    ``backend`` as in ``generate`` (pl_penalize / numpy; same bits)."""
    xyz, conn = lat.node_xyz, lat.beam_conn
    B = len(conn)
    h = (MESH_ELEMENT_LENGTH * lat.cell_size_nominal[0]) if mesh_size is None else mesh_size
    if backend in ("auto", "native"):
        try:
            from ._capi import penalize_arrays
            seg_len, seg_nsub, pen_xyz = penalize_arrays(xyz, conn, lzone, h)
            return PenalizedBeams(seg_len=seg_len, seg_nsub=seg_nsub, pen_xyz=pen_xyz,
                                  lzone=np.zeros((B, 2)) if lzone is None else np.asarray(lzone, float))
        except (OSError, FileNotFoundError, AttributeError):
            if backend == "native":
                raise
    pa, pb = xyz[conn[:, 0]], xyz[conn[:, 1]]
    d = pb - pa
    true_len = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2 + d[:, 2] ** 2)
    if lzone is None:
        lzone = np.zeros((B, 2))
    uniq, inv = np.unique(true_len, return_inverse=True)
    len4 = np.array([round(float(v), 4) for v in uniq])[inv.ravel()]          # Beam.length (beam.py:135)
    L1, L2 = lzone[:, 0], lzone[:, 1]
    has1, has2 = L1 > 0, L2 > 0
    q1 = pa + (d / len4[:, None]) * L1[:, None]
    q2 = pb + ((-d) / len4[:, None]) * L2[:, None]
    start = np.where(has1[:, None], q1, pa)
    mid_end = np.where(has2[:, None], q2, pb)

    def dist(u, v):
        w = v - u
        return np.sqrt(w[:, 0] ** 2 + w[:, 1] ** 2 + w[:, 2] ** 2)

    seg_len = np.stack([np.where(has1, dist(pa, q1), 0.0), dist(start, mid_end),
                        np.where(has2, dist(q2, pb), 0.0)], axis=1)
    seg_nsub = gmsh_subdivisions(seg_len, h)
    pen_xyz = np.stack([np.where(has1[:, None], q1, np.nan), np.where(has2[:, None], q2, np.nan)], axis=1)
    return PenalizedBeams(seg_len=seg_len, seg_nsub=seg_nsub, pen_xyz=pen_xyz, lzone=np.asarray(lzone, float))
