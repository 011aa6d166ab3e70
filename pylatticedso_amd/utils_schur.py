"""Drop-in for ``get_schur_complement`` (src/pyLatticeSim/utils_schur.py:22-53) on MI355X."""
from __future__ import annotations

import os

import numpy as np


def node_order_to_simulate(lattice, cell_index=0, tol=1e-9):
    """Cell.define_node_order_to_simulate (cell.py:611-680): boundary nodes bucketed by the first face they lie on
    in the priority Xmin,Xmax,Ymin,Ymax,Zmin,Zmax, each bucket sorted by its in-plane coordinates."""
    lat = lattice.lattice
    nodes = lat.cell_node_idx[lat.cell_node_ptr[cell_index]:lat.cell_node_ptr[cell_index + 1]]
    nodes = nodes[lattice.index_boundary[nodes] >= 0]
    xyz = lat.node_xyz[nodes]
    lo = lat.cell_coord[cell_index]
    hi = lo + lat.cell_size[cell_index]
    on = np.stack([np.abs(xyz[:, 0] - lo[0]) <= tol, np.abs(xyz[:, 0] - hi[0]) <= tol,
                   np.abs(xyz[:, 1] - lo[1]) <= tol, np.abs(xyz[:, 1] - hi[1]) <= tol,
                   np.abs(xyz[:, 2] - lo[2]) <= tol, np.abs(xyz[:, 2] - hi[2]) <= tol], axis=1)
    face = np.argmax(on, axis=1)
    keys = {0: (1, 2, 0), 1: (1, 2, 0), 2: (0, 2, 1), 3: (0, 2, 1), 4: (0, 1, 2), 5: (0, 1, 2)}
    ordered = []
    for f in range(6):
        sel = np.flatnonzero(face == f)
        k = keys[f]
        o = np.lexsort((xyz[sel, k[2]], xyz[sel, k[1]], xyz[sel, k[0]]))
        ordered.extend(nodes[sel[o]])
    return np.asarray(ordered, dtype=np.int64)


def node_order_all_cells(lattice, tol=1e-9):
    """node_order_to_simulate for every cell at once: (C, n_b) node ids, or None when the cells do not all have the
    same number of boundary nodes.  One lexsort over all (cell, node) pairs instead of a Python loop per cell."""
    lat = lattice.lattice
    C = lat.n_cells
    cell = np.repeat(np.arange(C), np.diff(lat.cell_node_ptr))
    nodes = lat.cell_node_idx
    keep = lattice.index_boundary[nodes] >= 0
    cell, nodes = cell[keep], nodes[keep]
    counts = np.bincount(cell, minlength=C)
    if C == 0 or not np.all(counts == counts[0]):
        return None
    xyz = lat.node_xyz[nodes]
    lo = lat.cell_coord[cell]
    hi = lo + lat.cell_size[cell]
    on = np.stack([np.abs(xyz[:, 0] - lo[:, 0]) <= tol, np.abs(xyz[:, 0] - hi[:, 0]) <= tol,
                   np.abs(xyz[:, 1] - lo[:, 1]) <= tol, np.abs(xyz[:, 1] - hi[:, 1]) <= tol,
                   np.abs(xyz[:, 2] - lo[:, 2]) <= tol, np.abs(xyz[:, 2] - hi[:, 2]) <= tol], axis=1)
    face = np.argmax(on, axis=1)
    # in-plane sort keys of the six faces: (y, z, x), (y, z, x), (x, z, y), (x, z, y), (x, y, z), (x, y, z)
    perm = np.array([[1, 2, 0], [1, 2, 0], [0, 2, 1], [0, 2, 1], [0, 1, 2], [0, 1, 2]])[face]
    k = np.take_along_axis(xyz, perm, axis=1)
    order = np.lexsort((k[:, 2], k[:, 1], k[:, 0], face, cell))
    return nodes[order].reshape(C, counts[0]).astype(np.int64)


def get_schur_complement(lattice, cell_index=None, rtol=1e-13, max_iter=200000):
    """S = K_BB - K_BI K_II^-1 K_IB of one cell on its boundary nodes, (6 n_b, 6 n_b), node order as the reference."""
    if cell_index is None and lattice.get_number_cells() > 1:
        raise ValueError("The lattice must contain only one cell for Schur complement calculation or specify a "
                         "cell_index.")
    if lattice.get_number_cells() > 1:
        # BeamModel(..., cell_index) meshes the struts and points listed in THAT cell (lattice_generation.py:104-175:
        # cell.points_cell / cell.beams_cell - struts shared with a neighbour included, with the radius and the
        # penalised end segments they have in the whole lattice); condensation on the cell's boundary nodes in
        # Cell.define_node_order_to_simulate order (utils_schur.py:36-41)
        dev, order = cell_device(lattice, int(cell_index))
        with dev:
            dev.assemble()
            return dev.schur(order, rtol=rtol, max_iter=max_iter)
    order = node_order_to_simulate(lattice, 0)
    dev = lattice.device_model()
    dev.assemble()
    return dev.schur(order, rtol=rtol, max_iter=max_iter)


def cell_device(lattice, cell_index):
    """(HipLattice of the sub-lattice made of one cell's struts and nodes, boundary nodes of that cell in the
    reference's simulation order, numbered inside the sub-lattice)."""
    from ._capi import HipLattice
    lat, pen = lattice.lattice, lattice.penalized
    if not 0 <= cell_index < lat.n_cells:
        raise IndexError("cell_index out of range")
    struts = np.unique(lat.cell_beam_idx[lat.cell_beam_ptr[cell_index]:lat.cell_beam_ptr[cell_index + 1]])
    struts = struts[lat.beam_radius[struts] > 0]                       # (lattice_generation.py:158)
    nodes = np.unique(np.concatenate([lat.cell_node_idx[lat.cell_node_ptr[cell_index]:lat.cell_node_ptr[cell_index + 1]],
                                      lat.beam_conn[struts].ravel()]))
    local = np.full(lat.n_nodes, -1, np.int64)
    local[nodes] = np.arange(len(nodes))
    # one cell: a few hundred dofs - the dense factor of P K P as the preconditioner (precond = 5), so that each of the 6 n_b
    # condensation solves of pl_schur is one or two PCG steps instead of hundreds of Jacobi iterations
    dev = HipLattice(lat.node_xyz[nodes], local[lat.beam_conn[struts]], lat.beam_radius[struts], pen.seg_len[struts],
                     pen.seg_nsub[struts], lattice.young_modulus, lattice.poisson_ratio,
                     **({"precond": 5} if 6 * len(nodes) <= 16384 else {}))
    return dev, local[node_order_to_simulate(lattice, cell_index)]


def define_path_schur_complement(lattice_object):
    """``<repo>/data/outputs/schur_complement/Schur_complement_<geoms>.npz`` (utils_schur.py:74-95)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name = "Schur_complement_" + "_".join(str(g) for g in lattice_object.geom_types) + ".npz"
    return os.path.join(root, "data", "outputs", "schur_complement", name)


def save_schur_complement_npz(lattice_object, radius_values, schur_matrices):
    """utils_schur.py:55-72: the dataset the reduced-basis construction starts from."""
    path = define_path_schur_complement(lattice_object)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez(path, radius_values=np.array(radius_values), schur_matrices=np.array(schur_matrices))
    print("Schur complement data saved to", path)


def load_schur_complement_dataset(lattice_object, enable_normalization: bool = False):
    """utils_schur.py:97-129: {radius tuple: matrix}."""
    data = np.load(define_path_schur_complement(lattice_object), allow_pickle=True)
    radius_values, schur_matrices = data["radius_values"], data["schur_matrices"]
    if np.ndim(radius_values[0]) == 0 or np.ndim(radius_values) == 1:
        out = {tuple(np.atleast_1d(radius_values)): schur_matrices}
    else:
        out = {tuple(r): m for r, m in zip(radius_values, schur_matrices)}
    if enable_normalization:
        out = {k: m / np.linalg.norm(m) for k, m in out.items()}
    return out
