"""Slab partition of a lattice over the GPUs of one node (one process per GPU).

The stiffness operator is a sum of independent per-strut products, so the lattice shards by unit-cell slab: rank r
owns the cells of a contiguous range of layers along ``axis`` and every strut whose FIRST creating cell
(``cell_belongings[0]``, reference beam.py:59-62) lies in that range.  Nodes on the planes between two slabs exist
on both ranks; after each local K*x their partial forces are summed (RCCL all-reduce of the packed interface
vector inside libpylattice_hip) and dot products count them once.

One ghost layer of cells on each side is generated so that the joint-penalisation lengths (which depend on ALL
struts meeting at a node, lattice.py:871-904) are identical to those of the un-partitioned lattice.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import lattice_arrays as LA


def slab_range(n_layers: int, rank: int, world: int):
    base, rem = divmod(n_layers, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass
class Slab:
    rank: int
    world: int
    axis: int
    layers: tuple                 # (lo, hi) owned cell layers along axis
    node_xyz: np.ndarray          # (n,3) local nodes (sorted by xyz like the global lattice)
    beam_conn: np.ndarray         # (b,2) local node ids
    beam_radius: np.ndarray
    seg_len: np.ndarray           # (b,3)
    seg_nsub: np.ndarray          # (b,3)
    iface_local: np.ndarray       # local ids of nodes lying on an interface plane of this slab
    iface_key: np.ndarray         # (m,4) int64 key (plane, qa, qb, 0) identifying the node across ranks
    iface_peer: np.ndarray = None  # (m,) rank of the slab on the other side of the node's plane (neighbour exchange)
    n_owned_beams_global: int = 0


def build_slab(cell_size, num_cells, geom_types, radii, rank, world, axis=1, grad_radius=None,
               mesh_size=None) -> Slab:
    n = num_cells[axis]
    if world > n:
        raise ValueError("more ranks than cell layers along the partition axis")
    lo, hi = slab_range(n, rank, world)
    g0, g1 = max(0, lo - 1), min(n, hi + 1)
    rng = [(0, num_cells[0]), (0, num_cells[1]), (0, num_cells[2])]
    rng[axis] = (g0, g1)
    lat = LA.generate(cell_size, num_cells, geom_types, radii, grad_radius=grad_radius, cell_range=tuple(rng))
    lz = LA.compute_lzone(lat, False)
    layer = lat.cell_pos[lat.beam_cell0, axis]
    own = (layer >= lo) & (layer < hi)
    conn = lat.beam_conn[own]
    used = np.unique(conn)
    remap = np.full(lat.n_nodes, -1, np.int64)
    remap[used] = np.arange(len(used))
    sub = LA.LatticeArrays(node_xyz=lat.node_xyz[used], beam_conn=remap[conn].astype(np.int32),
                           beam_radius=lat.beam_radius[own], beam_type=lat.beam_type[own],
                           beam_cell0=lat.beam_cell0[own], cell_pos=lat.cell_pos, cell_coord=lat.cell_coord,
                           cell_size=lat.cell_size, cell_radii=lat.cell_radii, cell_beam_ptr=lat.cell_beam_ptr,
                           cell_beam_idx=lat.cell_beam_idx, cell_node_ptr=lat.cell_node_ptr,
                           cell_node_idx=lat.cell_node_idx, bbox=lat.bbox, cell_size_nominal=lat.cell_size_nominal)
    pen = LA.penalize(sub, lz[own], mesh_size=mesh_size)
    # interface planes: lower face of layer `lo` (if rank > 0) and upper face of layer `hi-1` (if rank < world-1)
    xyz = sub.node_xyz
    planes = []
    cpos, ccoord, csize = lat.cell_pos[:, axis], lat.cell_coord[:, axis], lat.cell_size[:, axis]
    if rank > 0:
        planes.append((lo, float(ccoord[cpos == lo][0]), rank - 1))
    if rank < world - 1:
        c = np.flatnonzero(cpos == hi - 1)[0]
        planes.append((hi, float(ccoord[c] + csize[c]), rank + 1))
    others = [a for a in range(3) if a != axis]
    loc, keys, peers = [], [], []
    for pid, coord, peer in planes:
        sel = np.flatnonzero(np.abs(xyz[:, axis] - coord) <= 1e-9)
        q = np.round(xyz[sel][:, others] * 1e6).astype(np.int64)
        loc.append(sel)
        keys.append(np.c_[np.full(len(sel), pid, np.int64), q, np.zeros(len(sel), np.int64)])
        peers.append(np.full(len(sel), peer, np.int32))
    iface_local = np.concatenate(loc) if loc else np.zeros(0, np.int64)
    iface_key = np.concatenate(keys) if keys else np.zeros((0, 4), np.int64)
    iface_peer = np.concatenate(peers) if peers else np.zeros(0, np.int32)
    return Slab(rank=rank, world=world, axis=axis, layers=(lo, hi), node_xyz=xyz, beam_conn=sub.beam_conn,
                beam_radius=sub.beam_radius, seg_len=pen.seg_len, seg_nsub=pen.seg_nsub,
                iface_local=iface_local, iface_key=iface_key, iface_peer=iface_peer)


def global_interface_ids(all_keys: list[np.ndarray], rank: int):
    """From every rank's interface keys: dense global ids of the nodes present on >= 2 ranks.
    Returns (mask over this rank's interface nodes, their global ids, n_shared_global)."""
    cat = np.concatenate([k for k in all_keys if len(k)]) if any(len(k) for k in all_keys) else np.zeros((0, 4), np.int64)
    if len(cat) == 0:
        return np.zeros(0, bool), np.zeros(0, np.int32), 0
    uq, counts = np.unique(cat, axis=0, return_counts=True)
    shared = uq[counts >= 2]
    mine = all_keys[rank]
    # position of each of my keys in `shared` (or -1)
    def as_void(a):
        a = np.ascontiguousarray(a)
        return a.view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).ravel()
    sv, mv = as_void(shared), as_void(mine)
    order = np.argsort(sv)
    pos = np.searchsorted(sv[order], mv)
    pos = np.clip(pos, 0, max(len(sv) - 1, 0))
    ok = (len(sv) > 0) & (sv[order][pos] == mv) if len(mv) else np.zeros(0, bool)
    gid = order[pos]
    return ok, gid[ok].astype(np.int32), len(shared)
