"""R slab handles of ONE process on ONE GPU, joined by the library's loopback transport (pl_dist_loopback_id).

The multi-GPU solver path of libpylattice_hip (slab partition, interface exchange, weighted dot products, all-reduced
coarse level, single-reduction PCG, node elimination on multi-rank handles) normally needs one process per GPU and RCCL.
With the loopback transport the SAME device code runs with world = 2 ... 16 on a one-GPU box: every rank is a
``HipLattice`` of its own, every collective call is made by one host thread per rank (ctypes releases the GIL), and the
contributions meet in device buffers instead of travelling over xGMI.  ``tests/test_gpu_loopback.py`` holds the result to
the single-handle solve; ``bench.py --loopback R`` rehearses the partitioned configurations (BASELINE.json configs[2],
configs[4]) for iteration counts and per-rank kernel times.
"""
from __future__ import annotations

import threading

import numpy as np

from . import _capi, lattice_arrays as LA, partition as PT


def _keys(xyz):
    q = np.round(np.asarray(xyz) * 1e6).astype(np.int64)
    q = np.ascontiguousarray(q)
    return q.view(np.dtype((np.void, q.dtype.itemsize * 3))).ravel()


def match_nodes(xyz, global_xyz):
    """Index into ``global_xyz`` of every row of ``xyz`` (coordinates compared at 1e-6)."""
    gk, k = _keys(global_xyz), _keys(xyz)
    order = np.argsort(gk)
    pos = np.searchsorted(gk[order], k)
    idx = order[np.clip(pos, 0, len(gk) - 1)]
    if not np.array_equal(gk[idx], k):
        raise ValueError("slab node without a partner in the global lattice")
    return idx


class LoopbackGroup:
    """``world`` slabs of a lattice (cut along ``axis`` by :func:`partition.build_slab`), one device handle each."""

    def __init__(self, cell_size, num_cells, geom_types, radii, world, axis=0, young=1013.0, poisson=0.3, device=0,
                 grad_radius=None, p2p=True, **opts):
        self.world, self.axis, self.num_cells = int(world), int(axis), tuple(num_cells)
        self.slabs = [PT.build_slab(cell_size, num_cells, geom_types, radii, r, world, axis=axis, grad_radius=grad_radius)
                      for r in range(world)]
        n_nodes_sum = int(sum(len(s.node_xyz) for s in self.slabs))
        lo = tuple(float(min(s.node_xyz[:, k].min() for s in self.slabs)) for k in range(3))
        hi = tuple(float(max(s.node_xyz[:, k].max() for s in self.slabs)) for k in range(3))
        self.grid = (lo, hi, n_nodes_sum)
        self.devs = [None] * world
        try:
            self._build(young, poisson, device, p2p, opts)
        except BaseException:
            self.close()                   # a half-built group must not keep its handles (and their registry attachments)
            raise

    def _build(self, young, poisson, device, p2p, opts):
        world = self.world

        def create(r):
            s = self.slabs[r]
            self.devs[r] = _capi.HipLattice(s.node_xyz, s.beam_conn, s.beam_radius, s.seg_len, s.seg_nsub, young, poisson,
                                            device=device, grid=self.grid, **opts)
        self.each(create)
        keys = [s.iface_key for s in self.slabs]
        uid = _capi.HipLattice.dist_loopback_id()
        shared = [PT.global_interface_ids(keys, r) for r in range(world)]
        self.shared_local = [self.slabs[r].iface_local[shared[r][0]] for r in range(world)]

        def attach(r):
            ok, gid, nsg = shared[r]
            peers = self.slabs[r].iface_peer[ok] if p2p else None
            self.devs[r].dist_init(r, world, uid, self.shared_local[r], gid, nsg, shared_peer=peers)
        self.each(attach)
        self.n_beams = int(sum(len(s.beam_conn) for s in self.slabs))

    # -- one host thread per rank ------------------------------------------------------------------------------------
    def each(self, fn):
        """fn(rank) on one thread per rank (every collective call of the library must be made this way)."""
        out, err, lock = [None] * self.world, [], threading.Lock()

        def run(r):
            try:
                out[r] = fn(r)
            except BaseException as e:     # noqa: BLE001 - re-raised below, on the caller's thread
                with lock:
                    err.append(e)          # in the order the ranks failed
                # the other ranks may be waiting for this one at a collective: let them go.  The broken flag belongs to
                # the group, so any attached handle will do (this rank's own may not exist yet)
                for d in [self.devs[r] if r < len(self.devs) else None] + list(self.devs):
                    if d is not None:
                        d.dist_abort()
                        break
        th = [threading.Thread(target=run, args=(r,), name=f"pl-rank-{r}") for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            raise err[0]                   # the rank that failed first, not the ones that were let go afterwards
        return out

    # -- the collective calls ----------------------------------------------------------------------------------------
    def set_bc(self, fixed, ubar=None, f=None):
        """Per-rank lists of (n_r, 6) arrays (a shared node gets the SAME, full values on every rank that holds it)."""
        self.each(lambda r: self.devs[r].set_bc(fixed[r], None if ubar is None else ubar[r], None if f is None else f[r]))

    def assemble(self):
        self.each(lambda r: self.devs[r].assemble())

    def solve(self, **kw):
        """[(u_r, stats_r)] - every rank reports the same iteration count and residual."""
        return self.each(lambda r: self.devs[r].solve(**kw))

    def spmv_free(self, x):
        return self.each(lambda r: self.devs[r].spmv_free(x[r]))

    def time_kernel(self, which, reps=20):
        return self.each(lambda r: self.devs[r].time_kernel(which, reps))

    def close(self):
        for d in self.devs:
            if d is not None:
                d.close()
        self.devs = [None] * self.world

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- whole-lattice views -----------------------------------------------------------------------------------------
    def cantilever(self, x_max, total_fz=-0.1):
        """The bench's boundary conditions per rank: clamp Xmin, total force in Z spread over the Xmax nodes of the WHOLE
        lattice (shared nodes counted once)."""
        xyz = np.concatenate([s.node_xyz for s in self.slabs])
        n_tgt = len(np.unique(_keys(xyz[xyz[:, 0] == x_max])))
        fixed, f = [], []
        for s in self.slabs:
            fx = np.zeros((len(s.node_xyz), 6), np.uint8)
            fx[s.node_xyz[:, 0] == 0.0] = 1
            ff = np.zeros((len(s.node_xyz), 6))
            ff[s.node_xyz[:, 0] == x_max, 2] = total_fz / n_tgt
            fixed.append(fx)
            f.append(ff)
        return fixed, f

    def scatter(self, global_xyz, values):
        """Rows of a whole-lattice (N, 6) array for every rank's nodes."""
        return [np.asarray(values)[match_nodes(s.node_xyz, global_xyz)] for s in self.slabs]

    def gather(self, global_xyz, per_rank):
        """Whole-lattice (N, 6) array from per-rank arrays; copies of a shared node must agree (checked to 1e-9 of the
        field's size)."""
        out = np.full((len(global_xyz), 6), np.nan)
        scale = max(float(np.abs(np.concatenate([np.ravel(a) for a in per_rank])).max()), 1e-300)
        for s, a in zip(self.slabs, per_rank):
            idx = match_nodes(s.node_xyz, global_xyz)
            a = np.asarray(a).reshape(-1, 6)
            seen = ~np.isnan(out[idx, 0])
            if seen.any() and np.abs(out[idx[seen]] - a[seen]).max() > 1e-9 * scale:
                raise ValueError("copies of a shared node disagree between ranks")
            out[idx] = a
        if np.isnan(out).any():
            raise ValueError("global node not covered by any slab")
        return out


def whole_lattice(cell_size, num_cells, geom_types, radii, grad_radius=None):
    """The un-partitioned lattice + penalisation the slabs are cut from (for single-handle comparisons)."""
    lat = LA.generate(cell_size, num_cells, geom_types, radii, grad_radius=grad_radius)
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    return lat, pen
