"""Device model of ``LatticeSim(reference_compat=True)``: the reference's own FE model on lattices whose struts are shared
by several cells, on top of the same HIP handle.

What the reference builds (lattice_sim.py:250-303,405-458; lattice_generation.py:104-175):

* a strut with a penalised end that belongs to k cells is replaced by k copies of its segments (one per owner cell;
  ``Beam`` hashes by identity, beam.py:78-82, and the mesher de-duplicates by object) - k parallel chains between the same
  points.  On the device that is ONE strut whose record is k times stiffer (``pl_set_multiplicity``);
* penalisation points lying in a cell face get boundary indices, Dirichlet values and their share of a surface load
  (the total is divided by the number of loaded rows INCLUDING those points).  A penalisation point that carries boundary
  data cannot stay an interior point of a condensed strut: it is PROMOTED to a node of the device mesh and its strut is
  cut there into one-segment struts (``pl_mesh_t`` takes arbitrary [pen | middle | pen] lengths, including zero);
  all other penalisation points stay condensed and are recovered in closed form (``pl_node_mod``).

The wrapper speaks the rows of the reference's node list (design nodes, then every penalisation point, in ``Point.index``
order) and the design struts, so the callers (``utils_simulation``, ``LatticeOpti``) do not see the cut.

What cannot be pinned here: gmsh is absent, so how the reference's gmsh model treats the copies is taken from its intent
(k lines between the same two points, each meshed on its own).  What the reference literally hands to gmsh is worse than
that: the copies made by the second and later owner cells end on ``Point`` objects that never receive an index
(``Lattice.nodes`` is a set of points hashing by coordinates, so only the first copy of a point is indexed,
lattice.py:687-696), ``latticeGeneration.generate_nodes`` files them all under ``self.point[None]``
(lattice_generation.py:116-121) and every such line end is wired to that ONE gmsh point - recorded by
tests/golden/make_golden.py (``gmsh_input_*.npz``: 144 of the 864 lines of the 2x2x2 Octet end on one point, 48 are
degenerate).  That accident is not reproduced; the coordinates of the copies' own end points are.
"""
from __future__ import annotations

import numpy as np


class CompatDevice:
    """Same methods as ``_capi.HipLattice`` for the calls the host layer makes, on the reference's rows."""

    def __init__(self, sim, **kw):
        from .views import _tables
        self._sim_kw = kw
        lat, pen, t = sim.lattice, sim.penalized, _tables(sim)
        self._lat, self._pen, self._t = lat, pen, t
        self._E, self._nu, self._pen_coef = sim.young_modulus, sim.poisson_ratio, sim.penalization_coefficient
        self.N = lat.n_nodes
        self.n_nodes = t.n_nodes                      # rows: design nodes + every penalisation point
        self.n_beams = lat.n_beams
        self._mult = None if sim.beam_mult is None else np.asarray(sim.beam_mult, dtype=np.float64)
        self._radius = np.asarray(lat.beam_radius, dtype=np.float64).copy()
        self._promoted = np.zeros(self.n_nodes - self.N, bool)
        self._dev = None
        self._bc = None
        self._before_change = None
        self.last_stats = None

    # -- the cut mesh ----------------------------------------------------------------------------------------------
    def _build(self):
        lat, pen, t, N = self._lat, self._pen, self._t, self.N
        if self._dev is not None:
            self._dev._before_change = None
            self._dev.close()
        prom = self._promoted
        self._prom_rows = np.flatnonzero(prom)                       # pen-table positions of the promoted points
        dev_of_pen = np.full(len(prom), -1, np.int64)
        dev_of_pen[self._prom_rows] = N + np.arange(len(self._prom_rows))
        B = lat.n_beams
        pid = t.pen_id                                               # (B, 2) row of the pen point, -1 if absent
        cut = np.zeros((B, 2), bool)
        has = pid >= 0
        cut[has] = prom[pid[has] - N]
        q = np.where(cut, dev_of_pen[np.where(has, pid - N, 0)], -1)  # device node of a promoted pen point
        a, b = lat.beam_conn[:, 0].astype(np.int64), lat.beam_conn[:, 1].astype(np.int64)
        sl, sn = pen.seg_len, pen.seg_nsub
        z, zi = np.zeros(B), np.zeros(B, np.int32)
        c0, c1 = cut[:, 0], cut[:, 1]
        # piece 0 always exists: from point1 to the first cut (or to point2)
        end0 = np.where(c0, q[:, 0], np.where(c1, q[:, 1], b))
        len0 = np.stack([sl[:, 0], np.where(c0, z, sl[:, 1]), np.where(c0 | c1, z, sl[:, 2])], axis=1)
        sub0 = np.stack([sn[:, 0], np.where(c0, zi, sn[:, 1]), np.where(c0 | c1, zi, sn[:, 2])], axis=1)
        # piece 1: behind the cut at q1 (middle [+ pen2 if q2 is not cut]); or, without a cut at q1, pen2 behind q2
        p1 = np.flatnonzero(c0 | c1)
        beg1 = np.where(c0, q[:, 0], q[:, 1])[p1]
        end1 = np.where(c0 & c1, q[:, 1], b)[p1]
        len1 = np.stack([z, np.where(c0, sl[:, 1], z), np.where(c0 & c1, z, sl[:, 2])], axis=1)[p1]
        sub1 = np.stack([zi, np.where(c0, sn[:, 1], zi), np.where(c0 & c1, zi, sn[:, 2])], axis=1)[p1]
        # piece 2: pen2 alone, when both points are cut
        p2 = np.flatnonzero(c0 & c1)
        conn = np.concatenate([np.c_[a, end0], np.c_[beg1, end1], np.c_[q[p2, 1], b[p2]]]).astype(np.int32)
        seg_len = np.concatenate([len0, len1, np.stack([z[p2], z[p2], sl[p2, 2]], axis=1)])
        seg_nsub = np.concatenate([sub0, sub1, np.stack([zi[p2], zi[p2], sn[p2, 2]], axis=1)]).astype(np.int32)
        self._parent = np.concatenate([np.arange(B), p1, p2])
        # piece that holds the (condensed) junction of a NON-promoted penalisation point: pen1 -> piece 0 (its q1),
        # pen2 -> the last piece of the strut (its q2)
        last = np.arange(B)
        last[p1] = B + np.arange(len(p1))
        last[p2] = B + len(p1) + np.arange(len(p2))
        self._last_piece = last
        xyz = np.concatenate([lat.node_xyz, t.node_xyz[N + self._prom_rows]])
        from ._capi import HipLattice
        kw = dict(self._sim_kw)
        mult = None if self._mult is None else self._mult[self._parent]
        if mult is not None and not (mult != 1.0).any():
            mult = None
        self._dev = HipLattice(xyz, conn, self._radius[self._parent], seg_len, seg_nsub, self._E, self._nu,
                               pen_coef=self._pen_coef, beam_mult=mult, **kw)
        self._dev._before_change = self._forward_change
        self._assembled = False

    def _forward_change(self, why):
        cb = self._before_change
        if cb is not None:
            self._before_change = None
            cb(why)

    def _to_dev(self, rows):
        rows = np.asarray(rows).reshape(self.n_nodes, 6)
        return np.concatenate([rows[:self.N], rows[self.N + self._prom_rows]])

    def _from_dev(self, rows_dev, fill=0.0):
        out = np.full((self.n_nodes, 6), fill, dtype=rows_dev.dtype)
        out[:self.N] = rows_dev[:self.N]
        out[self.N + self._prom_rows] = rows_dev[self.N:]
        return out

    @property
    def device(self):
        """The HIP handle of the cut mesh (built at the first set_bc)."""
        if self._dev is None:
            self._build()
        return self._dev

    # -- HipLattice interface on the reference's rows ------------------------------------------------------------
    def set_bc(self, fixed, ubar=None, f=None):
        R, N = self.n_nodes, self.N
        fx = np.asarray(fixed).reshape(R, 6) != 0
        ub = np.zeros((R, 6)) if ubar is None else np.asarray(ubar, dtype=float).reshape(R, 6)
        ff = np.zeros((R, 6)) if f is None else np.asarray(f, dtype=float).reshape(R, 6)
        need = fx[N:].any(axis=1) | (ff[N:] != 0).any(axis=1)
        if self._dev is None or (need & ~self._promoted).any():
            self._promoted |= need                       # grows only: an adjoint load must not drop the equilibrium's cuts
            self._build()
        self._bc = (fx, ub, ff)
        self._dev.set_bc(self._to_dev(fx), self._to_dev(ub), self._to_dev(ff))

    def assemble(self):
        self.device.assemble()
        self._assembled = True

    def update_radii(self, radius):
        self._radius = np.asarray(radius, dtype=np.float64).reshape(self.n_beams).copy()
        if self._dev is not None:
            self._dev.update_radii(self._radius[self._parent])

    def _full_field(self, u_dev):
        """Rows of every reference node from the device solution: promoted points are unknowns of the cut mesh, the others
        are back-substituted inside their (piece of a) strut."""
        u = self._from_dev(u_dev)
        t, N = self._t, self.N
        rest = np.flatnonzero(~self._promoted)
        if len(rest):
            nm = self._dev.node_mod(u_dev)                                  # (pieces, 2, 6)
            s, e = t.pen_strut[rest], t.pen_end[rest]
            piece = np.where(e == 0, s, self._last_piece[s])
            u[N + rest] = nm[piece, e]
        return u

    def solve(self, rtol=1e-8, max_iter=20000, raise_on_noconv=True, download=True):
        if not download:
            raise ValueError("the reference-compatible model returns the field of every reference node")
        u_dev, st = self.device.solve(rtol=rtol, max_iter=max_iter, raise_on_noconv=raise_on_noconv)
        self.last_stats = st
        return self._full_field(u_dev), st

    def reactions(self, u):
        return self._from_dev(self.device.reactions(self._to_dev(u)))

    def spmv(self, x):
        return self._from_dev(self.device.spmv(self._to_dev(x)))

    def energy(self, u):
        return self.device.energy(self._to_dev(u))

    def sens(self, u, lam=None):
        s = self.device.sens(self._to_dev(u), None if lam is None else self._to_dev(lam))
        return np.bincount(self._parent, weights=s, minlength=self.n_beams)

    def node_mod(self, u):
        """(B, 2, 6) rows of the two penalisation points of every design strut (an absent one: the end node's rows)."""
        u = np.asarray(u, dtype=float).reshape(self.n_nodes, 6)
        pid, conn = self._t.pen_id, self._lat.beam_conn
        return u[np.where(pid >= 0, pid, conn)]

    def time_kernel(self, which, reps=20):
        return self.device.time_kernel(which, reps)

    def algorithmic_bytes(self):
        return self.device.algorithmic_bytes()

    def close(self):
        if self._dev is not None:
            self._forward_change("handle closed")
            self._dev._before_change = None
            self._dev.close()
            self._dev = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
