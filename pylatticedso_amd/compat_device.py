"""Device model of ``LatticeSim(reference_compat=True)``: the reference's own FE model on lattices whose struts are shared
by several cells, on top of the same HIP handle.

What the reference builds (lattice_sim.py:250-303,405-458; lattice_generation.py:104-175):

* a strut with a penalised end that belongs to k cells is replaced by k copies of its segments (one per owner cell;
  ``Beam`` hashes by identity, beam.py:78-82, and the mesher de-duplicates by object) - k parallel chains between the same
  points.  On the device that is ONE strut whose record is k times stiffer (``pl_set_multiplicity``);
* penalisation points lying in a cell face get boundary indices, Dirichlet values and their share of a surface load
  (the total is divided by the number of loaded rows INCLUDING those points).  Three cases:
  - a point that only carries a LOAD stays an interior point of its condensed strut: static condensation of a loaded
    interior node is exact - the load reaches the strut's ends as ``f_E = -K_EI K_II^-1 f_I`` (host, closed-form segment
    stiffnesses, a 6 x 6 or 12 x 12 solve per loaded strut) and the point's own displacement is the back-substitution
    of ``pl_node_mod`` plus ``K_II^-1 f_I / copies``;
  - a point whose whole strut is clamped (both ends and every point fixed to zero in all six dofs - the in-face struts of
    a clamped face) needs nothing: the strut moves nothing and carries nothing;
  - any other point with a Dirichlet dof is PROMOTED to a node of the device mesh and its strut is cut there into
    one-segment struts (``pl_mesh_t`` takes arbitrary [pen | middle | pen] lengths, including zero).
  A cantilever therefore runs on the design mesh with multiplicities (same tiles, same preconditioner, same iteration
  counts as the default model) - the first version promoted every point with boundary data and paid 321 instead of 120
  iterations at 50^3 Octet for the short stiff pieces that made.

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


def _segment_stiffness(length, nsub, radius, dvec, E, nu, kappa=0.9):
    """(P, 12, 12) stiffness of P straight segments [end A | end B] in global axes: the exact condensation of a chain of
    ``nsub`` P1 x P1 Timoshenko sub-elements with mid-point shear integration (pl_device.h) - tip
    flexibility L/(ES), L/(GJ), [[L/(kGS) + L^3/(3EI) (1 - 1/(4 n^2)), L^2/(2EI)], [L^2/(2EI), L/(EI)]], inverted, and
    carried to the other end by rigid transport.  Host arithmetic for the handful of struts with a loaded interior point."""
    L, n, R = np.asarray(length, float), np.asarray(nsub, float), np.asarray(radius, float)
    G = E / (2.0 * (1.0 + nu))
    S, I = np.pi * R ** 2, 0.25 * np.pi * R ** 4
    ka, kt = E * S / L, G * 2.0 * I / L
    f11 = L / (kappa * G * S) + L ** 3 / (3.0 * E * I) * (1.0 - 1.0 / (4.0 * n * n))
    f12, f22 = L ** 2 / (2.0 * E * I), L / (E * I)
    det = f11 * f22 - f12 * f12
    a, b, c = f22 / det, f12 / det, f11 / det
    d = np.asarray(dvec, float)
    L2 = (d * d).sum(axis=1)

    def tip_blocks(a, c, e1, e2, e3, d):
        P = len(a)
        eye = np.eye(3)[None]
        dd = d[:, :, None] * d[:, None, :]
        D = np.zeros((P, 3, 3))
        D[:, 0, 1], D[:, 0, 2], D[:, 1, 0] = -d[:, 2], d[:, 1], d[:, 2]
        D[:, 1, 2], D[:, 2, 0], D[:, 2, 1] = -d[:, 0], -d[:, 1], d[:, 0]
        Kss = np.zeros((P, 6, 6))
        Kss[:, :3, :3] = a[:, None, None] * eye + e1[:, None, None] * dd
        Kss[:, 3:, 3:] = c[:, None, None] * eye + e3[:, None, None] * dd
        Kss[:, :3, 3:] = e2[:, None, None] * D
        Kss[:, 3:, :3] = -e2[:, None, None] * D
        Rm = np.tile(np.eye(6), (P, 1, 1))
        Rm[:, :3, 3:] = -D
        return Kss, -Kss @ Rm

    e1, e2, e3 = (ka - a) / L2, b / np.sqrt(L2), (kt - c) / L2
    Kbb, Kba = tip_blocks(a, c, e1, e2, e3, d)
    g = a - 2.0 * e2
    Kaa, _ = tip_blocks(a, c + L2 * g, e1, a - e2, e3 - g, -d)
    K = np.zeros((len(L), 12, 12))
    K[:, :6, :6], K[:, 6:, 6:], K[:, 6:, :6] = Kaa, Kbb, Kba
    K[:, :6, 6:] = np.swapaxes(Kba, 1, 2)
    return K


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
        self._piece_conn, self._piece_len, self._piece_sub = conn.astype(np.int64), seg_len, seg_nsub.astype(np.int64)
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
        t, lat = self._t, self._lat
        # struts that are clamped as a whole (both ends and every penalisation point fixed to zero in all six dofs): they
        # move nothing and carry nothing, their points need no node of their own
        # (a load on a constrained dof counts as movement: dolfinx adds point loads after the Dirichlet rows are set, so that
        # dof's value is ubar + f - utils_simulation.solve_problem keeps that for parity)
        still = fx.all(axis=1) & ~(np.where(fx, ub, 0.0) != 0).any(axis=1) & ~(ff[:, :3] != 0).any(axis=1)
        pid = t.pen_id
        strut_still = still[lat.beam_conn[:, 0]] & still[lat.beam_conn[:, 1]]
        for e in (0, 1):
            strut_still &= np.where(pid[:, e] >= 0, still[np.maximum(pid[:, e], 0)], True)
        pen_rows_strut = t.pen_strut
        need = fx[N:].any(axis=1) & ~strut_still[pen_rows_strut]
        if self._dev is None or (need & ~self._promoted).any():
            self._promoted |= need                       # grows only: an adjoint load must not drop the equilibrium's cuts
            self._build()
        self._bc = (fx, ub, ff)
        f_dev = self._to_dev(ff)
        self._feq_dev = None
        self._particular = None
        loaded = np.flatnonzero((ff[N:, :3] != 0).any(axis=1) & ~self._promoted)
        if len(loaded):
            f_dev = f_dev + self._condense_point_loads(loaded, ff[N + loaded])
        self._dev.set_bc(self._to_dev(fx), self._to_dev(ub), f_dev)

    def _condense_point_loads(self, rows, f_rows):
        """Static condensation of loaded interior points (penalisation points that stay inside their condensed strut):
        returns the (device rows, 6) equivalent loads on the ends of their pieces, f_E = -K_EI K_II^-1 f_I, and keeps
        u_I^p = K_II^-1 f_I / copies for the back-substitution.  A piece [seg0 | seg1 | seg2] has the junctions J1 (behind
        seg0) and J2 (before seg2); only forces are loads (full_scale_lattice_simulation.py:144)."""
        t, N = self._t, self.N
        s, e = t.pen_strut[rows], t.pen_end[rows]
        piece = np.where(e == 0, s, self._last_piece[s])
        up, inv = np.unique(piece, return_inverse=True)
        P = len(up)
        conn, sl, sn = self._piece_conn[up], self._piece_len[up], self._piece_sub[up]
        xyz = self._dev.node_xyz
        dvec = xyz[conn[:, 1]] - xyz[conn[:, 0]]
        tvec = dvec / np.sqrt((dvec * dvec).sum(axis=1))[:, None]
        r = self._radius[self._parent[up]]
        rad = np.stack([self._pen_coef * r, r, self._pen_coef * r], axis=1)
        # chain nodes: 0 = end A, 1 = J1, 2 = J2, 3 = end B; an absent end segment merges its junction with the end node
        have = sl > 0
        node_of = np.zeros((P, 4), np.int64)
        node_of[:, 1] = np.where(have[:, 0], 1, 0)                      # J1 = A when there is no seg0
        node_of[:, 3] = 3
        node_of[:, 2] = np.where(have[:, 2], 2, 3)                      # J2 = B when there is no seg2
        # (pieces without a middle segment are the single-segment pieces of a cut strut: they have no interior junction)
        assert have[:, 1].all(), "a loaded penalisation point inside a piece without a middle segment"
        K = np.zeros((P, 24, 24))
        ends = [(0, 1), (1, 2), (2, 3)]
        for k, (na, nb_) in enumerate(ends):
            sel = np.flatnonzero(have[:, k])
            if not len(sel):
                continue
            Ks = _segment_stiffness(sl[sel, k], sn[sel, k], rad[sel, k], tvec[sel] * sl[sel, k][:, None], self._E, self._nu)
            ia = (6 * node_of[sel, na][:, None] + np.arange(6)[None, :])
            ib = (6 * node_of[sel, nb_][:, None] + np.arange(6)[None, :])
            idx = np.concatenate([ia, ib], axis=1)                      # (n, 12) chain dofs of the segment's two ends
            np.add.at(K, (sel[:, None, None], idx[:, :, None], idx[:, None, :]), Ks)
        # interior dofs = junctions 1 and 2 where they are nodes of their own; unused chain nodes get a unit diagonal
        own = np.zeros((P, 4), bool)
        for j in (1, 2):
            own[:, j] = node_of[:, j] == j
        fI = np.zeros((P, 24))
        jn = np.where(e == 0, 1, 2)                                     # the loaded point is J1 (end 0) or J2 (end 1)
        assert own[inv, jn].all(), "a loaded penalisation point must be an interior junction of its piece"
        np.add.at(fI, (inv[:, None], 6 * jn[:, None] + np.arange(3)[None, :]), f_rows[:, :3])
        interior = np.repeat(own, 6, axis=1)
        interior[:, :6] = False
        interior[:, 18:] = False
        unused = ~interior
        unused[:, :6] = False
        unused[:, 18:] = False
        Kii = K.copy()
        d24 = np.arange(24)
        # keep only interior-interior coupling; identity elsewhere
        mask = interior[:, :, None] & interior[:, None, :]
        Kii = np.where(mask, Kii, 0.0)
        Kii[:, d24, d24] = np.where(interior, Kii[:, d24, d24], 1.0)
        uI = np.linalg.solve(Kii, np.where(interior, fI, 0.0)[:, :, None])[:, :, 0]
        fE = -(K @ np.where(interior, uI, 0.0)[:, :, None])[:, :, 0]     # rows of the end dofs are what is wanted
        out = np.zeros((self._dev.n_nodes, 6))
        np.add.at(out, conn[:, 0], fE[:, :6])
        np.add.at(out, conn[:, 1], fE[:, 18:])
        self._feq_dev = out
        mult = np.ones(P) if self._mult is None else self._mult[self._parent[up]]
        self._particular = (rows, (uI.reshape(P, 4, 6)[inv, jn]) / mult[inv][:, None])
        return out

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
        if self._particular is not None:                     # loaded interior points: + K_II^-1 f_I of one copy
            rows, up = self._particular
            u[N + rows] += up
        return u

    def solve(self, rtol=1e-8, max_iter=20000, raise_on_noconv=True, download=True):
        if not download:
            raise ValueError("the reference-compatible model returns the field of every reference node")
        u_dev, st = self.device.solve(rtol=rtol, max_iter=max_iter, raise_on_noconv=raise_on_noconv)
        self.last_stats = st
        return self._full_field(u_dev), st

    def reactions(self, u):
        R = self.device.reactions(self._to_dev(u))
        if self._feq_dev is not None:        # K u of the sub-meshed model at an end of a loaded strut: S u_E - f_E
            R = R - self._feq_dev
        return self._from_dev(R)

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
