"""Drop-in for the reference's FEM driver on MI355X.

``solve_FEM_FenicsX(lattice)`` keeps the reference's signature and side effects
(src/pyLatticeSim/utils_simulation.py:21-56): every node's ``displacement_vector`` is set, constrained nodes get
their ``reaction_force_vector``, and ``xsol`` = ``lattice.get_global_displacement()[0]`` is returned together
with a model object.  The gmsh mesh + dolfinx assembly + PETSc LU of the reference are replaced by the condensed
per-strut operator and the Jacobi-PCG of libpylattice_hip (no CPU fallback).
"""
from __future__ import annotations

import numpy as np

from .timing import timing

DEFAULT_RTOL = 1e-9       # on ||r||/||b||; gives <1e-9 relative L2 error on displacements in the parity tests
DEFAULT_MAX_ITER = 200000


class FullScaleLatticeSimulation:
    """What the reference hands back as ``simulationModel`` (full_scale_lattice_simulation.py:28): exposes the
    solution field ``u`` (N,6) in lattice-node order, the lattice and the solver statistics."""

    def __init__(self, lattice, device):
        self.lattice = lattice
        self.BeamModel = self
        self.device = device
        self.u = None
        self.stats = None

    def apply_displacement_all_nodes_with_lattice_data(self):
        """full_scale_lattice_simulation.py:39-73."""
        self._fixed = self.lattice.fixed_DOF.copy()
        self._ubar = np.where(self._fixed, self.lattice.displacement_vector, 0.0)

    def apply_force_on_all_nodes_with_lattice_data(self):
        """full_scale_lattice_simulation.py:124-153 — only the three translational components reach the RHS."""
        f = np.zeros_like(self.lattice.applied_force)
        f[:, :3] = self.lattice.applied_force[:, :3]
        self._f = f

    def solve_problem(self, rtol=DEFAULT_RTOL, max_iter=DEFAULT_MAX_ITER):
        """simulation_base.py:465-514 (assemble with bcs, lifting, point loads, solve)."""
        dev = self.device
        dev.set_bc(self._fixed, self._ubar, self._f)
        dev.assemble()
        u, self.stats = dev.solve(rtol=rtol, max_iter=max_iter)
        # dolfinx adds point loads to the RHS AFTER set_bc, so a load on a constrained dof shows up in that dof's
        # value (identity row): u_c = ubar_c + f_c (simulation_base.py:494-498).  Kept for parity.
        self.u = np.where(self._fixed, self._ubar + self._f, u)
        self._u_solver = u
        return self.u

    def set_result_diplacement_on_lattice_object(self):
        """full_scale_lattice_simulation.py:77-107: displacements of every lattice node; the penalisation points
        (node_mod, FE vertices in the reference) are interior points of the condensed struts here and are recovered in
        closed form on the device (pl_node_mod) the first time one of them is looked at."""
        self.lattice.displacement_vector[:] = self.u
        if getattr(self.lattice, "_compat_rows", False):
            return                                   # reference_compat: the penalisation points are rows of u already
        if self.lattice.is_penalized:
            dev, u, lat = self.device, self._u_solver, self.lattice
            lat._node_mod_pending = lambda: dev.node_mod(u)
            lat._node_mod_stale = False

            def _device_changes(why, lat=lat):
                # the back-substitution reads the device's CURRENT records: once radii / segments change or the handle
                # closes, a pending evaluation would combine the old u with new records (round-2 advisor finding)
                if getattr(lat, "_node_mod_pending", None) is not None:
                    lat._node_mod_pending = None
                    lat._node_mod_stale = why
            dev._before_change = _device_changes

    @property
    def domain(self):
        """Stand-in for the dolfinx mesh the reference's model carries (``simulationModel.domain``): vertex
        coordinates (lattice nodes, then penalisation points) and the line cells (segments) between them."""
        from types import SimpleNamespace
        from .views import _tables
        t = _tables(self.lattice)
        return SimpleNamespace(geometry=SimpleNamespace(x=t.node_xyz, dim=3),
                               topology=SimpleNamespace(dim=1, cells=t.beam_conn, radius=t.beam_radius,
                                                        beam_mod=t.beam_mod))

    def set_reaction_force_on_lattice_with_FEM_results(self):
        """full_scale_lattice_simulation.py:111-120: R_i = v_i^T K u for the 6 dofs of every constrained node."""
        R = self.device.reactions(self.u) + self._loaded_constraint_correction()
        nodes = self._fixed.any(axis=1)
        # the reference loops ``for cell: for node in cell.points_cell`` and Point.set_reaction_force ACCUMULATES
        # (point.py:368-380), so a node shared by k cells ends up with k times its reaction.  Kept for parity.
        mult = np.bincount(self.lattice.cell_points()[1], minlength=len(self._fixed)).astype(float)
        self.lattice.reaction_force_vector[nodes] += mult[nodes, None] * R[nodes]
        self.reactions = R

    def _loaded_constraint_correction(self):
        """A point load on a CONSTRAINED dof (dolfinx adds loads after the Dirichlet rows: u_c = ubar_c + f_c,
        simulation_base.py:494-498) enters the reference's reactions R = K u through the FIRST SUB-ELEMENT of every strut at
        that node only - the sub-nodes behind it were solved with u_c = ubar_c - whereas the condensed strut here answers the
        displaced end as a whole.  For the (rare) dofs concerned: remove the condensed struts' answer to f_c and put the first
        sub-element's in (rows of the node itself, and of the segment's other end when the segment is one sub-element)."""
        from .compat_device import _segment_stiffness
        from .views import _tables
        sim, fixed, f = self.lattice, self._fixed, self._f
        hit = fixed & (f != 0)
        out = np.zeros_like(self.u)
        if not hit.any():
            return out
        delta = np.where(hit, f, 0.0)
        out -= self.device.spmv(delta)                                    # what the condensed struts answered to f_c (K delta)
        lat, pen, t = sim.lattice, sim.penalized, _tables(sim)
        mult = sim.beam_mult if getattr(sim, "beam_mult", None) is not None else np.ones(lat.n_beams)
        nodes = np.flatnonzero(hit.any(axis=1))
        rows_n = len(self.u)
        for end in (0, 1):
            struts = np.flatnonzero(np.isin(lat.beam_conn[:, end], nodes))
            if not len(struts):
                continue
            # first segment seen from this end: the penalised one if present, else the middle one
            k = np.where(pen.seg_len[struts, 2 * end] > 0, 2 * end, 1)
            L, n = pen.seg_len[struts, k], pen.seg_nsub[struts, k]
            rad = lat.beam_radius[struts] * np.where(k == 1, 1.0, sim.penalization_coefficient)
            a, b = lat.node_xyz[lat.beam_conn[struts, 0]], lat.node_xyz[lat.beam_conn[struts, 1]]
            tdir = (b - a) / np.linalg.norm(b - a, axis=1)[:, None] * (1.0 if end == 0 else -1.0)   # away from this end
            Ke = _segment_stiffness(L / n, np.ones(len(struts)), rad, tdir * (L / n)[:, None], sim.young_modulus,
                                    sim.poisson_ratio)
            i = lat.beam_conn[struts, end]
            d = delta[i]                                                   # (S, 6)
            np.add.at(out, i, mult[struts, None] * np.einsum("sij,sj->si", Ke[:, :6, :6], d))
            # one sub-element: its other end is a row of the model (the penalisation point, or the strut's other node)
            one = n == 1
            if one.any():
                other = np.where(k[one] == 1, lat.beam_conn[struts[one], 1 - end], t.pen_id[struts[one], end])
                # (a middle segment that ends on the far penalisation point)
                far_pen = t.pen_id[struts[one], 1 - end]
                other = np.where((k[one] == 1) & (far_pen >= 0), far_pen, other)
                ok = other < rows_n                                        # default model: penalisation points are no rows
                np.add.at(out, other[ok], (mult[struts[one], None] *
                                           np.einsum("sij,sj->si", Ke[one][:, 6:, :6], d[one]))[ok])
        return out

    def calculate_reaction_force_and_moment_at_position(self, position, tol: float = 1e-8):
        d = np.abs(self.lattice.node_coordinates() - np.asarray(position, float)).max(axis=1)
        i = int(np.argmin(d))
        if d[i] > tol:
            raise RuntimeError(f"No DOF found near point {position} with tol={tol}.")
        return list(self.device.reactions(self.u)[i])


@timing.category("simulation")
@timing.timeit
def solve_FEM_FenicsX(lattice, rtol=None, max_iter=DEFAULT_MAX_ITER):
    """Solve the lattice's FEM problem on the GPU; returns (xsol, simulationModel).  ``rtol`` (on ||r|| / ||b|| of the
    PCG; the reference solves directly) defaults to ``lattice.fem_rtol`` if set, else DEFAULT_RTOL."""
    if rtol is None:
        rtol = getattr(lattice, "fem_rtol", None) or DEFAULT_RTOL
    model = FullScaleLatticeSimulation(lattice, lattice.device_model())
    model.apply_displacement_all_nodes_with_lattice_data()
    model.apply_force_on_all_nodes_with_lattice_data()
    model.solve_problem(rtol=rtol, max_iter=max_iter)
    model.set_result_diplacement_on_lattice_object()
    model.set_reaction_force_on_lattice_with_FEM_results()
    xsol, _ = lattice.get_global_displacement()
    return xsol, model


def get_homogenized_properties(lattice):
    """Periodic homogenisation of a one-cell lattice (utils_simulation.py:83-119): returns the orthotropic compliance
    matrix and the analysis object (``homogenizeMatrix``, ``orthotropicMatrix``, ``saveDataToExport`` ...)."""
    from .homogenization_cell import HomogenizedCell
    if lattice.get_number_cells() > 1:
        raise ValueError("The lattice must contain only one cell for homogenization.")
    analysis = HomogenizedCell(lattice)
    analysis.prepare_simulation()
    analysis.apply_dirichlet_for_homogenization()
    analysis.periodic_boundary_condition()
    analysis.solve_full_homogenization()
    analysis.print_homogenized_matrix()
    analysis.print_errors()
    return analysis.get_S_orthotropic(), analysis
