"""First-order periodic homogenisation of one lattice cell (reference: src/pyLatticeSim/homogenization_cell.py and
``get_homogenized_properties``, utils_simulation.py:83-119).

What the reference computes, restated on the condensed strut operator:

* periodic boundary conditions tie all six dofs of opposite corner / edge / face nodes (``periodic_boundary_condition``,
  homogenization_cell.py:210-252) and the translations of the vertex at the cell centre are fixed
  (``apply_dirichlet_for_homogenization`` :367-376);
* six unit macro strains ``w(x)`` (``find_imposed_strain`` :112-147; the shear cases are tensorial, i.e. gamma = 2);
  the load of case k is ``l(v) = -a(w_k, v)`` (``define_L_form`` :200-206, only the three translational strains of w
  are non-zero, so this is exactly ``-K w_k`` with zero nodal rotations), solved for the periodic fluctuation u_k with
  ONE factorisation for the six right-hand sides (``initialize_solver`` / ``solve_multiple_linear_problem``);
* macroscopic stress of ``u_tot = w_k + u_k`` from the reaction forces on the cell-boundary nodes,
  ``sum_i f_i (x) r_i`` (``calculate_macro_stress`` :309-331; no division by the cell volume, as in the reference),
  one column ``[s00, s11, s22, s10, s20, s21]`` of ``homogenizeMatrix`` per case; orthotropic constants from the
  un-symmetrised matrix, then ``homogenizeMatrix`` is symmetrised (``solve_full_homogenization`` :405-436).

The sub-meshed interior dofs of every strut are free and un-constrained in the reference, so condensing them out (the
2-node strut records of libpylattice_hip) changes nothing in the equations above.  Everything with arithmetic in it runs
on the device: the strut records, every K*w / K*u_tot, and the six constrained solves - the periodic constraints are a
master / slave row map handed to the library (``pl_set_periodic``), whose Jacobi PCG then runs on Q K Q with Q the
orthogonal projector "average over each periodic group" (= P^T K P v = -P^T K w, u = P v; the reference: dolfinx_mpc +
PETSc LU).  The host only averages the six right-hand sides over the groups and sums reactions.  (Until round 5 the reduced
matrix was factorised on the host with scipy; ``solver="host"`` keeps that path for cross-checks in the tests.)

Two deliberate differences, both only visible where the reference misbehaves: periodic partners are matched by
coordinates modulo the cell size (the reference pairs ``locate_dofs_topological`` results of equal boundary tags in
index order, right only for one node per tag), and reactions are summed over every boundary node (the reference takes
the first vertex of each tag).  Cells without a vertex at the centre (e.g. Octet) leave the reference's matrix with
three translational null vectors; here the first master node is anchored instead (macro stresses do not depend on
it).  A cell that is not connected (a mechanism) raises.
"""
from __future__ import annotations

import numpy as np

_VOIGT = ((0, 0), (1, 1), (2, 2), (1, 0), (2, 0), (2, 1))       # rows of homogenizeMatrix (:428-430)


def imposed_displacement(case: int, xyz: np.ndarray) -> np.ndarray:
    """Nodal field (N, 6) of the unit macro strain ``case`` in 1..6 (find_imposed_strain, :112-147); rotations zero."""
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    w = np.zeros((len(xyz), 6))
    if case == 1:
        w[:, 0] = x
    elif case == 2:
        w[:, 1] = y
    elif case == 3:
        w[:, 2] = z
    elif case == 4:
        w[:, 0], w[:, 1] = y, x
    elif case == 5:
        w[:, 0], w[:, 2] = z, x
    elif case == 6:
        w[:, 1], w[:, 2] = z, y
    else:
        raise ValueError("Invalid case number. Must be between 1 and 6.")
    return w


def periodic_masters(xyz: np.ndarray, lo: np.ndarray, size: np.ndarray, tol: float = 1e-9):
    """master[i] = node carrying the dofs of node i under cell periodicity (itself for interior nodes), and the mask
    of nodes on the cell boundary.  Opposite nodes = equal coordinates modulo the cell size; the master of a group is
    its member with the smallest (x, y, z), i.e. the one on the min faces - tags 1000 / 10x / 10..12 of the reference
    (lattice.py:612-614)."""
    rel = (xyz - lo) / size
    on_lo, on_hi = np.abs(rel) <= tol, np.abs(rel - 1.0) <= tol
    boundary = (on_lo | on_hi).any(axis=1)
    wrapped = np.where(on_hi, 0.0, rel)
    key = np.round(wrapped / (10 * tol)).astype(np.int64)
    master = np.arange(len(xyz))
    groups = {}
    order = np.lexsort((xyz[:, 2], xyz[:, 1], xyz[:, 0]))
    for i in order:
        if boundary[i]:
            master[i] = groups.setdefault(tuple(key[i]), i)
    return master, boundary


class HomogenizedCell:
    """Mirror of the reference's ``HomogenizedCell`` results interface: ``homogenizeMatrix``, ``orthotropicMatrix``,
    ``saveDataToExport`` (six total displacement fields, (N, 6) each), ``get_S_orthotropic`` and the print helpers."""

    def __init__(self, lattice, device=None, solver="device", rtol=1e-13):
        if lattice.get_number_cells() > 1:
            raise ValueError("The lattice must contain only one cell for homogenization.")
        if solver not in ("device", "host"):
            raise ValueError("solver must be 'device' or 'host'")
        self.lattice = lattice
        self.BeamModel = self
        self.solver = solver
        self.rtol = rtol
        self.pcg_iterations = []
        self.device = device if device is not None else lattice.device_model(precond=1)
        self.homogenizeMatrix = None
        self.orthotropicMatrix = None
        self.saveDataToExport = None
        self.generalizedStress = None
        self._symmetryError = None
        self._master = None
        self._boundary = None
        self._anchor = None
        self._solver = None

    # ---- boundary conditions ---------------------------------------------------------------------------------
    def prepare_simulation(self):
        """Strut records + explicit K on the device (SimulationBase.prepare_simulation builds the forms there)."""
        dev = self.device
        dev.set_bc(np.zeros((dev.n_nodes, 6), bool))
        dev.assemble()
        if self.solver == "host":
            dev.assemble_bsr(with_bc=False)

    def apply_dirichlet_for_homogenization(self):
        lat = self.lattice.lattice
        centre = lat.cell_coord[0] + 0.5 * lat.cell_size[0]
        d = np.abs(lat.node_xyz - centre).max(axis=1)
        i = int(np.argmin(d))
        self._anchor = i if d[i] <= 1e-6 else None

    def periodic_boundary_condition(self):
        lat = self.lattice.lattice
        self._master, self._boundary = periodic_masters(lat.node_xyz, lat.cell_coord[0], lat.cell_size[0])

    def find_boundary_tags(self):
        return np.flatnonzero(self._boundary)

    # ---- solver ----------------------------------------------------------------------------------------------
    def initialize_solver(self):
        """Reduce the device-assembled K to the periodic master dofs and factor it once (:256-276)."""
        if self._solver is not None:
            return
        import scipy.linalg
        import scipy.sparse as sp
        if self._master is None:
            self.periodic_boundary_condition()
        rowptr, col, vals = self.device.get_bsr()
        N = self.device.n_nodes
        K = sp.bsr_matrix((vals, col, rowptr), shape=(6 * N, 6 * N)).tocsr()
        masters = np.unique(self._master)
        slot = np.full(N, -1, np.int64)
        slot[masters] = np.arange(len(masters))
        node_slot = slot[self._master]
        cols = (6 * node_slot[:, None] + np.arange(6)).ravel()
        P = sp.csr_matrix((np.ones(6 * N), (np.arange(6 * N), cols)), shape=(6 * N, 6 * len(masters)))
        Kr = (P.T @ K @ P).toarray()
        # translation anchor: the centre vertex where the reference puts its Dirichlet condition, else the first
        # master node (macro stresses do not depend on it)
        anchor = int(node_slot[self._anchor]) if self._anchor is not None else 0
        fixed = 6 * anchor + np.arange(3)
        free = np.setdiff1d(np.arange(Kr.shape[0]), fixed)
        try:
            factor = scipy.linalg.cho_factor(Kr[np.ix_(free, free)])
        except np.linalg.LinAlgError as err:
            raise RuntimeError("homogenisation: the periodic cell operator is singular (mechanism in the cell?)") from err
        self._solver = (P, free, factor)

    def _group_average(self, v):
        """Q v: the average over every periodic group, on all its members (rows of an (N, 6) array)."""
        m = self._master
        cnt = np.bincount(m, minlength=len(m)).astype(float)
        acc = np.zeros_like(v)
        np.add.at(acc, m, v)
        return (acc / np.maximum(cnt, 1.0)[:, None])[m]

    def _anchor_group(self):
        """Nodes whose translations are fixed: the group of the centre vertex where the reference puts its Dirichlet
        condition, else of the first master node (macro stresses do not depend on it)."""
        a = self._master[self._anchor] if self._anchor is not None else int(np.unique(self._master)[0])
        return np.flatnonzero(self._master == a)

    def solve_multiple_linear_problem(self, w):
        """u (N, 6) periodic with  P^T K (w + u) = 0  and the anchor translations zero."""
        if self.solver == "device":
            if self._master is None:
                self.periodic_boundary_condition()
            dev = self.device
            if not getattr(self, "_periodic_set", False):
                dev.set_periodic(self._master)
                self._periodic_set = True
            fixed = np.zeros((dev.n_nodes, 6), bool)
            fixed[self._anchor_group(), :3] = True
            f = self._group_average(-dev.spmv(w))
            f[fixed] = 0.0
            dev.set_bc(fixed, None, f)
            u, st = dev.solve(rtol=self.rtol, max_iter=200000)
            self.pcg_iterations.append(int(st["iterations"]))
            return u
        import scipy.linalg
        self.initialize_solver()
        P, free, factor = self._solver
        b = -(P.T @ self.device.spmv(w).ravel())
        ur = np.zeros(P.shape[1])
        ur[free] = scipy.linalg.cho_solve(factor, b[free])
        return (P @ ur).reshape(-1, 6)

    def calculate_macro_stress(self, u_tot):
        """sum over boundary nodes of f (x) r with f = (K u_tot)[translations]  (:309-331)."""
        R = self.device.spmv(u_tot)
        self.generalizedStress = R
        b = self._boundary
        return R[b, :3].T @ self.lattice.lattice.node_xyz[b]

    def solve_full_homogenization(self):
        if self._master is None:
            self.periodic_boundary_condition()
        if self._anchor is None:
            self.apply_dirichlet_for_homogenization()
        xyz = self.lattice.lattice.node_xyz
        columns, self.saveDataToExport = [], []
        for case in range(1, 7):
            w = imposed_displacement(case, xyz)
            u_tot = w + self.solve_multiple_linear_problem(w)
            s = self.calculate_macro_stress(u_tot)
            columns.append(np.array([s[a][b] for a, b in _VOIGT]))
            self.saveDataToExport.append(u_tot)
        self.homogenizeMatrix = np.column_stack(columns)
        self.convert_to_orthotropic_form()
        self.compute_errors()
        self.homogenizeMatrix = 0.5 * (self.homogenizeMatrix + self.homogenizeMatrix.T)
        return self.homogenizeMatrix

    # ---- post-processing (:444-541) --------------------------------------------------------------------------
    def _engineering_constants(self):
        Hinv = np.linalg.inv(self.homogenizeMatrix)
        Ex, Ey, Ez = 1 / Hinv[0, 0], 1 / Hinv[1, 1], 1 / Hinv[2, 2]
        Gxy, Gxz, Gyz = 1 / (2 * Hinv[3, 3]), 1 / (2 * Hinv[4, 4]), 1 / (2 * Hinv[5, 5])
        nuxy, nuxz, nuyz = -Hinv[0, 1] * Ey, -Hinv[0, 2] * Ez, -Hinv[1, 2] * Ez
        return Ex, Ey, Ez, Gxy, Gxz, Gyz, nuxy, nuxz, nuyz

    def convert_to_orthotropic_form(self):
        Ex, Ey, Ez, Gxy, Gxz, Gyz, nuxy, nuxz, nuyz = self._engineering_constants()
        M = np.zeros_like(self.homogenizeMatrix)
        M[0, 0], M[1, 1], M[2, 2], M[3, 3], M[4, 4], M[5, 5] = Ex, Ey, Ez, Gxy, Gxz, Gyz
        M[0, 1] = M[1, 0] = nuxy
        M[0, 2] = M[2, 0] = nuxz
        M[1, 2] = M[2, 1] = nuyz
        self.orthotropicMatrix = M

    def get_S_orthotropic(self):
        Ex, Ey, Ez, Gxy, Gxz, Gyz, nuxy, nuxz, nuyz = self._engineering_constants()
        return np.array([[1 / Ex, -nuxy / Ex, -nuxz / Ex, 0.0, 0.0, 0.0],
                         [-nuxy / Ex, 1 / Ey, -nuyz / Ey, 0.0, 0.0, 0.0],
                         [-nuxz / Ex, -nuyz / Ey, 1 / Ez, 0.0, 0.0, 0.0],
                         [0.0, 0.0, 0.0, 1 / Gxy, 0.0, 0.0],
                         [0.0, 0.0, 0.0, 0.0, 1 / Gxz, 0.0],
                         [0.0, 0.0, 0.0, 0.0, 0.0, 1 / Gyz]])

    def compute_errors(self):
        Cm = self.homogenizeMatrix
        self._symmetryError = np.linalg.norm(0.5 * (Cm + Cm.T) - Cm) / np.linalg.norm(Cm)

    def print_homogenized_matrix(self):
        print("Homogenized matrix: ")
        for row in self.homogenizeMatrix:
            print(" ".join(f"{val:10.3f}" for val in row))

    def print_orthotropic_form(self):
        M = self.orthotropicMatrix
        for name, (i, j) in (("Ex", (0, 0)), ("Ey", (1, 1)), ("Ez", (2, 2)), ("nuxy", (0, 1)), ("nuxz", (0, 2)),
                             ("nuyz", (1, 2)), ("Gxy", (3, 3)), ("Gxz", (4, 4)), ("Gyz", (5, 5))):
            print(name + " ", M[i, j])

    def print_errors(self):
        print("Symmetry error: ", self._symmetryError)


def directional_modulus(matS: np.ndarray, theta: float, phi: float):
    """Directional stiffness E(theta, phi) * unit vector from a Voigt compliance matrix (pyLatticeSim/utils.py:35-73;
    Voigt index of (i, j), i != j, is 2 + i + j, shear compliances carry the factors 1/2 and 1/4)."""
    ct, st = np.cos(np.deg2rad(theta)), np.sin(np.deg2rad(theta))
    cp, sp = np.cos(np.deg2rad(phi)), np.sin(np.deg2rad(phi))
    u = np.array([st * cp, st * sp, ct])
    idx = np.array([[0, 3, 4], [3, 1, 5], [4, 5, 2]])
    coef = np.where(np.eye(3, dtype=bool), 1.0, 2.0)
    S4 = matS[idx[:, :, None, None], idx[None, None, :, :]] / (coef[:, :, None, None] * coef[None, None, :, :])
    inv_e = np.einsum("ijkl,i,j,k,l->", S4, u, u, u, u)
    return u / inv_e
