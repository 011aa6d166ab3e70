"""Host-side mirror of the reference's ``LatticeOpti`` for the sensitivity pass of the hot path.

Mirrors ``src/pyLatticeOpti/lattice_opti.py``: preset block ``optimization_informations`` (:228-256), the
parameterisations ``constant`` / ``constant+hybrid`` / ``unit_cell`` / ``linear`` (:467-560, :1468-1485),
normalisation (:1318-1400), ``objective`` (:430-465), ``compute_compliance`` (:645-663), ``gradient`` (:701-731) with
its sign flip and ``(r_max − r_min)/C0`` scaling, and the SciPy SLSQP driver (:141-195, stays SciPy).

What runs on the GPU: the equilibrium solve (``solve_FEM_FenicsX``), the adjoint solve for displacement objectives and
the per-strut contraction ``λ_eᵀ (∂K_e/∂r) u_e`` (``pl_sens``) — the reference obtains the same quantity as
``u_cellᵀ (∂S/∂r) u_cell`` with a finite-differenced cell Schur complement (``lattice_sim.py:1020-1054``), at fixed
penalised-segment geometry (convention (i) of SURVEY.md appendix A).

``simulation_type: "DDM"`` (what most of the reference's optimisation presets use): the equilibrium goes through
``solve_DDM`` on the device with surrogate cell Schur complements, the gradient contracts their spline derivatives
(``schur_surrogate.py``) cell by cell.

Out of scope here: the kriging relative-density surrogate; the density constraint uses the
direct strut-volume formula instead.
"""
from __future__ import annotations

import time

import numpy as np

from .lattice_sim import LatticeSim, open_lattice_parameters
from .utils_simulation import solve_FEM_FenicsX

_DOF = {"X": 0, "Y": 1, "Z": 2, "RX": 3, "RY": 4, "RZ": 5}


class LatticeOpti(LatticeSim):
    def __init__(self, name_file, mesh_trimmer=None, verbose: int = 0, convergence_plotting: bool = False,
                 data_roots=None, reference_compat=None):
        """``reference_compat`` (one switch with LatticeSim's): the reference's behaviour where it differs from the consistent
        one - its model of struts shared by several cells (lattice_sim.py docstring here), and for the "linear"
        parameterisation its gradient exactly as it computes it (lattice_opti.py:787-841, 719-720) instead of the
        derivative of the objective (see calculate_gradient)."""
        info = open_lattice_parameters(name_file).get("optimization_informations", {})
        # lattice_opti.py:96-103: simulation_type "DDM" runs every equilibrium through solve_DDM with the cell Schur
        # complements (exact or surrogate) and contracts their derivatives dS/dr for the gradient
        self._ddm_mode = info.get("simulation_type", None) == "DDM"
        super().__init__(name_file, mesh_trimmer, verbose, self._ddm_mode, data_roots=data_roots,
                         reference_compat=reference_compat)
        self.solution = None
        self.actual_objective = None
        self.denorm_objective = None
        self.initial_value_objective = None
        self.actualGradient = None
        self.initial_parameters = None
        self.bounds = None
        self.constraints = []
        self.iteration = 0
        self.optim_ftol = 1e-6
        self.optim_disp = True
        self.optim_eps = 1e-3
        self.actual_optimization_parameters = []
        self._sim_is_current = False
        self.min_radius = 0.01
        self.max_radius = 0.1
        self._history = {"iteration": [], "objective_norm": [], "objective": [], "relative_density": [],
                         "parameters": [], "timestamp": []}
        lat = self.lattice
        self._get_optimization_parameters(name_file)
        self._set_number_parameters_optimization()
        # strut -> (cell, type) of the LAST cell that holds it: Cell.change_beam_radius (cell.py:896-917) is called
        # cell after cell, so a strut shared by several cells ends with the last one's radius
        cell_of = np.repeat(np.arange(lat.n_cells), np.diff(lat.cell_beam_ptr))
        self._beam_cell = np.zeros(lat.n_beams, np.int64)
        self._beam_cell[lat.cell_beam_idx] = cell_of          # later cells overwrite earlier ones
        gr = self.grad_radius if self.grad_radius is not None else np.ones((max(self.num_cells_x, self.num_cells_y,
                                                                                self.num_cells_z), 3))
        pos = lat.cell_pos
        self._cell_gfac = gr[pos[:, 0], 0] * gr[pos[:, 1], 1] * gr[pos[:, 2], 2]     # Cell.get_radius (cell.py:385-412)
        self._cell_center = lat.cell_coord + 0.5 * lat.cell_size
        self._x0 = self._y0 = self._z0 = 0.0

    # -- preset ----------------------------------------------------------------------------------------
    def _get_optimization_parameters(self, name_file):
        info = open_lattice_parameters(name_file).get("optimization_informations", {})
        self.objective_function = info.get("objective_function", None)
        self.objective_type = info.get("objective_type", None)
        self.objectif_data = info.get("objective_data", None)
        self.optim_max_iteration = info.get("max_iterations", 20)
        self.constraints_dict = info.get("constraints", {})
        self.optimization_parameters = info.get("optimization_parameters", None)
        if self.optimization_parameters is None:
            raise ValueError("No optimization parameters defined.")
        self._simulation_type = info.get("simulation_type", None)
        if self._simulation_type not in {"FEM", "DDM"}:
            raise ValueError("Invalid simulation type for optimization. Choose 'FEM' or 'DDM'.")
        self.enable_normalization = info.get("enable_parameter_normalization", False)
        self.enable_gradient_computing = info.get("enable_gradient_computing", False)

    def _set_number_parameters_optimization(self):
        t = self.optimization_parameters["type"]
        if t == "unit_cell":
            self.number_parameters = self.lattice.n_cells * len(self.geom_types)
        elif t == "linear":
            dirs = self.optimization_parameters.get("direction", [])
            if not dirs:
                raise ValueError("No directions provided for linear optimization.")
            if any(d not in {"x", "y", "z"} for d in dirs):
                raise ValueError(f"Invalid direction in {dirs}; valid are 'x', 'y', 'z'.")
            self.number_parameters = len(dirs) + 1
        elif t == "constant":
            self.number_parameters = len(self.geom_types) if self.optimization_parameters.get("hybrid", False) else 1
        else:
            raise ValueError("Invalid optimization parameters type.")

    def redefine_optim_parameters(self, max_iteration=None, ftol=None, disp=None, eps=None):
        if max_iteration is not None:
            self.optim_max_iteration = max_iteration
        if ftol is not None:
            self.optim_ftol = ftol
        if disp is not None:
            self.optim_disp = disp
        if eps is not None:
            self.optim_eps = eps

    # -- normalisation (lattice_opti.py:1318-1400) ---------------------------------------------------------
    def _clamp_radius(self, v):
        return max(self.min_radius, min(self.max_radius, float(v)))

    def denormalize_optimization_parameters(self, r_norm):
        if not self.enable_normalization:
            return list(r_norm)
        r = np.asarray(r_norm, dtype=float) * (self.max_radius - self.min_radius) + self.min_radius
        return np.minimum(self.max_radius, np.maximum(self.min_radius, r)).tolist()       # (_clamp_radius, vectorised)

    def normalize_optimization_parameters(self, r):
        if not self.enable_normalization:
            return list(r)
        out = []
        for v in r:
            if v < self.min_radius or v > self.max_radius:
                raise ValueError("Optimization parameter out of bounds.")
            out.append((v - self.min_radius) / (self.max_radius - self.min_radius))
        return out

    def normalize_objective(self, value):
        if not self.enable_normalization:
            return float(value)
        if self.initial_value_objective is None:
            s = abs(float(value))
            self.initial_value_objective = s if s != 0.0 else 1.0
        return float(value) / self.initial_value_objective

    def _to_normalized_theta_space(self, grad_dr):
        if not self.enable_normalization:
            return grad_dr
        if self.initial_value_objective in (None, 0.0):
            raise RuntimeError("Normalization scale not initialized; call objective() once before grad.")
        return grad_dr * (self.max_radius - self.min_radius) / self.initial_value_objective

    # -- parameters -> radii --------------------------------------------------------------------------------
    def _cell_radii_from_parameters(self, theta):
        """(C, G) base radii per cell and geometry + d r_cell / d theta as a sparse description."""
        t = self.optimization_parameters["type"]
        C, G = self.lattice.n_cells, len(self.geom_types)
        theta = list(map(float, theta))
        if t == "constant":
            if self.optimization_parameters.get("hybrid", False):
                if len(theta) != G:
                    raise ValueError(f"Expected {G} parameters for hybrid constant mode, got {len(theta)}.")
                per = self.denormalize_optimization_parameters(theta)
                return np.tile(np.asarray(per), (C, 1))
            r = self.denormalize_optimization_parameters([theta[0]])[0]
            return np.full((C, G), r)
        if t == "unit_cell":
            return np.asarray(self.denormalize_optimization_parameters(theta)).reshape(C, G)
        if t == "linear":
            dirs = self.optimization_parameters.get("direction", [])
            d_phys = self.denormalize_optimization_parameters([theta[-1]])[0]
            span = self.max_radius - self.min_radius
            L = np.maximum([self.size_x, self.size_y, self.size_z], 1e-16)
            h = (self._cell_center - [self._x0, self._y0, self._z0]) / L
            s = np.zeros(C)
            for i, dkey in enumerate(dirs):
                s += theta[i] * h[:, "xyz".index(dkey)]
            val = np.clip(d_phys + span * s, self.min_radius, self.max_radius)
            return np.tile(val[:, None], (1, G))
        raise ValueError("Invalid optimization parameters type.")

    def set_optimization_parameters(self, optimization_parameters_actual):
        """lattice_opti.py:467-560: map the optimiser's vector to strut radii (device update, topology untouched)."""
        # (this runs once per design variable and SLSQP iteration - the finite-difference density constraint - so it stays in
        #  numpy: the list conversions and np.allclose of the first version were 10 ms per iteration at 192 variables)
        th = np.array(optimization_parameters_actual, dtype=float).ravel()
        if len(th) != self.number_parameters:
            raise ValueError("Invalid number of optimization parameters.")
        prev = getattr(self, "_theta_arr", None)
        if (self.actual_optimization_parameters is not None and prev is not None
                and len(self.actual_optimization_parameters) == len(th) == len(prev)
                and bool(np.all(np.abs(th - prev) <= 1e-10 + 1e-10 * np.abs(prev)))):      # (np.allclose, rtol = atol = 1e-10)
            return
        self._sim_is_current = False
        theta = th.tolist()
        self._theta_arr = th
        self.actual_optimization_parameters = theta
        self.cell_radii = self._cell_radii_from_parameters(theta)
        lat = self.lattice
        lat.beam_radius = (self.cell_radii[self._beam_cell, lat.beam_type] * self._cell_gfac[self._beam_cell])
        if self._ddm_mode:
            # reset_cell_with_new_radii (lattice_sim.py:1421-1497): new cell radii -> new Schur complements (+ dS/dr)
            lat.cell_radii = self.cell_radii * self._cell_gfac[:, None]
            # exact: one device condensation per distinct radius set (+ central differences for dS/dr when gradients
            # are on, as lattice_sim.py:1020-1054 does with dolfinx solves); surrogates: one batched evaluation - made when
            # the matrices are next asked for (_flush_schur): SLSQP's finite-difference density constraint sets one vector per
            # design variable and iteration that is never simulated (round 5, 12 x 4 x 4 cells: 197 evaluations of all cell
            # matrices per SLSQP iteration, 146 of its 186 ms)
            self._schur_stale = True
            return
        # the device gets the new radii when it is next asked for (device_model): SLSQP differentiates the density constraint by
        # finite differences - 54 parameter vectors per iteration on the reference's 6x3x3 preset, none of which is simulated
        # (round 5: 1 174 of 1 253 calls of this method in a 20-iteration run uploaded radii nobody used)
        self._device_radii_stale = True

    def _flush_schur(self):
        """Cell matrices (and dS/dr) for the radii set last: evaluated here, not in set_optimization_parameters."""
        if getattr(self, "_schur_stale", False):
            self._schur_stale = False
            self.calculate_schur_complement_cells()

    def ddm_model(self):
        self._flush_schur()
        return super().ddm_model()

    def device_model(self, **kw):
        """As LatticeSim.device_model; a design loop solves ONE slowly changing system over and over, so its handle starts every
        solve from the best combination of its last six solutions for the system at hand (pl_opts_t.warm_start = 4, the Galerkin
        start; round 5 - until then the previous solution, and for compliance only: an objective with an adjoint solve alternates
        two right-hand sides on the handle, which a projection does not mind - the states and the adjoints of the last design
        iterations are all candidates).  Radii set since the last call are uploaded here."""
        if self._device is None:
            kw.setdefault("warm_start", 4)
        fresh = self._device is None
        dev = super().device_model(**kw)
        if getattr(self, "_device_radii_stale", False) and not self._ddm_mode:
            if not fresh:                      # (a handle created just now was built from the current radii)
                dev.update_radii(self.lattice.beam_radius)
            self._device_radii_stale = False
        return dev

    # -- equilibrium / objective ------------------------------------------------------------------------------
    def _initialize_simulation_parameters(self):
        self.reaction_force_vector[:] = 0.0
        self.displacement_vector[:] = 0.0
        self.fixed_DOF[:] = False
        self.applied_force[:] = 0.0
        self.set_boundary_conditions()

    def _simulate_lattice_equilibrium(self):
        self._initialize_simulation_parameters()
        if self._ddm_mode:
            xsol, info, _, _ = self.solve_DDM()
            if xsol is None:
                raise RuntimeError("solve_DDM: zero right-hand side")
        else:
            _, self._model = solve_FEM_FenicsX(self)
        self._sim_is_current = True

    def compute_compliance(self):
        """C = sum_k f_k u_k over loaded dofs (lattice_opti.py:645-663)."""
        return float((self.applied_force * self.displacement_vector).sum())

    def _objective_nodes(self, surfaces):
        return self.find_point_on_lattice_surface(surfaces)

    def calculate_objective(self):
        if self.objective_type == "compliance":
            return self.compute_compliance()
        if self.objective_type == "displacement":
            nodes = self._objective_nodes(self.objectif_data["Surface"])
            vals = [self.displacement_vector[n, _DOF[d]] for n in nodes for d in self.objectif_data["DOF"]]
            mean_disp = float(np.mean(vals))
            if self.objective_function == "max":
                return -mean_disp
            if self.objective_function == "min":
                return mean_disp
            raise ValueError("objective_function must be 'min' or 'max'")
        if self.objective_type == "displacement_ratio":
            u_in, u_out, _ = self._ratio_terms()
            return -(u_out * u_in)                 # lattice_opti.py:616-636 (inverse mechanism: u_out ~ -u_in)
        if self.objective_type == "stiffness":
            raise NotImplementedError("Stiffness objective not implemented yet.")     # as the reference (:638)
        raise ValueError("Invalid objective function type.")

    def _ratio_terms(self):
        """displacement_ratio (lattice_opti.py:616-636, 1560-1621): mean displacement of the loaded dofs (the boundary
        condition named "Load", under "Force" if there is a force block, else under "Displacement"), mean displacement of
        the objective's dofs, and q = dJ/du of J = -(u_out u_in) on the nodes.  For one DOF per set - what the reference's
        presets use - these are the reference's own coefficients (-u_in / n_out on the output nodes, -u_out / n_in on the
        loaded ones); with several DOFs the reference divides by the node count only, which is not the derivative of its
        mean over nodes x DOFs: here q is the derivative."""
        bd = self.boundary_conditions
        if bd.get("Force", None) is not None:
            bd = bd["Force"]
        elif bd.get("Displacement", None) is not None:
            bd = bd["Displacement"]
        else:
            raise ValueError("No boundary conditions defined for displacement ratio objective.")
        if "Load" not in bd:
            raise ValueError('displacement_ratio needs a boundary condition named "Load" (lattice_opti.py:625)')
        nodes_in = np.asarray(self._objective_nodes(bd["Load"]["Surface"]), dtype=np.int64)
        nodes_out = np.asarray(self._objective_nodes(self.objectif_data["Surface"]), dtype=np.int64)
        c_in = [_DOF[d] for d in bd["Load"]["DOF"]]
        c_out = [_DOF[d] for d in self.objectif_data["DOF"]]
        u = self.displacement_vector
        u_in = float(np.mean(u[np.ix_(nodes_in, c_in)]))
        u_out = float(np.mean(u[np.ix_(nodes_out, c_out)]))
        q = np.zeros_like(u)
        for k in c_out:
            np.add.at(q, (nodes_out, k), -u_in / (len(nodes_out) * len(c_out)))
        for k in c_in:
            np.add.at(q, (nodes_in, k), -u_out / (len(nodes_in) * len(c_in)))
        return u_in, u_out, q

    def objective(self, r):
        self.set_optimization_parameters(r)
        if not self._sim_is_current:
            self._simulate_lattice_equilibrium()
        objective = self.calculate_objective()
        self.denorm_objective = objective
        val = self.normalize_objective(objective)
        if self.objective_function == "max":
            val = -val
        elif self.objective_function != "min":
            raise ValueError("objective_function must be 'min' or 'max'")
        self.actual_objective = val
        return val

    # -- sensitivities ------------------------------------------------------------------------------------------
    def _adjoint(self, q):
        """Solve K lam = q on the free dofs (lam = 0 on constrained ones) with the device PCG."""
        dev = self.device_model()
        fixed = self._model._fixed
        dev.set_bc(fixed, None, np.where(fixed, 0.0, q))
        lam, _ = dev.solve(rtol=1e-10, max_iter=200000)
        dev.set_bc(fixed, self._model._ubar, self._model._f)     # restore the equilibrium problem
        return lam

    def strut_sensitivities(self):
        """s_b such that d(objective)/d r_b = −s_b at fixed segment geometry (compliance: s_b = u_eᵀ ∂K_e/∂r u_e)."""
        dev = self.device_model()
        u = self._model.u
        if self.objective_type == "compliance":
            lam = u
            if np.any(self._model._ubar != 0.0):      # prescribed displacements: the adjoint is not u itself
                lam = self._adjoint(self._model._f)
            return dev.sens(u, None if lam is u else lam)
        if self.objective_type == "displacement":
            nodes = self._objective_nodes(self.objectif_data["Surface"])
            q = np.zeros_like(u)
            cnt = len(nodes) * len(self.objectif_data["DOF"])
            sign = -1.0 if self.objective_function == "max" else 1.0
            for d in self.objectif_data["DOF"]:
                q[nodes, _DOF[d]] += sign / cnt
            return dev.sens(u, self._adjoint(q))
        if self.objective_type == "displacement_ratio":
            return dev.sens(u, self._adjoint(self._ratio_terms()[2]))
        raise NotImplementedError(f"Gradient for objective '{self.objective_type}' not implemented yet.")

    def _ddm_adjoint(self, q_nodes):
        """S lam = q on the free cell-boundary dofs (lattice_opti.py:1560-1650: CG to 1e-10, no preconditioner)."""
        dev = self.ddm_model()
        bn = self._boundary_nodes_by_index()
        fixed = self.fixed_DOF[bn]
        dev.set_bc(fixed, None, np.where(fixed, 0.0, q_nodes[bn]))
        if getattr(self, "_ddm_precond", 0) in (2, 3, 4):
            # pl_set_bc drops what depends on the Dirichlet mask: the factorised assembled-Schur preconditioner (2) and the
            # inverted node blocks (3; with their dense level: 4) - the handle is then "not assembled" and pl_solve refuses
            dev.assemble()
        lam_b, _ = dev.solve(rtol=1e-10, max_iter=max(2000, self.number_iteration_max or 0))
        lam = np.zeros_like(self.displacement_vector)
        lam[bn] = lam_b
        return lam

    def _ddm_cell_sensitivities(self):
        """(C, G):  lam_c^T (dS_c/dr_j) u_c per cell and geometry, u_c / lam_c on the cell's boundary nodes in
        Cell.define_node_order_to_simulate order (lattice_opti.py:746-760, 866-890)."""
        self._flush_schur()
        if self.schur_gradients is None:
            raise RuntimeError("Schur complement gradients are not available: enable_gradient_computing must be true")
        cb = self.cell_boundary_nodes()
        U = self.displacement_vector[cb].reshape(len(cb), -1)
        if self.objective_type == "compliance":
            Lam = U
        elif self.objective_type == "displacement":
            nodes = self._objective_nodes(self.objectif_data["Surface"])
            q = np.zeros_like(self.displacement_vector)
            cnt = len(nodes) * len(self.objectif_data["DOF"])
            sign = -1.0 if self.objective_function == "max" else 1.0
            for d in self.objectif_data["DOF"]:
                q[nodes, _DOF[d]] += sign / cnt
            Lam = self._ddm_adjoint(q)[cb].reshape(len(cb), -1)
        elif self.objective_type == "displacement_ratio":
            Lam = self._ddm_adjoint(self._ratio_terms()[2])[cb].reshape(len(cb), -1)
        else:
            raise NotImplementedError(f"Gradient for objective '{self.objective_type}' not implemented yet.")
        G = len(self.geom_types)
        GA = getattr(self, "_schur_gradients_array", None)
        if GA is not None and GA.shape[1] == G:
            # every cell at once: t = dS[idx[c], j] u_c (batched matrix-vector products), then lam_c . t  (a design with one
            # radius set per cell has as many distinct matrices as cells: the loop below made 4 096 einsum calls at 16^3 cells)
            idx = self.cell_schur_index
            T = np.matmul(GA[idx], U[:, None, :, None])[..., 0]          # (C, G, n)
            s_cell = np.einsum("cgn,cn->cg", T, Lam)
            return s_cell * self._cell_gfac[:, None]
        s_cell = np.zeros((len(cb), G))
        for k, dS_list in enumerate(self.schur_gradients):            # one batched contraction per distinct matrix
            sel = np.flatnonzero(self.cell_schur_index == k)
            for j, dS in enumerate(dS_list):
                s_cell[sel, j] = np.einsum("ci,ij,cj->c", Lam[sel], dS, U[sel])
        return s_cell * self._cell_gfac[:, None]

    def calculate_gradient(self):
        """Raw gradient in the reference's sign convention (lattice_opti.py:735-907): sum over the struts driven by
        each parameter of λ_eᵀ (∂K_e/∂r) u_e (FEM) or over the cells of λ_cᵀ (∂S_c/∂r) u_c (DDM), chained through
        Cell.get_radius and the parameterisation."""
        lat = self.lattice
        G = len(self.geom_types)
        if self._ddm_mode:
            s_cell = self._ddm_cell_sensitivities()
        else:
            s = self.strut_sensitivities()
            s_cell = np.zeros((lat.n_cells, G))
            np.add.at(s_cell, (self._beam_cell, lat.beam_type), s * self._cell_gfac[self._beam_cell])
        t = self.optimization_parameters["type"]
        if t == "unit_cell":
            return s_cell.ravel()
        if t == "constant":
            if self.optimization_parameters.get("hybrid", False):
                return s_cell.sum(axis=0)
            return np.array([s_cell.sum()])
        if t == "linear" and self.reference_compat:
            # The reference's chain rule, verbatim in effect (lattice_opti.py:787-841): slopes are de-normalised like
            # radii (clamped into [r_min, r_max]), the "unclamped radius" r = sum a_k c_k + d is formed with the ABSOLUTE
            # cell-centre coordinates and only decides which cells count; d r / d a_k = c_k, d r / d d = 1.  This is not
            # the derivative of the objective (whose field is r = d + span * sum theta_k (c_k - c0_k) / L_k, :467-560);
            # kept behind this switch so that runs can be compared with the reference number for number.
            dirs = self.optimization_parameters.get("direction", [])
            theta = self.actual_optimization_parameters
            a = [self.denormalize_optimization_parameters([float(theta[i])])[0] for i in range(len(dirs))]
            d0 = self.denormalize_optimization_parameters([float(theta[-1])])[0]
            cen = self._cell_center
            r_un = d0 + sum(a[i] * cen[:, "xyz".index(k)] for i, k in enumerate(dirs))
            active = (r_un > self.min_radius + 1e-12) & (r_un < self.max_radius - 1e-12)
            sc = (s_cell / self._cell_gfac[:, None]).sum(axis=1) * active
            grad = np.zeros(self.number_parameters)
            for i, k in enumerate(dirs):
                grad[i] = (sc * cen[:, "xyz".index(k)]).sum()
            grad[-1] = sc.sum()
            return grad
        if t == "linear":
            dirs = self.optimization_parameters.get("direction", [])
            theta = self.actual_optimization_parameters
            span = self.max_radius - self.min_radius
            L = np.maximum([self.size_x, self.size_y, self.size_z], 1e-16)
            h = (self._cell_center - [self._x0, self._y0, self._z0]) / L
            d_phys = self.denormalize_optimization_parameters([theta[-1]])[0]
            r_un = d_phys + span * sum(theta[i] * h[:, "xyz".index(k)] for i, k in enumerate(dirs))
            active = (r_un > self.min_radius + 1e-12) & (r_un < self.max_radius - 1e-12)
            sc = s_cell.sum(axis=1) * active
            grad = np.zeros(self.number_parameters)
            for i, k in enumerate(dirs):
                grad[i] = (sc * span * h[:, "xyz".index(k)]).sum()
            grad[-1] = sc.sum()
            return grad
        raise NotImplementedError(f"Gradient for optimization type '{t}' not implemented yet.")

    def gradient(self, r):
        """d(normalised objective)/d(theta) (lattice_opti.py:701-731)."""
        self.set_optimization_parameters(r)
        if not self._sim_is_current:
            self._simulate_lattice_equilibrium()
        g = -self.calculate_gradient()
        if self.objective_function == "max":
            g = -g        # objective() negates the (normalised) value for 'max'; keep the pair consistent
        if self.optimization_parameters["type"] == "linear" and not self.reference_compat:
            # slopes act on the physical radius directly (span already applied); only the intercept is normalised
            scale = np.ones(self.number_parameters)
            if self.enable_normalization:
                scale[:] = 1.0 / self.initial_value_objective
                scale[-1] *= (self.max_radius - self.min_radius)
            g = g * scale
        else:
            g = self._to_normalized_theta_space(g)
        self.actualGradient = g.copy()
        return g

    # -- density constraint (direct strut-volume formula) ---------------------------------------------------------
    def relative_density(self):
        lat = self.lattice
        geo = getattr(self, "_density_geometry", None)
        if geo is None or geo[0] is not lat.beam_conn:          # strut lengths and the cells' volume: fixed by the topology
            d = lat.node_xyz[lat.beam_conn[:, 1]] - lat.node_xyz[lat.beam_conn[:, 0]]
            geo = self._density_geometry = (lat.beam_conn, np.pi * np.linalg.norm(d, axis=1),
                                            float(lat.cell_size.prod(axis=1).sum()))
        r = lat.beam_radius
        return float((geo[1] * r * r).sum() / geo[2])

    def density_constraint(self, r):
        self.set_optimization_parameters(r)
        return self.relative_density() - float(self.constraints_dict["relative_density"]["value"])

    # -- driver (SciPy SLSQP, as the reference) ---------------------------------------------------------------------
    def _initialize_optimization_solver(self):
        from scipy.optimize import Bounds
        lo, hi = (0.0, 1.0) if self.enable_normalization else (self.min_radius, self.max_radius)
        t = self.optimization_parameters["type"]
        init = float(np.mean(self.normalize_optimization_parameters(self.radii)))
        if t == "linear":
            self.bounds = Bounds(lb=[-1.0] * (self.number_parameters - 1) + [lo],
                                 ub=[1.0] * (self.number_parameters - 1) + [hi])
            self.initial_parameters = [0.0] * (self.number_parameters - 1) + [init]
        else:
            self.bounds = Bounds(lb=[lo] * self.number_parameters, ub=[hi] * self.number_parameters)
            if t == "constant" and self.optimization_parameters.get("hybrid", False):
                self.initial_parameters = self.normalize_optimization_parameters(self.radii)
            else:
                self.initial_parameters = [init] * self.number_parameters

    def callback_function(self, r):
        self._opt_iteration = getattr(self, "_opt_iteration", 0) + 1   # (solve_DDM keeps its CG count in .iteration)
        self.iteration = self._opt_iteration
        self._history["iteration"].append(self.iteration)
        self._history["objective_norm"].append(self.actual_objective)
        self._history["objective"].append(self.denorm_objective)
        self._history["relative_density"].append(self.relative_density())
        self._history["parameters"].append(list(map(float, r)))
        self._history["timestamp"].append(time.time())

    def optimize_lattice(self):
        from scipy.optimize import NonlinearConstraint, minimize
        self.initial_value_objective = None
        self.iteration = 0
        self._opt_iteration = 0
        self._sim_is_current = False
        self.actual_optimization_parameters = []
        self._initialize_optimization_solver()
        self.constraints = []
        if "relative_density" in self.constraints_dict:
            mode = self.constraints_dict["relative_density"].get("mode", "upper")
            lb, ub = {"upper": (-np.inf, 0.0), "lower": (0.0, np.inf), "eq": (0.0, 0.0)}.get(mode, (-np.inf, 0.0))
            self.constraints.append(NonlinearConstraint(self.density_constraint, lb, ub))
        kw = dict(fun=self.objective, x0=self.initial_parameters, method="SLSQP", bounds=self.bounds,
                  constraints=self.constraints, callback=self.callback_function,
                  options={"maxiter": self.optim_max_iteration, "ftol": self.optim_ftol, "disp": self.optim_disp,
                           "eps": self.optim_eps})
        if self.enable_gradient_computing:
            kw["jac"] = self.gradient
        self.solution = minimize(**kw)
        self.set_optimization_parameters(self.solution.x)
        return self.solution
