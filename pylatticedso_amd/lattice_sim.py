"""Host-side mirror of the reference's ``LatticeSim`` for the FEM hot path, backed by arrays.

Mirrors (same names, argument meaning and error behaviour) the parts of
``src/pyLatticeSim/lattice_sim.py`` and ``src/pyLatticeDesign/lattice.py`` that feed
``solve_FEM_FenicsX``: preset parsing (lattice.py:212-311, lattice_sim.py:201-238), joint penalisation
(lattice_sim.py:245-308), boundary conditions (lattice_sim.py:405-494, lattice.py:1320-1411) and the
``xsol`` packing (lattice_sim.py:502-542).  Node data live in (N,6) arrays instead of ``Point`` lists:

    fixed_DOF, displacement_vector, applied_force, reaction_force_vector, index_boundary

Struts shared by several cells (lattices with struts in cell faces or on cell edges: Octet, Cubic, Kelvin, Auxetic ...):
the reference splits such a strut once PER OWNER CELL (lattice_sim.py:250-303) and keeps every copy, gives the
penalisation points lying in a cell face a boundary index, boundary conditions and a place in ``xsol``
(lattice_sim.py:405-458,502-563).  Two models are offered:

* ``reference_compat=False`` (default): every strut is penalised exactly once and only design nodes carry boundary
  data - the physical lattice;
* ``reference_compat=True`` (or environment ``PYLATTICE_REFERENCE_COMPAT=1``): the reference's own state - strut
  multiplicity = number of owner cells, every per-node array has one row per REFERENCE node (design nodes, then
  penalisation points, in the reference's index order), boundary data and ``xsol`` entries on penalisation points.
  Bit-exact against the states dumped from the running reference for all 29 golden lattices; the solve treats the
  copies of a strut as parallel chains between the same points (see ``compat_device.py`` for the one thing that cannot
  be pinned here: what the reference's gmsh model makes of them).
"""
from __future__ import annotations

import json
import os
from pathlib import Path

import numpy as np

from . import lattice_arrays as LA
from .timing import timing
from .views import LatticeViews

DDM_DENSE_MAX = 16384      # PL_DDM_DENSE_MAX of include/pylattice_hip.h
DDM_LARGE_PRECOND = 4       # above it: 4 = node blocks + dense level on node aggregates (3 = the node blocks alone)
DDM_COARSE_MAX_DOFS = 0     # size of that dense level (pl_opts_t.coarse_max_dofs; 0 = the library's default, 1 536)
# device_model(): lattice sizes from which the handle is asked for the multi-level preconditioner + record palette, and for
# fp32-stored PCG vectors with fp64 refinement (measured crossovers, DESIGN.md sections 7 / 7a)
MULTILEVEL_MIN_NODES = 600
DIRECT_MAX_NODES = 200        # up to here: dense factorisation of K as the preconditioner (opts.precond = 5), see device_model
SMALL_TILES_MAX_NODES = 5000  # multi-level PCG below this size: 32-node tiles and 12-mode dense level
FP32_VECTORS_MIN_NODES = 2_000_000

_ROOT = Path(__file__).resolve().parents[1]
PRESET_DIR = _ROOT / "data" / "inputs" / "preset_lattice"

# E, nu of the reference's material cards (src/pyLatticeDesign/materials/*.json)
MATERIALS = {"VeroClear": (1013.0, 0.3), "TPU": (20000.0, 0.3), "Ti-6Al-4V": (104000.0, 0.35)}
_SURFACES = ["Xmin", "Xmax", "Ymin", "Ymax", "Zmin", "Zmax", "Xmid", "Ymid", "Zmid"]
_DOF_MAP = {"X": 0, "Y": 1, "Z": 2, "RX": 3, "RY": 4, "RZ": 5}


def open_lattice_parameters(file_name):
    """utils.open_lattice_parameters (utils.py:111-130): preset name relative to data/inputs/preset_lattice,
    or an absolute path; '.json' appended when missing.  A dict is passed through (extension)."""
    if isinstance(file_name, dict):
        return file_name
    json_path = PRESET_DIR / file_name
    if json_path.suffix != ".json":
        json_path = json_path.with_suffix(".json")
    try:
        with open(json_path, "r") as fh:
            return json.load(fh)
    except FileNotFoundError:
        raise FileNotFoundError(f"The file {json_path} does not exist.")


def material_properties(name):
    if name in MATERIALS:
        return MATERIALS[name]
    p = os.environ.get("PYLATTICE_MATERIAL_PATH")
    if p and os.path.exists(os.path.join(p, f"{name}.json")):
        with open(os.path.join(p, f"{name}.json")) as fh:
            d = json.load(fh)
        return float(d["Young_modulus"]), float(d["Poisson_ratio"])
    raise FileNotFoundError(f"Material file not found: {name}.json")


class LatticeSim(LatticeViews):
    _design_mult_note_done = False      # the "strut copies" warning is given once per process

    def __init__(self, name_file, mesh_trimmer=None, verbose: int = 0,
                 enable_domain_decomposition_solver: bool = False, data_roots=None, reference_compat=None):
        """Same arguments as the reference (lattice_sim.py:44-47) plus ``data_roots``: extra directories in which the
        reduced-basis files of the surrogate DDM modes are looked up (the reference finds them in its own checkout), and
        ``reference_compat`` (module docstring; None = environment PYLATTICE_REFERENCE_COMPAT, else False)."""
        if mesh_trimmer is not None:
            raise NotImplementedError("mesh_trimmer is outside the accelerated path")
        if reference_compat is None:
            reference_compat = os.environ.get("PYLATTICE_REFERENCE_COMPAT", "0") not in ("", "0")
        self.reference_compat = bool(reference_compat)
        self._verbose = verbose
        self.data_roots = list(data_roots or [])
        self.used_schur_preconditioner = None
        self._ddm_precond = 0
        self.domain_decomposition_solver = enable_domain_decomposition_solver
        self.n_DOF_per_node = 6
        self.penalization_coefficient = LA.PENALIZATION_COEFFICIENT
        self.is_penalized = False
        params = open_lattice_parameters(name_file)
        self._extract_geometry(params)
        self.define_simulation_parameters(params)
        self._base_radii = [float(r) for r in self.radii]
        self._cell_radii_override = None
        self._device = self._ddm_device = None
        self._generate_and_prepare()
        self.cell_schur_index = None       # (C,) index into self.schur_complements
        self.schur_complements = None      # (n_S, 6 n_b, 6 n_b)
        self.iteration = 0
        self.enable_gradient_computing = False     # lattice_sim.py:114: also keep dS/dr per unique cell
        self.schur_gradients = None                # [n_S] lists of dS/dr_j
        self.schur_surrogate = None
        if self.domain_decomposition_solver:
            if self.type_schur_complement_computation == "FE2":
                raise NotImplementedError("FE2 Schur complements are outside the accelerated path")
            if self.type_schur_complement_computation != "exact":
                # reduced basis + surrogate for its coefficients (lattice_sim.py:126-135); the .npz is looked up
                # like the reference does, below <root>/data/outputs/schur_complement/reduced_basis/
                from .schur_surrogate import SchurSurrogate
                self.schur_surrogate = SchurSurrogate.load(self.geom_types, self.precision_greedy,
                                                           self.type_schur_complement_computation,
                                                           search_dirs=data_roots)
                self.reduce_basis_dict = {"basis_reduced_ortho": self.schur_surrogate.basis,
                                          "alpha_ortho": self.schur_surrogate.alpha_train.T,
                                          "list_elements": self.schur_surrogate.points}
                self.alpha_coefficients_greedy = self.schur_surrogate.alpha_train
                self.shape_schur_complement = self.schur_surrogate.n
            if self.enable_preconditioner and self.preconditioner_type not in ("mean", "nearest_reference", "exact"):
                raise NotImplementedError("Not implemented preconditioner approximation method.")   # lattice_sim.py:1323
            self.calculate_schur_complement_cells()

    # ------------------------------------------------------------------------------------------------
    def _extract_geometry(self, p):
        geometry = p.get("geometry", {})
        cs, nc = geometry.get("cell_size", {}), geometry.get("number_of_cells", {})
        self.cell_size_x, self.cell_size_y, self.cell_size_z = cs.get("x"), cs.get("y"), cs.get("z")
        self.num_cells_x, self.num_cells_y, self.num_cells_z = nc.get("x"), nc.get("y"), nc.get("z")
        self.radii = geometry.get("radii")
        self.geom_types = geometry.get("geom_types")
        if None in [self.cell_size_x, self.cell_size_y, self.cell_size_z, self.num_cells_x, self.num_cells_y,
                    self.num_cells_z, self.radii, self.geom_types]:
            raise ValueError("Missing geometry parameters in JSON file.")
        # lattice.py:236-238
        self.enable_randomness = bool(geometry.get("enable_randomness", False))
        self.range_radius = geometry.get("range_radius", [0.01, 0.1])
        self.randomness_hybrid = bool(geometry.get("randomness_hybrid", False))
        grad = p.get("gradient", {})

        def table(block, keys):
            if not block:
                return None
            return LA.gradient_table(self.num_cells_x, self.num_cells_y, self.num_cells_z,
                                     block.get("rule", "constant"),
                                     [block.get(f"direction_{a}", False) for a in "xyz"],
                                     [block.get(f"parameter_{a}", 0.0) for a in "xyz"])

        self.grad_radius = table(grad.get("radii", {}), None)
        self.grad_dim = table(grad.get("cell_dimension", {}), None)
        sup = p.get("supplementary", {})
        blocks = []
        for blk in sup.get("erased_blocks", {}).values():
            s, d = blk.get("start_point", {}), blk.get("dimensions_block", {})
            blocks.append([s.get("x", 0.0), s.get("y", 0.0), s.get("z", 0.0), d.get("x", 0.0), d.get("y", 0.0),
                           d.get("z", 0.0)])
        self.eraser_blocks = blocks or None
        self.symmetry_lattice = None                    # lattice.py:294-303
        sym = sup.get("symmetries", {})
        if sym:
            pt = sym.get("reference_point", {})
            self.symmetry_lattice = {"sym_plane": sym.get("plane", None),
                                     "sym_point": (pt.get("x", 0.0), pt.get("y", 0.0), pt.get("z", 0.0))}

    def define_simulation_parameters(self, name_file):
        """lattice_sim.py:201-238."""
        p = open_lattice_parameters(name_file)
        sim = p.get("simulation_parameters", {})
        self.enable_simulation_properties = bool(sim.get("enable", False))
        self.material_name = sim.get("material", "VeroClear")
        self.enable_periodicity = sim.get("periodicity", False)
        ddm = sim.get("DDM", None)
        self.enable_preconditioner = None
        self.preconditioner_type = None
        self.number_iteration_max = None
        self.type_schur_complement_computation = None
        self.precision_greedy = None
        if ddm is not None:
            self.enable_preconditioner = ddm.get("enable_preconditioner", False)
            self.preconditioner_type = ddm.get("preconditioner_type", None)
            if self.preconditioner_type is None and self.enable_preconditioner:
                raise ValueError("Preconditioner type must be defined in the input file.")
            self.number_iteration_max = ddm.get("max_iterations", 1000)
            comp = ddm.get("schur_complement_computation", None)
            if comp is None:
                raise ValueError("Schur complement computation method must be defined in the input file.")
            self.type_schur_complement_computation = comp.get("type", None)
            if self.type_schur_complement_computation not in ["exact", "FE2"]:
                self.precision_greedy = comp.get("precision_greedy", None)
                if self.precision_greedy is None:
                    raise ValueError("Precision for greedy algorithm must be defined in the input file.")
        elif self.domain_decomposition_solver:
            raise ValueError("Schur complement computation method must be defined in the input file.")
        self.boundary_conditions = p.get("boundary_conditions", {})
        self.young_modulus, self.poisson_ratio = material_properties(self.material_name)

    # ------------------------------------------------------------------------------------------------
    @timing.category("simulation")
    @timing.timeit
    def reset_cell_with_new_radii(self, new_radii, index_cell: int = 0):
        """lattice_sim.py:1421-1497: give one cell new base radii and redo everything that depends on them - the
        struts of that cell, the penalisation lengths (L_zone depends on the neighbours' radii), boundary indices and
        boundary conditions.  Array-backed: the lattice is regenerated with a per-cell radius override."""
        if len(new_radii) != len(self.radii):
            raise ValueError("Invalid hybrid radii data.")
        if not (0 <= index_cell < self.lattice.n_cells):
            raise IndexError("Invalid cell index.")
        self.radii = [float(r) for r in new_radii]
        if getattr(self, "_cell_radii_override", None) is None:
            self._cell_radii_override = np.tile(np.asarray(self._base_radii, dtype=float), (self.lattice.n_cells, 1))
        self._cell_radii_override[index_cell] = self.radii
        self._generate_and_prepare()

    @timing.category("simulation")
    @timing.timeit
    def set_cell_radii(self, radii):
        """Give every cell its own radii, (C, n_geometries) in the cell order of ``lattice.cell_pos`` - what a loop of
        ``Cell.change_beam_radius`` (cell.py:896-917) does in the reference (LatticeOpti works this way).  The lattice
        arrays are regenerated once; in DDM mode the cell Schur complements are re-evaluated."""
        radii = np.asarray(radii, dtype=float).reshape(self.lattice.n_cells, len(self._base_radii))
        self._cell_radii_override = radii.copy()
        self._generate_and_prepare()
        if self.domain_decomposition_solver:
            self.calculate_schur_complement_cells()

    @timing.category("design")
    @timing.timeit
    def _generate_and_prepare(self):
        """Lattice arrays + penalisation + boundary indices + boundary conditions from the current parameters."""
        for dev in (getattr(self, "_device", None), getattr(self, "_ddm_device", None)):
            if dev is not None:
                dev.close()
        self._device = self._ddm_device = None
        if self.enable_randomness and self._cell_radii_override is None:
            # lattice.py:458-465: every cell draws its radii from the reference's seeded stream.  The position in that
            # stream depends on how many points the earlier cells created (LA.random_cell_radii): a geometry-only pass
            # first, then the lattice with those radii
            geo = LA.generate((self.cell_size_x, self.cell_size_y, self.cell_size_z),
                              (self.num_cells_x, self.num_cells_y, self.num_cells_z), self.geom_types,
                              self._base_radii, grad_radius=self.grad_radius, grad_dim=self.grad_dim,
                              erased_blocks=self.eraser_blocks, want_creator=True)
            self._cell_gfac = geo.cell_radii[:, 0] / self._base_radii[0]
            self._cell_radii_override = LA.random_cell_radii(geo.extras["node_creator"], geo.n_cells,
                                                             len(self._base_radii), self.range_radius,
                                                             self.randomness_hybrid)
        self.__dict__.pop("_surface_points_cache", None)      # (tables derived from the old topology)
        self.__dict__.pop("_gdi_cache", None)
        self.lattice = LA.generate((self.cell_size_x, self.cell_size_y, self.cell_size_z),
                                   (self.num_cells_x, self.num_cells_y, self.num_cells_z), self.geom_types,
                                   self._base_radii, grad_radius=self.grad_radius, grad_dim=self.grad_dim,
                                   erased_blocks=self.eraser_blocks,
                                   cell_radii_override=(None if self._cell_radii_override is None else
                                                        self._cell_radii_override * self._cell_gfac[:, None]))
        if self._cell_radii_override is None:      # gradient factor of Cell.get_radius (cell.py:385-412), per cell
            self._cell_gfac = self.lattice.cell_radii[:, 0] / self._base_radii[0]
        if self.symmetry_lattice is not None:      # lattice.py:90-91
            plane, point = self.symmetry_lattice["sym_plane"], self.symmetry_lattice["sym_point"]
            if plane is None or point is None:
                raise ValueError("Both symmetry_plane and reference_point must be provided.")
            self.lattice = LA.apply_symmetry(self.lattice, self.geom_types, plane, point)
            if len(self._cell_gfac) != self.lattice.n_cells:
                self._cell_gfac = np.concatenate([self._cell_gfac, self._cell_gfac])
        lat = self.lattice
        self.x_min, self.x_max, self.y_min, self.y_max, self.z_min, self.z_max = map(float, lat.bbox)
        self.penalized = None
        self.is_penalized = False
        self.beam_mult = None
        # lattice_sim.py:119-122: joints are penalised for the FEM path and for DDM with exact Schur complements;
        # with a surrogate the penalisation lives inside the stored Schur matrices
        if self.enable_simulation_properties and (not self.domain_decomposition_solver
                                                  or self.type_schur_complement_computation == "exact"):
            self.define_angles_between_beams()
            self.set_penalized_beams()
        else:
            self.lzone = np.zeros((lat.n_beams, 2))
            self.penalized = LA.penalize(lat, None)
        # rows of the per-node arrays: the design nodes; in reference_compat mode the reference's whole node list
        # (design nodes, then the penalisation points in coordinate order = Point.index, lattice.py:687-696)
        self._compat_rows = self.reference_compat and self.is_penalized
        if self._compat_rows:
            self._define_strut_multiplicity()
        # Struts the reference would hold several copies of at DESIGN level (hybrid collision on a strut shared by several
        # cells, apply_symmetry twins): the default model keeps each once - say so once, and leave a flag on the object
        dm = lat.extras.get("design_mult")
        self.differs_from_reference_by_strut_copies = bool(dm is not None and (np.asarray(dm) > 1).any()
                                                           and not self.reference_compat)
        if self.differs_from_reference_by_strut_copies and not LatticeSim._design_mult_note_done:
            LatticeSim._design_mult_note_done = True
            import warnings
            warnings.warn(f"{int((np.asarray(dm) > 1).sum())} struts of this lattice exist in several copies in the reference's "
                          "model (hybrid collision on struts shared by several cells / symmetry twins); the default "
                          "LatticeSim keeps each strut once.  Pass reference_compat=True (or PYLATTICE_REFERENCE_COMPAT=1) "
                          "for the reference's own model (INTEGRATION.md).", stacklevel=2)
        R = self.get_number_nodes() if self._compat_rows else lat.n_nodes
        self.fixed_DOF = np.zeros((R, 6), dtype=bool)
        self.displacement_vector = np.zeros((R, 6))
        self.applied_force = np.zeros((R, 6))
        self.reaction_force_vector = np.zeros((R, 6))
        self._cell_points = None
        self.define_node_index_boundary()
        self.set_boundary_conditions()

    def get_number_cells(self):
        return self.lattice.n_cells

    def get_lattice_boundary_box(self):
        return [self.x_min, self.x_max, self.y_min, self.y_max, self.z_min, self.z_max]

    # ------------------------------------------------------------------------------------------------
    @timing.category("design")
    @timing.timeit
    def define_angles_between_beams(self):
        """lattice.py:805-904.  Large non-periodic lattices take the device kernel (pl_lzone: the valence^2 angle search
        is 10^7-10^8 pair evaluations at 10^6 struts); small ones and periodic single cells the numpy restatement."""
        if not self.enable_periodicity and self.lattice.n_nodes >= 20000:
            from ._capi import lzone
            self.lzone = lzone(self.lattice.node_xyz, self.lattice.beam_conn, self.lattice.beam_radius)
        else:
            self.lzone = LA.compute_lzone(self.lattice, bool(self.enable_periodicity))

    @timing.category("simulation")
    @timing.timeit
    def set_penalized_beams(self):
        self.penalized = LA.penalize(self.lattice, self.lzone)
        self.is_penalized = True

    def _define_strut_multiplicity(self):
        """reference_compat: how many copies of every strut the reference's model holds.  set_penalized_beams loops
        ``for cell: for beam in cell.beams_cell`` and replaces a strut with a penalised end by NEW segment objects in that
        cell only (lattice_sim.py:250-303), so a strut shared by k cells ends up as k copies of each segment; a strut
        without penalised ends stays one shared object.  (Copies made by check_hybrid_collision come on top:
        ``lattice.extras["design_mult"]``.)"""
        lat, pen = self.lattice, self.penalized
        owners = np.bincount(lat.cell_beam_idx, minlength=lat.n_beams)
        split = (pen.seg_len[:, 0] > 0) | (pen.seg_len[:, 2] > 0)
        mult = np.where(split, owners, 1).astype(np.int64)
        dm = lat.extras.get("design_mult")
        if dm is not None and lat.extras.get("design_mult_kind") == "per_cell":
            # twins of apply_symmetry: every copy sits in one cell's beams_cell and is penalised there once - as many
            # segment copies as owner cells (counted by the cell -> strut table), dm objects when nothing is split
            mult = np.where(split, owners, dm)
        elif dm is not None:
            # every design copy is a separate object in every owner cell's beams_cell (lattice.py:1188-1195): each is
            # penalised once per owner cell
            mult = np.where(split, owners * dm, dm)
        self.beam_mult = mult

    def cell_points(self):
        """CSR cell -> rows of the per-node arrays: ``Cell.points_cell`` of the reference.  Design nodes of the cell; in
        reference_compat mode also the penalisation points of the cell's struts (cell.add_point, lattice_sim.py:301)."""
        if self._cell_points is None:
            lat = self.lattice
            if not self._compat_rows:
                self._cell_points = (lat.cell_node_ptr, lat.cell_node_idx)
            else:
                from .views import _tables
                t = _tables(self)
                cell_of_b = np.repeat(np.arange(lat.n_cells), np.diff(lat.cell_beam_ptr))
                pens = t.pen_id[lat.cell_beam_idx]                    # (n_pairs, 2)
                ok = pens >= 0
                rows = np.concatenate([np.repeat(np.arange(lat.n_cells), np.diff(lat.cell_node_ptr)),
                                       np.repeat(cell_of_b, 2).reshape(-1, 2)[ok]])
                cols = np.concatenate([lat.cell_node_idx, pens[ok]])
                self._cell_points = LA._csr_from_pairs(rows, cols, lat.n_cells)
        return self._cell_points

    def node_coordinates(self):
        """(rows, 3) coordinates of the rows of the per-node arrays."""
        if self._compat_rows:
            from .views import _tables
            return _tables(self).node_xyz
        return self.lattice.node_xyz

    def reset_penalized_beams(self) -> None:
        """lattice_sim.py:313-400: undo set_penalized_beams - every strut is one segment again, the penalisation points
        go away (the design struts and nodes were never replaced here, so nothing has to be rewired); a device handle
        built for the penalised model is dropped."""
        if not self.is_penalized:
            print("Warning: lattice does not appear to be penalized.")
            return
        self.lzone = np.zeros((self.lattice.n_beams, 2))
        self.penalized = LA.penalize(self.lattice, None)
        self.is_penalized = False
        if self._device is not None:
            self._device.close()
            self._device = None
        for name in ("_view_tables", "_node_mod_store"):
            if hasattr(self, name):
                delattr(self, name)

    @timing.category("simulation")
    @timing.timeit
    def define_node_index_boundary(self):
        """Boundary index of every node lying on the box of one of its cells (lattice_sim.py:546-563), numbered in
        the order get_global_displacement visits them (cells in order, nodes by rounded coordinates)."""
        lat = self.lattice
        ptr, idx = self.cell_points()
        node_xyz = self.node_coordinates()
        N = len(node_xyz)
        if N >= 20000:   # large lattices: the same rule in multi-threaded C++ (pl_boundary_index / pl_boundary_index_rows)
            from ._capi import boundary_index
            self.index_boundary, visit = boundary_index(ptr, idx, node_xyz, lat.cell_coord, lat.cell_size,
                                                        by_coordinates=bool(self._compat_rows))
            self.max_index_boundary = len(visit) - 1
            self._boundary_visit_order = visit
            return
        self.index_boundary = np.full(N, -1, np.int64)
        on_box = np.zeros(N, bool)
        cell_of = np.repeat(np.arange(lat.n_cells), np.diff(ptr))
        xyz = node_xyz[idx]
        lo, hi = lat.cell_coord[cell_of], lat.cell_coord[cell_of] + lat.cell_size[cell_of]
        on = ((xyz == lo) | (xyz == hi)).any(axis=1)
        on_box[idx[on]] = True
        # visit order: cell-major, then (round(x,9), round(y,9), round(z,9), index); node index already sorts by xyz
        inside = np.ones(len(idx), bool)
        inside[ptr[1:-1]] = False                           # first entry of every cell but the first
        if self._compat_rows:
            # rows inside a cell come in (round(x, 9), round(y, 9), round(z, 9), index) order (_sorted_nodes,
            # lattice_sim.py:193-199): design nodes and penalisation points interleave
            key = np.round(xyz, 9)
            idx = idx[np.lexsort((idx, key[:, 2], key[:, 1], key[:, 0], cell_of))]
        elif len(idx) > 1 and not np.all((np.diff(idx) > 0) | ~inside[1:]):
            idx = idx[np.lexsort((idx, cell_of))]           # rows not yet ascending inside their cell
        seq = idx[on_box[idx]]
        # first visit of every node, without sorting: of duplicate targets of a fancy assignment the last one written
        # stays, so writing the positions in reverse leaves the first one
        pos = np.arange(len(seq))
        first = np.full(N, -1, np.int64)
        first[seq[::-1]] = pos[::-1]
        visit = seq[first[seq] == pos]
        self.index_boundary[visit] = np.arange(len(visit))
        self.max_index_boundary = len(visit) - 1
        self._boundary_visit_order = visit

    # ------------------------------------------------------------------------------------------------
    def get_cells_on_surfaces(self, surfaces):
        """lattice.py:1363-1411: iterative extrema filter on the integer cell positions."""
        pos = self.lattice.cell_pos
        cand = np.arange(len(pos))
        for token in surfaces:
            t = token.strip().lower()
            if not t:
                continue
            ax = {"x": 0, "y": 1, "z": 2}.get(t[0])
            if ax is None:
                raise ValueError(f"Invalid axis in constraint '{token}', expected X/Y/Z with min/max.")
            if "min" in t:
                ext = pos[cand, ax].min()
            elif "max" in t:
                ext = pos[cand, ax].max()
            else:
                raise ValueError(f"Invalid extrema in constraint '{token}', expected 'min' or 'max'.")
            cand = cand[pos[cand, ax] == ext]
            if len(cand) == 0:
                return cand
        return cand

    def find_point_on_lattice_surface(self, surfaceNames, surface_cells=None):
        """Node ids on the given surfaces (lattice.py:1320-1359)."""
        if not all(s in _SURFACES for s in surfaceNames):
            raise ValueError("Invalid surface name_lattice(s).")
        lat = self.lattice
        # (the node sets of a lattice's surfaces do not change while its topology stands: an optimisation re-applies the same
        #  boundary conditions before every simulation - 4.5 ms of an 18-ms objective + gradient at 24^3 cells went here)
        cache = self.__dict__.setdefault("_surface_points_cache", {})
        ckey = (id(lat), lat.n_nodes, tuple(surfaceNames), None if surface_cells is None else tuple(surface_cells))
        if ckey in cache:
            return cache[ckey].copy()
        cells = self.get_cells_on_surfaces(surfaceNames)
        names = surface_cells if surface_cells is not None else surfaceNames
        ptr, idx = self.cell_points()
        node_xyz = self.node_coordinates()
        # every (surface cell, point of that cell) pair at once (a Python loop over the cells cost 0.1 s at 50^3 cells)
        cells = np.asarray(cells, dtype=np.int64)
        if len(cells):
            cnt = (ptr[cells + 1] - ptr[cells]).astype(np.int64)
            owner = np.repeat(np.arange(len(cells)), cnt)
            first = np.repeat(np.asarray(ptr)[cells].astype(np.int64) - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt)
            nodes = np.asarray(idx)[first + np.arange(int(cnt.sum()))]
            keep = np.ones(len(nodes), bool)
            for s in names:
                ax = "XYZ".index(s[0])
                val = lat.cell_coord[cells, ax] + (lat.cell_size[cells, ax] if s.endswith("max") else 0.0)
                # (min and mid both refer to the cell's lower corner, cell.py:451-462)
                keep &= node_xyz[nodes, ax] == val[owner]
            pts = np.unique(nodes[keep])
        else:
            pts = np.zeros(0, np.int64)
        if len(pts) == 0:
            raise ValueError("No points found on the specified surfaces.")
        cache[ckey] = pts
        return pts.copy()

    def apply_constraints_nodes(self, surfaces, value, DOF, type_constraint="Displacement", surface_cells=None):
        """lattice_sim.py:405-458: displacement -> value + fixed flag; force -> total / number of target nodes
        whose dof is free."""
        pts = self.find_point_on_lattice_surface(surfaces, surface_cells)
        pts = pts[self.index_boundary[pts] >= 0]
        if len(pts) == 0:
            raise ValueError("No nodes found on the specified surfaces for constraint application.")
        targets = {d: int((~self.fixed_DOF[pts, d]).sum()) for d in DOF}
        for val, d in zip(value, DOF):
            if type_constraint == "Displacement":
                self.displacement_vector[pts, d] = val
                self.fixed_DOF[pts, d] = True
            elif type_constraint == "Force":
                self.applied_force[pts, d] = val / max(1, targets[d])
            else:
                raise ValueError("Invalid type of constraint. Use 'Displacement' or 'Force'.")

    @timing.category("simulation")
    @timing.timeit
    def set_boundary_conditions(self):
        """lattice_sim.py:460-494."""
        for key, block in self.boundary_conditions.items():
            if key not in ["Force", "Displacement"]:
                raise ValueError(f"Invalid boundary condition type: {key}. Must be 'Force' or 'Displacement'.")
            for _name, data in block.items():
                if "Surface" not in data or "Value" not in data or "DOF" not in data:
                    raise ValueError("Invalid boundary condition data. 'Surface', 'Value' and 'DOF' are required.")
                if not isinstance(data["Surface"], list):
                    raise ValueError("Surface must be a list of strings.")
                if not isinstance(data["Value"], list):
                    raise ValueError("Value must be a list of floats.")
                if not isinstance(data["DOF"], list):
                    raise ValueError("DOF must be a list of strings.")
                if len(data["Value"]) != len(data["DOF"]):
                    raise ValueError("Value and DOF must have the same length.")
                if not all(d in _DOF_MAP for d in data["DOF"]):
                    raise ValueError("DOF must be one of 'X', 'Y', 'Z', 'RX', 'RY', 'RZ'.")
                if not all(s in _SURFACES for s in data["Surface"]):
                    raise ValueError("Surface must be one of 'Xmin', 'Xmax', 'Ymin', 'Ymax', 'Zmin', 'Zmax', "
                                     "'Xmid', 'Ymid', 'Zmid'.")
                self.apply_constraints_nodes(data["Surface"], data["Value"], [_DOF_MAP[d] for d in data["DOF"]],
                                             key, data.get("SurfaceCells", None))

    # ------------------------------------------------------------------------------------------------
    @timing.category("simulation")
    @timing.timeit
    def get_global_displacement(self, withFixed: bool = False, OnlyImposed: bool = False):
        """lattice_sim.py:502-542: free dofs of the cell-boundary nodes in visit order."""
        V = np.asarray(self._boundary_visit_order, dtype=np.int64)
        U, fixed = self.displacement_vector[V], self.fixed_DOF[V]
        idx6 = np.repeat(self.index_boundary[V].astype(np.int64)[:, None], 6, axis=1)
        if not OnlyImposed:                       # free dofs (+ the constrained ones if asked for), node-major
            take = ~fixed | bool(withFixed)
            disp, index = U[take], idx6[take]
        else:                                     # every dof: 0 for unloaded free ones (no index entry for those)
            silent = ~fixed & (self.applied_force[V] == 0)
            disp, index = np.where(silent, 0.0, U).ravel(), idx6[~silent]
        # a Python list, as in the reference, whatever the size (one type for every caller: `index + other`, `.index()`,
        # JSON ...; 3 M entries cost ~50 ms at 50^3 cells); the int64 array stays available beside it
        self.global_displacement_index_array = index if not OnlyImposed else getattr(self, "global_displacement_index_array", None)
        prev = getattr(self, "_gdi_cache", None)      # (the list only changes with the Dirichlet mask: a design loop asks
        if prev is not None and prev[0].shape == index.shape and np.array_equal(prev[0], index):   # for the same one again)
            index = list(prev[1])
        else:
            arr = index
            index = index.tolist()
            self._gdi_cache = (arr.copy(), list(index))
        if not OnlyImposed:
            self.global_displacement_index = index
        return np.asarray(disp, dtype=float), index

    # ------------------------------------------------------------------------------------------------
    def device_model(self, **kw):
        """The HIP handle for this lattice (created on first use; no CPU fallback)."""
        from ._capi import HipLattice
        if self._device is None:
            pen = self.penalized
            if self.lattice.n_nodes <= DIRECT_MAX_NODES:
                # the sizes of the reference's own presets (6x3x3 cells = 166 nodes): a slender cantilever of a few hundred nodes
                # needs 200 - 300 PCG iterations whatever the preconditioner (3.3 ms), the dense Cholesky factor of its ~1 000
                # dofs takes 16 launches; the PCG then converges in one or two steps (tools/experiments/small_lattice_precond.py)
                # (measured, profiles/r05_g_small_precond.txt: BCC 6x3x3 3.05 -> 1.66 ms per assembly + solve, 8x4x4 4.5 -> 3.2;
                # beyond ~350 nodes the factorisation chain of n / 64 links overtakes the PCG, and Octet lattices, which
                # converge in 60 - 100 iterations, are served better by the PCG from 300 nodes on)
                kw.setdefault("precond", 5)
            if self.lattice.n_nodes >= MULTILEVEL_MIN_NODES:
                # multi-level preconditioner + record palette (what bench.py measures).  Round 5: from 600 nodes instead of
                # 20 000 - with the short iteration (pl_small.h) and small tiles it beats Jacobi from there on (BCC 16x8x8:
                # 10.3 -> 5.2 ms, Octet 10x5x5 2.9 -> 1.8, 12^3 Octet 4.8 -> 2.1, 16^3 BCC 11.9 -> 5.8 ms per assembly + solve)
                kw.setdefault("precond", 3)
                kw.setdefault("palette", 1)
                if self.lattice.n_nodes < SMALL_TILES_MAX_NODES:
                    kw.setdefault("tile_nodes", 32)
                    kw.setdefault("coarse_modes", 12)
            if self.lattice.n_nodes >= FP32_VECTORS_MIN_NODES:
                # from ~2 M nodes the PCG vectors no longer fit the caches and stream from HBM: fp32 inner solves with fp64
                # refinement (rtol still refers to the TRUE fp64 residual).  Measured: 100^3 Octet 284 against 234 M
                # beams/s, 100^3 BCC 80.8 against 76.3; 50^3 Octet (0.77 M nodes) 220 against 230 - hence the threshold
                kw.setdefault("precision", 1)
            if self._compat_rows:
                from .compat_device import CompatDevice
                self._device = CompatDevice(self, **kw)
                return self._device
            self._device = HipLattice(self.lattice.node_xyz, self.lattice.beam_conn, self.lattice.beam_radius,
                                      pen.seg_len, pen.seg_nsub, self.young_modulus, self.poisson_ratio,
                                      pen_coef=self.penalization_coefficient, **kw)
        return self._device

    # ------------------------------------------------------------------------------------------------
    # Domain decomposition (lattice_sim.py:846-919, 1111-1252)
    # ------------------------------------------------------------------------------------------------
    def cell_boundary_nodes(self):
        """(C, n_b) node ids of every cell in the order of Cell.define_node_order_to_simulate (cell.py:611-680)."""
        from .utils_schur import node_order_all_cells
        rows = node_order_all_cells(self)
        if rows is None:
            raise NotImplementedError("cells with different numbers of boundary nodes")
        return rows

    def set_schur_complements(self, S, cell_index=None):
        """Install cell Schur complements: one (6n_b)^2 matrix for every cell, or a stack plus a per-cell index.
        (The reference fills Cell.schur_complement from dolfinx or from its reduced-basis surrogates,
        lattice_sim.py:846-978; the surrogate files are outside this repository.)"""
        S = np.asarray(S, dtype=np.float64)
        if S.ndim == 2:
            S = S[None]
        self.schur_complements = S
        self.cell_schur_index = (np.zeros(self.lattice.n_cells, np.int32) if cell_index is None
                                 else np.asarray(cell_index, np.int32))
        dev = self._ddm_device
        if dev is not None:
            if S.shape[1] == dev._m and len(self.cell_schur_index) == dev._n_cells:
                # same cells, other matrices (every iteration of a design loop): the handle takes them (round 5; it used to be
                # destroyed and created again - 10 ms per solve_DDM on a 12 x 4 x 4 lattice against 2 ms of device work)
                dev.update_ddm_matrices(S, self.cell_schur_index)
                if getattr(self, "_ddm_precond", 0) == 2:
                    self.define_preconditioner()
            else:
                dev.close()
                self._ddm_device = None

    @timing.category("simulation")
    @timing.timeit
    def calculate_schur_complement_cells(self):
        """Exact Schur complement of one representative cell per (geometry, radii) group (lattice_sim.py:846-919),
        condensed on the device (pl_schur) from that cell's own struts with their penalised segments."""
        from ._capi import HipLattice
        if self.type_schur_complement_computation != "exact":
            return self._surrogate_schur_complement_cells()
        lat, pen = self.lattice, self.penalized
        cb = self.cell_boundary_nodes()
        par = self._cell_parameter_radii()       # the reference groups by Cell.radii, which ignores the preset gradient
        keys = [tuple(np.round(par[c], 8)) for c in range(lat.n_cells)]
        groups, mats, grads, idx = {}, [], [], np.zeros(lat.n_cells, np.int32)
        for c, k in enumerate(keys):
            if k not in groups:
                beams = lat.cell_beam_idx[lat.cell_beam_ptr[c]:lat.cell_beam_ptr[c + 1]]
                nodes = np.unique(lat.beam_conn[beams])
                remap = np.full(lat.n_nodes, -1, np.int64)
                remap[nodes] = np.arange(len(nodes))
                with HipLattice(lat.node_xyz[nodes], remap[lat.beam_conn[beams]], lat.beam_radius[beams],
                                pen.seg_len[beams], pen.seg_nsub[beams], self.young_modulus, self.poisson_ratio,
                                pen_coef=self.penalization_coefficient, reorder=0,
                                **({"precond": 5} if 6 * len(nodes) <= DDM_DENSE_MAX else {})) as dev:   # (see cell_device)
                    dev.assemble()
                    mats.append(dev.schur(remap[cb[c]], rtol=1e-13, max_iter=200000))
                    if self.enable_gradient_computing:
                        # _compute_schur_gradients (lattice_sim.py:1020-1054): central differences in every radius
                        # parameter of the cell, h = max(1e-8, 1e-6 max(1, |r|)), radii changed at FIXED penalised
                        # segment lengths (Cell.change_beam_radius, cell.py:896-917)
                        gl = []
                        for j, rj in enumerate(par[c]):
                            h = max(1e-8, 1e-6 * max(1.0, abs(rj)))
                            rp, rm = rj + h, max(1e-12, rj - h)
                            S_pm = []
                            for rv in (rp, rm):
                                rad = lat.beam_radius[beams].copy()
                                sel = lat.beam_type[beams] == j
                                rad[sel] = rv * self._cell_gfac[c]
                                dev.update_radii(rad)
                                dev.assemble()
                                S_pm.append(dev.schur(remap[cb[c]], rtol=1e-13, max_iter=200000))
                            gl.append((S_pm[0] - S_pm[1]) / (rp - rm))
                        grads.append(gl)
                groups[k] = len(mats) - 1
            idx[c] = groups[k]
        self.schur_gradients = grads if self.enable_gradient_computing else None
        self._schur_gradients_array = None
        self.set_schur_complements(np.stack(mats), idx)

    def _surrogate_schur_complement_cells(self):
        """Surrogate branch of lattice_sim.py:846-919: one batched evaluation S(r) = B alpha(r) for the distinct
        radius sets of the lattice (keys rounded to 8 decimals like the reference's cache), plus dS/dr when
        enable_gradient_computing is set."""
        if self.schur_surrogate is None:
            raise NotImplementedError("Not implemented schur complement computation method.")
        par = np.asarray(self._cell_parameter_radii(), dtype=float)
        # distinct radius sets, rounded to 8 decimals like the reference's cache keys (one numpy pass; the per-cell tuples and the
        # dictionary of the first version were 10 ms at 4 096 cells)
        uniq, idx = np.unique(np.round(par.reshape(len(par), -1), 8), axis=0, return_inverse=True)
        idx = np.asarray(idx, dtype=np.int32).ravel()
        radii_batch = uniq.tolist()
        S = self.schur_surrogate.schur_batch(radii_batch)
        self.schur_gradients, self._schur_gradients_array = None, None
        if self.enable_gradient_computing:
            G = self.schur_surrogate.schur_gradients_batch(radii_batch)          # (n_q, d, n, n) in one go, or None
            if G is not None:
                self._schur_gradients_array = G
                self.schur_gradients = [list(g) for g in G]                      # (views)
            else:
                self.schur_gradients = [self.schur_surrogate.schur_gradients(r) for r in radii_batch]
        if self._verbose > 1:
            print("Number of unique Schur complements computed:", len(radii_batch))
        self.set_schur_complements(S, idx)

    def get_schur_complement_from_reduced_basis_batch(self, geometric_params_list):
        """lattice_sim.py:919-977."""
        if self.schur_surrogate is None:
            raise NotImplementedError("Not implemented schur complement computation method.")
        return self.schur_surrogate.schur_batch(geometric_params_list)

    def get_schur_complement_from_reduced_basis(self, geometric_params):
        """lattice_sim.py:979-1018."""
        return self.get_schur_complement_from_reduced_basis_batch([list(geometric_params)])[0]

    def ddm_model(self):
        from ._capi import HipLattice
        if self.schur_complements is None:
            raise ValueError("Schur complements are not defined: call calculate_schur_complement_cells() or "
                             "set_schur_complements()")
        if self._ddm_device is None:
            cb = self.cell_boundary_nodes()
            n_nodes = self.max_index_boundary + 1
            # enable_preconditioner: the reference factorises the assembled Schur matrix (lattice_sim.py:1333-1415).
            # The device does the same (dense Cholesky, precond = 2) up to DDM_DENSE_MAX dofs; beyond that the CG gets
            # the node-block Jacobi preconditioner of the same matrix (its 6 x 6 diagonal blocks, inverted: same solution,
            # more iterations - a quarter fewer than with the diagonal alone).
            # Round 5: above the dense limit the node blocks get a dense level on top (precond = 4: 12 rigid / strain modes
            # per aggregate of boundary nodes, Galerkin operator from the cell matrices) - the iteration count then stops
            # growing with the lattice (32^3 BCC cells: 2.2 x fewer iterations than the node blocks alone).
            self._ddm_precond = 0
            xyz = None
            if self.enable_preconditioner:
                self._ddm_precond = 2 if 6 * n_nodes <= DDM_DENSE_MAX else DDM_LARGE_PRECOND
                if self._ddm_precond == 4:
                    xyz = self.lattice.node_xyz[self._boundary_nodes_by_index()]
            # CG parameters of the reference's solve_DDM (lattice_sim.py:1156-1159): alpha clamp 100, direction-norm
            # stop 1e-12, restart every 500 000 iterations
            self._ddm_device = HipLattice.ddm(n_nodes, self.index_boundary[cb], self.schur_complements,
                                              self.cell_schur_index, precond=self._ddm_precond, alpha_max=100.0,
                                              mintol=1e-12, restart_every=500000, node_xyz=xyz,
                                              coarse_max_dofs=DDM_COARSE_MAX_DOFS)
            if self._ddm_precond == 2:
                self.define_preconditioner()
        return self._ddm_device

    def _define_preconditioner_approximation(self):
        """lattice_sim.py:1312-1329: the dataset the approximate preconditioners are built from -
        ``Schur_complement_mean_<geoms>.npz`` ("mean") or ``Schur_complement_<geoms>.npz`` ("nearest_reference") under
        ``data/outputs/schur_complement/`` of a data root; nothing for "exact"."""
        if self.preconditioner_type == "exact":
            return None
        if self.preconditioner_type not in ("mean", "nearest_reference"):
            raise NotImplementedError("Not implemented preconditioner approximation method.")
        stem = "Schur_complement_mean_" if self.preconditioner_type == "mean" else "Schur_complement_"
        name = stem + "_".join(str(g) for g in self.geom_types) + ".npz"
        roots = list(self.data_roots or [])
        if os.environ.get("PYLATTICE_DATA_ROOT"):
            roots.append(os.environ["PYLATTICE_DATA_ROOT"])
        roots.append(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        for root in roots:
            for sub in (os.path.join("data", "outputs", "schur_complement"), ""):
                path = os.path.join(root, sub, name)
                if os.path.isfile(path):
                    self.used_schur_preconditioner = np.load(path, allow_pickle=True)
                    return path
        if self.preconditioner_type == "mean":
            # the reference's checkout does not carry its Schur_complement_mean_*.npz files: build the mean matrix from
            # the radius dataset of the same geometries, else from this lattice's own cell matrices
            full = "Schur_complement_" + "_".join(str(g) for g in self.geom_types) + ".npz"
            for root in roots:
                for sub in (os.path.join("data", "outputs", "schur_complement"), ""):
                    path = os.path.join(root, sub, full)
                    if os.path.isfile(path):
                        mats = np.asarray(np.load(path, allow_pickle=True)["schur_matrices"], dtype=float)
                        self.used_schur_preconditioner = {"schur_matrices": mats.mean(axis=0)}
                        return path
            self._mean_of_own_cells = True      # re-evaluated whenever the cell matrices change (define_preconditioner)
            return None
        raise FileNotFoundError(f"Schur complement dataset for the '{self.preconditioner_type}' preconditioner not "
                                f"found: {name} (looked under {roots})")

    @timing.category("simulation")
    @timing.timeit
    def define_preconditioner(self):
        """lattice_sim.py:1333-1415 (define_preconditioner + build_preconditioner): choose the cell matrices the
        assembled preconditioner is made of and hand them to the device, which assembles and factorises
        ``sum_c B_c^T Shat_c B_c`` at the next ``assemble()``.  "exact": the cells' own Schur complements; "mean": the one
        matrix of the mean dataset for every cell; "nearest_reference": for every cell the dataset matrix whose radii
        are nearest (Euclidean, first on ties - sklearn's NearestNeighbors(n_neighbors=1) in the reference) to
        Cell.radii."""
        dev = self._ddm_device
        if dev is None or not self.enable_preconditioner or getattr(self, "_ddm_precond", 0) != 2:
            return
        if self.preconditioner_type == "exact" or self.preconditioner_type is None:
            dev.set_ddm_preconditioner(None)
            return
        if self.used_schur_preconditioner is None and not getattr(self, "_mean_of_own_cells", False):
            self._define_preconditioner_approximation()
        if self.used_schur_preconditioner is None:          # "mean" without any dataset: mean over this lattice's cells
            w = np.bincount(self.cell_schur_index, minlength=len(self.schur_complements)).astype(float)
            dev.set_ddm_preconditioner(np.tensordot(w / w.sum(), self.schur_complements, axes=(0, 0)))
            return
        data = self.used_schur_preconditioner
        mats = np.asarray(data["schur_matrices"], dtype=float)
        if self.preconditioner_type == "mean":
            dev.set_ddm_preconditioner(mats.reshape(-1, mats.shape[-2], mats.shape[-1])[:1])
            return
        pts = np.asarray(data["radius_values"], dtype=float)
        if pts.ndim == 1:
            pts = pts[:, None]
        radii = self._cell_parameter_radii()
        d2 = ((radii[:, None, :] - pts[None, :, :]) ** 2).sum(axis=2)
        dev.set_ddm_preconditioner(mats, np.argmin(d2, axis=1).astype(np.int32))

    def _cell_parameter_radii(self):
        """``Cell.radii`` of the reference for every cell: the radii the cell was GIVEN (preset value,
        reset_cell_with_new_radii, LatticeOpti), which the preset's radius gradient does not touch - it only scales
        the struts (cell.py:86,407-412).  The key of the reference's Schur-complement cache and the argument of its
        surrogates and of the nearest-reference preconditioner."""
        return self.lattice.cell_radii / self._cell_gfac[:, None]

    @timing.category("simulation")
    @timing.timeit
    def solve_DDM(self):
        """Domain-decomposition solve on the cell-boundary nodes (lattice_sim.py:1111-1176): right-hand side
        b = f_free - (S u_imposed)_free, plain CG with the reference's parameters (tol 1e-6, alpha clamp 100,
        max_iterations from the preset), results written back to displacement_vector; returns
        (xsol, info, global_displacement_index, b) or four None when b == 0."""
        if not self.domain_decomposition_solver:
            raise ValueError("LatticeSim was not created with enable_domain_decomposition_solver=True")
        dev = self.ddm_model()
        if self._ddm_precond in (1, 3, 4) and not getattr(self, "_precond_note_done", False):
            print(f"solve_DDM: {6 * (self.max_index_boundary + 1)} boundary dofs exceed the {DDM_DENSE_MAX} the device "
                  "factorises densely for the assembled-Schur preconditioner; running CG preconditioned by its node blocks"
                  + (" and a dense level on aggregates of nodes" if self._ddm_precond == 4 else "") +
                  " to the same tolerance (max_iterations of the preset then only applies if larger than 20000)")
            self._precond_note_done = True
        bn = self._boundary_nodes_by_index()
        fixed = self.fixed_DOF[bn]
        if (~fixed).sum() == 0:
            raise ValueError("No free DOF in the lattice. Process aborted.")
        ubar = np.where(fixed, self.displacement_vector[bn], 0.0)
        f = self.applied_force[bn]
        dev.set_bc(fixed, ubar, f)
        dev.assemble()
        b = np.where(fixed, 0.0, f - dev.spmv(ubar))
        if not b.any():       # (not np.linalg.norm: a threaded BLAS dot whose spinning worker threads starve the HIP runtime's
                              #  helper threads - measured at 32^3 cells: 12 ms in the norm and +70 ms in every other solve)
            print("No external forces or imposed displacements in the lattice. Process aborted.")
            return None, None, None, None
        maxit = self.number_iteration_max or 1000
        if self._ddm_precond in (1, 3, 4):
            # presets written for the LU-preconditioned CG cap it at a handful of iterations; the block-Jacobi CG that
            # replaces it above the dense limit needs O(sqrt(cond)) of them to reach the same 1e-6
            maxit = max(maxit, 20000)
        u, st = dev.solve(rtol=1e-6, max_iter=maxit, raise_on_noconv=False)
        if self._ddm_precond == 2 and int(st["precond_used"]) != 2:
            # the assembled matrix was not positive definite (surrogate matrices far from their training points can be
            # indefinite): the device ran Jacobi CG; give it the room the factorised preconditioner would not need
            if not getattr(self, "_spd_note_done", False):
                print("solve_DDM: the assembled Schur preconditioner is not positive definite (indefinite surrogate "
                      "matrix?); Jacobi-preconditioned CG instead")
                self._spd_note_done = True
            if not st["converged"] and maxit < 20000:
                u, st = dev.solve(rtol=1e-6, max_iter=20000, raise_on_noconv=False)
        self.iteration = st["iterations"]
        info = int(st["info"])               # 0 converged, 1 not, 2 not and a step fell below 1e-6 (the reference's codes)
        if info and self._verbose > -1:
            print(f"Conjugate Gradient did not converge ({self.iteration} iterations, relative residual "
                  f"{st['rel_residual']:.2e}).")
        self.displacement_vector[bn] = u
        self.reaction_force_vector[bn] = dev.reactions(u)
        xsol, idx = self.get_global_displacement()
        return xsol, info, self.global_displacement_index, b[~fixed]

    def _boundary_nodes_by_index(self):
        bn = np.empty(self.max_index_boundary + 1, np.int64)
        nodes = np.flatnonzero(self.index_boundary >= 0)
        bn[self.index_boundary[nodes]] = nodes
        return bn


# -- design-side attributes of the reference's base class (lattice.py:58-60, 346-385), for every class on the array model
def _design_properties(cls):
    from statistics import mean
    cls.size_x = property(lambda self: float(self.x_max - self.x_min))
    cls.size_y = property(lambda self: float(self.y_max - self.y_min))
    cls.size_z = property(lambda self: float(self.z_max - self.z_min))

    def get_relative_density(self) -> float:
        """Mean of the cells' relative densities (lattice.py:346-360)."""
        return float(mean(c.relative_density for c in self.cells))

    def get_beam_radius_min_max(self):
        r = self.lattice.beam_radius
        return float(r.max()), float(r.min())

    def getName(self):
        return getattr(self, "name_lattice", "lattice")
    cls.get_relative_density = get_relative_density
    cls.get_beam_radius_min_max = get_beam_radius_min_max
    cls.getName = getName


_design_properties(LatticeSim)
