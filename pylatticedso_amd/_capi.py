"""ctypes binding of libpylattice_hip.so (C ABI: include/pylattice_hip.h).

There is NO CPU fallback: if the shared library is missing or no MI355X is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .timing import timing as _timing

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYLATTICE_HIP_LIB: load another build of the library (kernel experiments: tools/exp_variants.sh)
LIB_PATH = os.environ.get("PYLATTICE_HIP_LIB") or os.path.join(_HERE, "libpylattice_hip.so")

PL_OK, PL_ERR_ARG, PL_ERR_HIP, PL_ERR_STATE, PL_ERR_NOCONV, PL_ERR_NAN, PL_ERR_NODEVICE = 0, -1, -2, -3, -4, -5, -6

# every symbol include/pylattice_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = ["pl_default_opts", "pl_opts_size", "pl_stats_size", "pl_abi_version", "pl_last_error", "pl_version", "pl_lzone", "pl_create", "pl_create_ddm",
           "pl_ddm_set_preconditioner", "pl_ddm_set_geometry", "pl_ddm_update_matrices", "pl_destroy", "pl_set_bc", "pl_set_periodic",
           "pl_update_radii", "pl_set_multiplicity", "pl_update_segments", "pl_assemble", "pl_assemble_bsr", "pl_get_bsr", "pl_spmv",
           "pl_spmv_free", "pl_spmv_bsr", "pl_solve", "pl_reactions", "pl_sens", "pl_energy", "pl_node_mod", "pl_schur",
           "pl_get_records", "pl_time_kernel", "pl_algorithmic_bytes", "pl_forget_history", "pl_debug_spd_solve", "pl_dist_unique_id_bytes",
           "pl_dist_unique_id", "pl_dist_loopback_id", "pl_dist_abort", "pl_dist_init", "pl_dist_set_peers", "pl_generate_lattice", "pl_lattice_fetch",
           "pl_lattice_free", "pl_penalize", "pl_boundary_index", "pl_boundary_index_rows"]


class PlMesh(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_beams", C.c_int64), ("node_xyz", C.c_void_p),
                ("beam_conn", C.c_void_p), ("beam_radius", C.c_void_p), ("seg_len", C.c_void_p),
                ("seg_nsub", C.c_void_p)]


class PlOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
                ("young", C.c_double), ("poisson", C.c_double), ("kappa", C.c_double), ("pen_coef", C.c_double),
                ("device", C.c_int32), ("spmv_kernel", C.c_int32), ("precond", C.c_int32), ("reorder", C.c_int32),
                ("check_every", C.c_int32), ("lanes_per_node", C.c_int32),
                ("tile_nodes", C.c_int32), ("coarse_max_dofs", C.c_int32), ("palette", C.c_int32),
                ("local_max_dofs", C.c_int32), ("precision", C.c_int32), ("restart_every", C.c_int32),
                ("alpha_max", C.c_double), ("grid_lo", C.c_double * 3), ("grid_hi", C.c_double * 3),
                ("grid_nodes", C.c_int64), ("mintol", C.c_double), ("compact_records", C.c_int32),
                ("condense", C.c_int32), ("chol_persistent", C.c_int32), ("cg_form", C.c_int32),
                ("tile_modes", C.c_int32), ("coarse_modes", C.c_int32), ("overlap", C.c_int32),
                ("coarse_storage", C.c_int32), ("warm_start", C.c_int32), ("short_iteration", C.c_int32)]


class PlStats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("iterations", C.c_int32), ("converged", C.c_int32),
                ("reserved_i", C.c_int32), ("rel_residual", C.c_double),
                ("b_norm", C.c_double), ("ms_assembly", C.c_double), ("ms_solve", C.c_double),
                ("ms_spmv_avg", C.c_double), ("precond_used", C.c_double), ("restarts", C.c_double),
                ("precision_used", C.c_double), ("info", C.c_double), ("stop_reason", C.c_double),
                ("condensed_nodes", C.c_double), ("cg_form_used", C.c_double), ("kp_form", C.c_double),
                ("comm_world", C.c_double), ("comm_rank", C.c_double), ("short_iteration_used", C.c_double),
                ("reserved", C.c_double * 2)]


class PlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpylattice_hip error {code}: {msg}")
        self.code = code


_lib = None


def load_library(path: str | None = None):
    """dlopen libpylattice_hip.so (built in-tree by pylatticedso_amd/csrc/Makefile or __graft_entry__.build())."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FileNotFoundError(f"{p} not found - build it with `make -C pylatticedso_amd/csrc` "
                                "(the HIP path has no CPU fallback)")
    lib = C.CDLL(p)
    lib.pl_last_error.restype = C.c_char_p
    lib.pl_version.restype = C.c_char_p
    lib.pl_destroy.restype = None
    for name in ("pl_opts_size", "pl_stats_size", "pl_abi_version"):
        getattr(lib, name).restype = C.c_uint32
    lib.pl_lattice_free.restype = None
    V, I32, I64, D = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    sig = {"pl_default_opts": [V, C.c_uint32], "pl_opts_size": [], "pl_stats_size": [], "pl_abi_version": [], "pl_lzone": [I32, I64, I64, V, V, V, V], "pl_create": [V, V, V], "pl_create_ddm": [I64, I64, I32, V, I32, V, V, V, V],
           "pl_ddm_set_preconditioner": [V, I32, V, V], "pl_ddm_set_geometry": [V, V], "pl_ddm_update_matrices": [V, I32, V, V], "pl_destroy": [V], "pl_set_bc": [V, V, V, V], "pl_set_periodic": [V, V],
           "pl_update_radii": [V, V], "pl_set_multiplicity": [V, V], "pl_update_segments": [V, V, V], "pl_assemble": [V],
           "pl_assemble_bsr": [V, I32, V, V], "pl_get_bsr": [V, V, V, V], "pl_spmv": [V, V, V],
           "pl_spmv_free": [V, V, V], "pl_spmv_bsr": [V, V, V], "pl_solve": [V, D, I32, V, V],
           "pl_reactions": [V, V, V], "pl_sens": [V, V, V, V], "pl_energy": [V, V, V], "pl_node_mod": [V, V, V],
           "pl_schur": [V, V, I32, D, I32, V], "pl_get_records": [V, V], "pl_time_kernel": [V, I32, I32, V],
           "pl_algorithmic_bytes": [V, V], "pl_forget_history": [V], "pl_debug_spd_solve": [I32, I32, V, V, V, V, I32], "pl_dist_unique_id_bytes": [], "pl_dist_unique_id": [V], "pl_dist_loopback_id": [V], "pl_dist_abort": [V],
           "pl_dist_init": [V, I32, I32, V, V, V, I32, I32], "pl_dist_set_peers": [V, V],
           "pl_generate_lattice": [I64, V, V, V, I32, I32, V, V, V, V], "pl_lattice_fetch": [V] * 12,
           "pl_lattice_free": [V], "pl_penalize": [I64, V, V, V, D, V, V, V],
           "pl_boundary_index": [I64, V, V, I64, V, V, V, V, V, V],
           "pl_boundary_index_rows": [I64, V, V, I64, V, V, V, V, V, V]}
    for name, args in sig.items():
        getattr(lib, name).argtypes = args
    # ABI handshake (include/pylattice_hip.h): this binding's struct layouts must be the library's
    if lib.pl_opts_size() != C.sizeof(PlOpts) or lib.pl_stats_size() != C.sizeof(PlStats):
        raise ImportError(f"{p}: pl_opts_t / pl_stats_t are {lib.pl_opts_size()} / {lib.pl_stats_size()} bytes in the "
                          f"library, {C.sizeof(PlOpts)} / {C.sizeof(PlStats)} in this binding (ABI version "
                          f"{lib.pl_abi_version()}): rebuild libpylattice_hip.so")
    _lib = lib
    return lib


def default_opts(lib=None) -> "PlOpts":
    """A pl_opts_t stamped by pl_default_opts (struct size + ABI version checked by the library)."""
    lib = lib or load_library()
    opts = PlOpts()
    _check(lib, lib.pl_default_opts(C.byref(opts), C.sizeof(PlOpts)))
    return opts


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} values, got {a.size}")
    return a


def _check(lib, rc, allow=()):
    if rc != PL_OK and rc not in allow:
        raise PlError(rc, lib.pl_last_error().decode())
    return rc


def lzone(node_xyz, beam_conn, beam_radius, device=0):
    """(B, 2) joint-penalisation lengths of a non-periodic lattice on the device (pl_lzone)."""
    lib = load_library()
    xyz = _f64(np.asarray(node_xyz).reshape(-1))
    conn = np.ascontiguousarray(beam_conn, dtype=np.int32).reshape(-1)
    rad = _f64(np.asarray(beam_radius).reshape(-1), len(conn) // 2)
    out = np.empty(len(conn), np.float64)
    _check(lib, lib.pl_lzone(int(device), len(xyz) // 3, len(conn) // 2, _ptr(xyz), _ptr(conn), _ptr(rad), _ptr(out)))
    return out.reshape(-1, 2)


def debug_spd_solve(A, b, device=0, fp32_factor=False):
    """Device dense SPD solve (test hook of the coarse solver): returns (x, b^T A^-1 b).  fp32_factor stores the
    inverse Cholesky factor in fp32, as the preconditioner does."""
    lib = load_library()
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.empty_like(b)
    q = C.c_double()
    _check(lib, lib.pl_debug_spd_solve(device, len(b), _ptr(A), _ptr(b), _ptr(x), C.byref(q), int(bool(fp32_factor))))
    return x, q.value


class PlLatticeInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int64), ("n_beams", C.c_int64), ("n_cell_beam", C.c_int64), ("n_cell_node", C.c_int64),
                ("n_created", C.c_int64)]


def generate_lattice(cell_coord, cell_size, cell_radii, tmpl, tmpl_type, want_created=False):
    """pl_generate_lattice: the multi-threaded host generator.  Returns a dict of arrays, or None when the library
    declines (lattice too irregular for its node table) and the numpy path has to be taken."""
    lib = load_library()
    cc, cs = _f64(cell_coord).reshape(-1, 3), _f64(cell_size).reshape(-1, 3)
    cr = _f64(cell_radii).reshape(len(cc), -1)
    tm = _f64(tmpl).reshape(-1, 6)
    tt = np.ascontiguousarray(tmpl_type, dtype=np.int32)
    h = C.c_void_p()
    info = PlLatticeInfo()
    rc = lib.pl_generate_lattice(len(cc), _ptr(cc), _ptr(cs), _ptr(cr), cr.shape[1], len(tm), _ptr(tm), _ptr(tt),
                                 C.byref(h), C.byref(info))
    if rc == PL_ERR_STATE:
        return None
    if rc != PL_OK:
        raise PlError(rc, "pl_generate_lattice: bad argument")
    try:
        Cn = len(cc)
        out = {"node_xyz": np.empty((info.n_nodes, 3)), "beam_conn": np.empty((info.n_beams, 2), np.int32),
               "beam_radius": np.empty(info.n_beams), "beam_type": np.empty(info.n_beams, np.int32),
               "beam_cell0": np.empty(info.n_beams, np.int32), "cell_beam_ptr": np.empty(Cn + 1, np.int64),
               "cell_beam_idx": np.empty(info.n_cell_beam, np.int64), "cell_node_ptr": np.empty(Cn + 1, np.int64),
               "cell_node_idx": np.empty(info.n_cell_node, np.int64)}
        if want_created:
            out["pid"] = np.empty((Cn, len(tm), 2), np.int32)
            out["bid"] = np.empty(Cn * len(tm), np.int32)
        order = ["node_xyz", "beam_conn", "beam_radius", "beam_type", "beam_cell0", "cell_beam_ptr", "cell_beam_idx",
                 "cell_node_ptr", "cell_node_idx", "pid", "bid"]
        lib.pl_lattice_fetch(h, *[_ptr(out.get(k)) for k in order])
    finally:
        lib.pl_lattice_free(h)
    return out


def penalize_arrays(node_xyz, beam_conn, lzone, mesh_size):
    """pl_penalize: (seg_len (B,3), seg_nsub (B,3) i32, pen_xyz (B,2,3))."""
    lib = load_library()
    xyz = _f64(node_xyz).reshape(-1, 3)
    conn = np.ascontiguousarray(beam_conn, dtype=np.int32).reshape(-1, 2)
    lz = None if lzone is None else _f64(lzone, 2 * len(conn))
    B = len(conn)
    seg_len, seg_nsub, pen = np.empty((B, 3)), np.empty((B, 3), np.int32), np.empty((B, 2, 3))
    _check(lib, lib.pl_penalize(B, _ptr(xyz), _ptr(conn), _ptr(lz), float(mesh_size), _ptr(seg_len), _ptr(seg_nsub),
                                _ptr(pen)))
    return seg_len, seg_nsub, pen


def boundary_index(cell_node_ptr, cell_node_idx, node_xyz, cell_coord, cell_size, by_coordinates=False):
    """pl_boundary_index: (index_boundary (N,) int64 with -1 off the cell boxes, nodes in visit order).
    by_coordinates: pl_boundary_index_rows - rows of a cell visited in rounded-coordinate order (reference_compat rows)."""
    lib = load_library()
    ptr = np.ascontiguousarray(cell_node_ptr, dtype=np.int64)
    idx = np.ascontiguousarray(cell_node_idx, dtype=np.int64)
    xyz = _f64(node_xyz).reshape(-1, 3)
    cc, cs = _f64(cell_coord).reshape(-1, 3), _f64(cell_size).reshape(-1, 3)
    ib = np.empty(len(xyz), np.int64)
    visit = np.empty(len(xyz), np.int64)
    nv = C.c_int64()
    fn = lib.pl_boundary_index_rows if by_coordinates else lib.pl_boundary_index
    _check(lib, fn(len(cc), _ptr(ptr), _ptr(idx), len(xyz), _ptr(xyz), _ptr(cc), _ptr(cs), _ptr(ib), _ptr(visit), C.byref(nv)))
    return ib, visit[:nv.value].copy()


class HipLattice:
    """Owner of one device handle: the condensed lattice operator + its PCG on one MI355X."""

    def __init__(self, node_xyz, beam_conn, beam_radius, seg_len, seg_nsub, young, poisson, kappa=0.9,
                 pen_coef=1.5, device=0, spmv_kernel=0, reorder=1, check_every=0, lanes_per_node=0, tile_nodes=0, precond=1,
                 coarse_max_dofs=0, grid=None, palette=0, local_max_dofs=0, precision=0, compact_records=0, condense=0, chol_persistent=0,
                 cg_form=0, tile_modes=0, coarse_modes=0, overlap=0, coarse_storage=0, warm_start=0, beam_mult=None, short_iteration=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.node_xyz = _f64(node_xyz).reshape(-1, 3)
        self.beam_conn = np.ascontiguousarray(beam_conn, dtype=np.int32).reshape(-1, 2)
        self.n_nodes, self.n_beams = len(self.node_xyz), len(self.beam_conn)
        self._radius = _f64(beam_radius, self.n_beams)
        self._seg_len = _f64(seg_len, 3 * self.n_beams)
        self._seg_nsub = np.ascontiguousarray(seg_nsub, dtype=np.int32).reshape(-1)
        mesh = PlMesh(self.n_nodes, self.n_beams, _ptr(self.node_xyz), _ptr(self.beam_conn), _ptr(self._radius),
                      _ptr(self._seg_len), _ptr(self._seg_nsub))
        opts = default_opts(self._lib)
        opts.young, opts.poisson, opts.kappa, opts.pen_coef = young, poisson, kappa, pen_coef
        opts.device, opts.spmv_kernel, opts.reorder, opts.check_every = device, spmv_kernel, reorder, check_every
        opts.lanes_per_node = lanes_per_node
        opts.tile_nodes = tile_nodes
        opts.precond, opts.coarse_max_dofs = precond, coarse_max_dofs
        opts.local_max_dofs = local_max_dofs
        opts.palette = palette
        opts.compact_records = compact_records
        opts.condense = condense
        opts.chol_persistent = chol_persistent
        opts.cg_form = cg_form
        opts.tile_modes = tile_modes
        opts.coarse_modes = coarse_modes
        opts.overlap = overlap
        opts.coarse_storage = coarse_storage
        opts.warm_start = warm_start
        opts.short_iteration = short_iteration
        opts.precision = precision            # 0 fp64, 1 fp32 inner PCG + fp64 refinement, 2 fp32 p / K*p only
        if grid is not None:                       # (lo[3], hi[3], n_nodes) of the whole lattice (multi-GPU)
            lo, hi, nn = grid
            for k in range(3):
                opts.grid_lo[k], opts.grid_hi[k] = float(lo[k]), float(hi[k])
            opts.grid_nodes = int(nn)
        _check(self._lib, self._lib.pl_create(C.byref(mesh), C.byref(opts), C.byref(self._h)))
        self.last_stats = None
        self._mult = None
        if beam_mult is not None:
            self.set_multiplicity(beam_mult)

    @classmethod
    def ddm(cls, n_nodes, cell_nodes, S, cell_S, device=0, alpha_max=100.0, check_every=0, precond=0, mintol=0.0,
            restart_every=0, node_xyz=None, coarse_max_dofs=0):
        """Handle for the domain-decomposition operator sum_c B^T S B (pl_create_ddm).  precond = 0: plain CG as the
        reference's default; 1: Jacobi on the assembled diagonal; 2: the reference's factorised assembled matrix
        (of the operator's own cell matrices unless ``set_ddm_preconditioner`` installs others); 3: its 6 x 6 node blocks;
        4: node blocks + a dense level of 12 modes per aggregate of nodes (needs ``node_xyz``, pl_ddm_set_geometry).
        check_every = 0: the host looks at the residual history every few dozen iterations; the device applies the
        reference's stopping rules itself and freezes the iterate at the iteration that meets them (k_pcg_direction), so the
        result is the reference's iterate whatever the host had queued (until round 5 the default was a look - and a drained
        stream - after every iteration)."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = C.c_void_p()
        cn = np.ascontiguousarray(cell_nodes, dtype=np.int32)
        Sm = np.ascontiguousarray(S, dtype=np.float64)
        if Sm.ndim == 2:
            Sm = Sm[None]
        cs = np.ascontiguousarray(cell_S, dtype=np.int32)
        self.n_nodes, self.n_beams = int(n_nodes), 0
        opts = default_opts(self._lib)
        opts.device, opts.alpha_max, opts.check_every = device, alpha_max, check_every
        opts.precond = precond
        opts.coarse_max_dofs = int(coarse_max_dofs)
        opts.mintol, opts.restart_every = float(mintol), int(restart_every)   # conjugate_gradient_solver.py:96-109
        if precond == 4 and node_xyz is None:
            raise ValueError("precond = 4 on a DDM handle needs node_xyz (the modes of its dense level)")
        _check(self._lib, self._lib.pl_create_ddm(self.n_nodes, cn.shape[0], cn.shape[1], _ptr(cn), Sm.shape[0],
                                                  _ptr(Sm), _ptr(cs), C.byref(opts), C.byref(self._h)))
        self.last_stats = None
        self._n_cells, self._m = cn.shape[0], 6 * cn.shape[1]
        if node_xyz is not None:
            self.set_ddm_geometry(node_xyz)
        return self

    def update_ddm_matrices(self, S, cell_S):
        """New cell matrices on this DDM handle (pl_ddm_update_matrices): same cells and nodes, other S_c; assemble() again
        before the next solve."""
        Sm = np.ascontiguousarray(S, dtype=np.float64)
        if Sm.ndim == 2:
            Sm = Sm[None]
        if Sm.shape[1:] != (self._m, self._m):
            raise ValueError(f"cell matrices must be {self._m} x {self._m}, got {Sm.shape[1:]}")
        cs = np.ascontiguousarray(cell_S, dtype=np.int32)
        if cs.shape != (self._n_cells,):
            raise ValueError("one matrix index per cell expected")
        _check(self._lib, self._lib.pl_ddm_update_matrices(self._h, Sm.shape[0], _ptr(Sm), _ptr(cs)))

    def set_ddm_geometry(self, node_xyz):
        """Node positions of a DDM handle (pl_ddm_set_geometry): aggregates and modes of the dense level of precond = 4."""
        xyz = np.ascontiguousarray(node_xyz, dtype=np.float64)
        if xyz.shape != (self.n_nodes, 3):
            raise ValueError(f"node_xyz must be ({self.n_nodes}, 3), got {xyz.shape}")
        _check(self._lib, self._lib.pl_ddm_set_geometry(self._h, _ptr(xyz)))

    def set_ddm_preconditioner(self, S=None, cell_S=None):
        """Cell matrices of the assembled-Schur preconditioner (pl_ddm_set_preconditioner); None = the operator's."""
        if S is None:
            _check(self._lib, self._lib.pl_ddm_set_preconditioner(self._h, 0, None, None))
            return
        Sm = np.ascontiguousarray(S, dtype=np.float64)
        if Sm.ndim == 2:
            Sm = Sm[None]
        if Sm.shape[1:] != (self._m, self._m):
            raise ValueError(f"preconditioner matrices must be {self._m} x {self._m}, got {Sm.shape[1:]}")
        cs = (np.zeros(self._n_cells, np.int32) if cell_S is None else np.ascontiguousarray(cell_S, dtype=np.int32))
        if cs.shape != (self._n_cells,):
            raise ValueError("one preconditioner matrix index per cell expected")
        _check(self._lib, self._lib.pl_ddm_set_preconditioner(self._h, Sm.shape[0], _ptr(Sm), _ptr(cs)))

    # -- lifetime ---------------------------------------------------------------------------------------
    def _changing(self, why):
        cb = getattr(self, "_before_change", None)
        if cb is not None:
            self._before_change = None
            cb(why)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._changing("handle closed")
            self._lib.pl_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- data -------------------------------------------------------------------------------------------
    def set_bc(self, fixed, ubar=None, f=None):
        n6 = 6 * self.n_nodes
        fx = np.ascontiguousarray(np.asarray(fixed).reshape(-1) != 0, dtype=np.uint8)
        if fx.size != n6:
            raise ValueError("fixed must have 6*n_nodes entries")
        ub = None if ubar is None else _f64(np.asarray(ubar).reshape(-1), n6)
        ff = None if f is None else _f64(np.asarray(f).reshape(-1), n6)
        _check(self._lib, self._lib.pl_set_bc(self._h, _ptr(fx), _ptr(ub), _ptr(ff)))

    def update_radii(self, radius):
        self._changing("update_radii")
        self._radius = _f64(radius, self.n_beams)
        _check(self._lib, self._lib.pl_update_radii(self._h, _ptr(self._radius)))

    def set_multiplicity(self, beam_mult):
        """Strut b counts as beam_mult[b] identical struts in parallel (pl_set_multiplicity); None = 1 everywhere."""
        self._changing("set_multiplicity")
        self._mult = None if beam_mult is None else _f64(beam_mult, self.n_beams)
        _check(self._lib, self._lib.pl_set_multiplicity(self._h, _ptr(self._mult)))

    def update_segments(self, seg_len, seg_nsub):
        self._changing("update_segments")
        self._seg_len = _f64(seg_len, 3 * self.n_beams)
        self._seg_nsub = np.ascontiguousarray(seg_nsub, dtype=np.int32).reshape(-1)
        _check(self._lib, self._lib.pl_update_segments(self._h, _ptr(self._seg_len), _ptr(self._seg_nsub)))

    def assemble(self):
        _check(self._lib, self._lib.pl_assemble(self._h))

    def assemble_bsr(self, with_bc=False):
        nr, nb = C.c_int64(), C.c_int64()
        _check(self._lib, self._lib.pl_assemble_bsr(self._h, int(bool(with_bc)), C.byref(nr), C.byref(nb)))
        return nr.value, nb.value

    def get_bsr(self):
        nr, nb = self.n_nodes, self.n_nodes + 2 * self.n_beams
        rowptr = np.empty(nr + 1, np.int64)
        col = np.empty(nb, np.int32)
        vals = np.empty((nb, 6, 6), np.float64)
        _check(self._lib, self._lib.pl_get_bsr(self._h, _ptr(rowptr), _ptr(col), _ptr(vals)))
        return rowptr, col, vals

    def records(self):
        rec = np.empty((self.n_beams, 8), np.float64)
        _check(self._lib, self._lib.pl_get_records(self._h, _ptr(rec)))
        return rec

    # -- operator ---------------------------------------------------------------------------------------
    def _vec_op(self, fn, x):
        x = _f64(np.asarray(x).reshape(-1), 6 * self.n_nodes)
        y = np.empty_like(x)
        _check(self._lib, fn(self._h, _ptr(x), _ptr(y)))
        return y.reshape(self.n_nodes, 6)

    def spmv(self, x):
        return self._vec_op(self._lib.pl_spmv, x)

    def spmv_free(self, x):
        return self._vec_op(self._lib.pl_spmv_free, x)

    def spmv_bsr(self, x):
        return self._vec_op(self._lib.pl_spmv_bsr, x)

    def reactions(self, u):
        return self._vec_op(self._lib.pl_reactions, u)

    def solve(self, rtol=1e-8, max_iter=20000, raise_on_noconv=True, download=True):
        """PCG solve; returns (u[N,6], stats).  download=False leaves u on the device and returns stats only."""
        u = np.empty(6 * self.n_nodes, np.float64) if download else None
        st = PlStats()
        st.struct_size = C.sizeof(PlStats)
        rc = self._lib.pl_solve(self._h, float(rtol), int(max_iter), _ptr(u), C.byref(st))
        self.last_stats = {k: getattr(st, k) for k, _ in PlStats._fields_ if k not in ("reserved", "reserved_i", "struct_size")}
        _timing.device("pl_solve: PCG (HIP events)", st.ms_solve)
        if st.ms_assembly > 0.0:
            _timing.device("pl_assemble of this solve (HIP events)", st.ms_assembly)
        _check(self._lib, rc, allow=() if raise_on_noconv else (PL_ERR_NOCONV,))
        if not download:
            return self.last_stats
        return u.reshape(self.n_nodes, 6), self.last_stats

    def sens(self, u, lam=None):
        """Per-strut sensitivities lam^T (dK_e/dr) u (pl_sens).  u = None: the solution of the last solve() of this handle,
        still on the device (no upload)."""
        u = None if u is None else _f64(np.asarray(u).reshape(-1), 6 * self.n_nodes)
        lam_a = None if lam is None else _f64(np.asarray(lam).reshape(-1), 6 * self.n_nodes)
        out = np.empty(self.n_beams, np.float64)
        _check(self._lib, self._lib.pl_sens(self._h, _ptr(u), _ptr(lam_a), _ptr(out)))
        return out

    def node_mod(self, u):
        """(B, 2, 6) displacements of the two penalisation points of every strut (pl_node_mod)."""
        u = _f64(np.asarray(u).reshape(-1), 6 * self.n_nodes)
        out = np.empty((self.n_beams, 2, 6), np.float64)
        _check(self._lib, self._lib.pl_node_mod(self._h, _ptr(u), _ptr(out)))
        return out

    def energy(self, u):
        u = _f64(np.asarray(u).reshape(-1), 6 * self.n_nodes)
        e = C.c_double()
        _check(self._lib, self._lib.pl_energy(self._h, _ptr(u), C.byref(e)))
        return e.value

    def schur(self, boundary_nodes, rtol=1e-12, max_iter=20000):
        bn = np.ascontiguousarray(boundary_nodes, dtype=np.int32)
        S = np.empty((6 * len(bn), 6 * len(bn)), np.float64)
        _check(self._lib, self._lib.pl_schur(self._h, _ptr(bn), len(bn), float(rtol), int(max_iter), _ptr(S)))
        return S

    # -- measurement ------------------------------------------------------------------------------------
    def time_kernel(self, which, reps=20):
        ms = C.c_double()
        _check(self._lib, self._lib.pl_time_kernel(self._h, int(which), int(reps), C.byref(ms)))
        return ms.value

    def algorithmic_bytes(self):
        out = (C.c_double * 3)()
        _check(self._lib, self._lib.pl_algorithmic_bytes(self._h, out))
        return {"spmv": out[0], "pcg_iter": out[1], "bsr": out[2]}

    def set_periodic(self, master):
        """Periodic constraints: master[i] = node whose six dofs node i shares (pl_set_periodic); None removes them."""
        m = None if master is None else np.ascontiguousarray(master, dtype=np.int32)
        if m is not None and m.size != self.n_nodes:
            raise ValueError(f"expected {self.n_nodes} entries, got {m.size}")
        _check(self._lib, self._lib.pl_set_periodic(self._h, _ptr(m)))

    def forget_history(self):
        """Drop the previous solve's iteration count (first look at the residual history) and the warm-start solution."""
        _check(self._lib, self._lib.pl_forget_history(self._h))

    # -- multi-GPU --------------------------------------------------------------------------------------
    @staticmethod
    def dist_unique_id() -> bytes:
        lib = load_library()
        n = lib.pl_dist_unique_id_bytes()
        buf = C.create_string_buffer(n)
        _check(lib, lib.pl_dist_unique_id(buf))
        return buf.raw

    @staticmethod
    def dist_loopback_id() -> bytes:
        """Id of a new in-process loopback group (pl_dist_loopback_id): `world` handles on one device, one host thread
        per rank - see ``pylatticedso_amd.loopback.LoopbackGroup``."""
        lib = load_library()
        buf = C.create_string_buffer(lib.pl_dist_unique_id_bytes())
        _check(lib, lib.pl_dist_loopback_id(buf))
        return buf.raw

    def dist_abort(self):
        """Break this handle's loopback group (ranks waiting at a collective return an error at once)."""
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pl_dist_abort(self._h)

    def dist_init(self, rank, world, unique_id: bytes, shared_local, shared_global, n_shared_global, shared_peer=None):
        """Attach the handle to the RCCL communicator; ``shared_peer`` (rank on the other side of every shared entry)
        switches the interface rows from the all-planes all-reduce to the neighbour exchange (pl_dist_set_peers)."""
        sl = np.ascontiguousarray(shared_local, dtype=np.int32)
        sg = np.ascontiguousarray(shared_global, dtype=np.int32)
        buf = C.create_string_buffer(unique_id, len(unique_id))
        _check(self._lib, self._lib.pl_dist_init(self._h, int(rank), int(world), buf, _ptr(sl), _ptr(sg),
                                                 int(len(sl)), int(n_shared_global)))
        if shared_peer is not None:
            sp = np.ascontiguousarray(shared_peer, dtype=np.int32)
            if sp.shape != sl.shape:
                raise ValueError("one peer rank per shared entry expected")
            _check(self._lib, self._lib.pl_dist_set_peers(self._h, _ptr(sp)))


# the reference wraps every hot-path method in @timing.category(..) @timing.timeit; here the
# C-ABI calls are the hot path: host wall clock per call, plus the device's own HIP-event times (see solve)
for _name in ("assemble", "assemble_bsr", "get_bsr", "solve", "set_bc", "spmv", "spmv_free", "spmv_bsr", "reactions",
              "sens", "energy", "schur", "update_radii", "update_segments", "records"):
    _f = getattr(HipLattice, _name, None)
    if _f is not None:
        _f._timing_category = "hip"
        setattr(HipLattice, _name, _timing.timeit(_f))
