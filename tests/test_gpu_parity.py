"""GPU parity tests: every call goes through the C ABI (ctypes) of libpylattice_hip.so on a real MI355X and is
compared with the CPU oracle / the reference's golden data.  Floating-point path: tolerances are written at each
assert; BASELINE.json's bar is 1e-6 relative L2 on displacements."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import c_oracle, timoshenko_oracle as O   # noqa: E402
from pylatticedso_amd import _capi, lattice_arrays as LA   # noqa: E402
from pylatticedso_amd.lattice_sim import LatticeSim     # noqa: E402
from pylatticedso_amd.utils_schur import get_schur_complement  # noqa: E402
from pylatticedso_amd.utils_simulation import solve_FEM_FenicsX  # noqa: E402

E, NU = 1013.0, 0.3
KERNELS = [1, 2, 3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"lattice_{name}.npz"))
    return g, LatticeSim(json.loads(str(g["preset_json"])))


def _device(L, **kw):
    lat, pen = L.lattice, L.penalized
    return _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub,
                            L.young_modulus, L.poisson_ratio, **kw)


def _oracle_scalars(L):
    lat, pen = L.lattice, L.penalized
    return np.array([O.condensed_beam(r, l, n, E, NU) for r, l, n in zip(lat.beam_radius, pen.seg_len, pen.seg_nsub)])


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


# every unit cell of the reference sees the ORACLE's numbers directly (round-3 verdict: 14 of the 18 reached it only through
# the gather kernel), plus the hybrids, the graded / sized lattices and the random-radius ones of round 4
ORACLE_DIRECT = sorted(set(
    ["bcc_2x2x2", "bccoctet_2x2x2", "octet_3x2x2_size", "bcc_3x2x2_gradradius", "hybrid4_1x1x1_periodic", "bcchybrid1_2x2x2",
     "bcchybrid4_1x1x1_periodic", "bcchybrid1hybrid4_3x2x1_size", "cubichybrid1_2x2x2", "random_bccoctet_2x2x3_hybrid",
     "random_octet_3x3x2_erased", "auxetic_2x2x2", "bccz_2x2x2", "cubic_2x2x2", "diamond_2x2x2", "hybrid1_1x1x1_periodic",
     "hybrid2_2x2x2", "hybrid3_2x2x2", "hybrid5_2x2x2", "kelvin_2x2x2", "octahedron_2x2x2", "octahedronyz_2x2x2",
     "octahedronz_2x2x2", "octet_2x2x2", "octetext_2x2x2", "original2_2x2x2", "original_2x2x2"]))


@pytest.mark.parametrize("name", ORACLE_DIRECT)
def test_records_match_oracle(golden_dir, name):
    _, L = _sim(golden_dir, name)
    with _device(L) as dev:
        dev.assemble()
        rec = dev.records()
    sc = _oracle_scalars(L)
    lat = L.lattice
    d = lat.node_xyz[lat.beam_conn[:, 1]] - lat.node_xyz[lat.beam_conn[:, 0]]
    L2 = (d * d).sum(1)
    ref = np.c_[sc[:, 2], sc[:, 4], (sc[:, 0] - sc[:, 2]) / L2, sc[:, 3] / np.sqrt(L2), (sc[:, 1] - sc[:, 4]) / L2, d]
    assert np.allclose(rec, ref, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("reorder", [0, 1])
@pytest.mark.parametrize("name", ["bcc_2x2x2", "bccoctet_2x2x2", "octet_3x2x2_size", "bcc_1x1x1_periodic",
                                  "bcchybrid1_2x2x2", "bcchybrid1hybrid4_3x2x1_size"])
def test_spmv_matches_oracle(golden_dir, name, kernel, reorder):
    _spmv_against_oracle(golden_dir, name, kernel, reorder)


@pytest.mark.parametrize("name", [n for n in ORACLE_DIRECT if n not in ("bcc_2x2x2", "bccoctet_2x2x2", "octet_3x2x2_size",
                                                                       "bcchybrid1_2x2x2", "bcchybrid1hybrid4_3x2x1_size")])
def test_spmv_matches_oracle_on_every_unit_cell(golden_dir, name):
    """The default kernel choice (LDS-tile K*p on the brick order) against the oracle's assembled K on the remaining unit
    cells and lattices: 1e-13 on K x, the masked operator and the energy."""
    _spmv_against_oracle(golden_dir, name, 0, 1)


def _spmv_against_oracle(golden_dir, name, kernel, reorder):
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    rng = np.random.default_rng(3)
    x = rng.standard_normal(6 * lat.n_nodes)
    with _device(L, spmv_kernel=kernel, reorder=reorder) as dev:
        dev.assemble()
        y = dev.spmv(x).ravel()
        assert _rel(y, K @ x) < 1e-13
        # masked operator P K P
        fixed = rng.random((lat.n_nodes, 6)) < 0.2
        dev.set_bc(fixed)
        yf = dev.spmv_free(x).ravel()
        m = (~fixed).ravel().astype(float)
        assert _rel(yf, m * (K @ (m * x))) < 1e-13
        assert abs(dev.energy(x) - 0.5 * x @ (K @ x)) < 1e-12 * abs(x @ (K @ x))


@pytest.mark.parametrize("name", ["bcc_2x2x2", "bccoctet_2x2x2", "octet_3x2x2_size", "bcchybrid1hybrid4_3x2x1_size",
                                  "bcc_4x4x4"])
def test_lds_resident_spmv_matches_oracle(golden_dir, name):
    """palette = 1 on a periodic lattice takes the LDS-resident tile kernel (k_spmv_tile_lds: own rows of x and the record
    palette in LDS, one 32-bit word per strut visit): plain and masked product against the oracle's K, small tiles so that
    the crossing visits (one end in another tile) are exercised too; the same through the gather kernel (PL_TILE_LDS=0
    is read once per process, hence a child process)."""
    import subprocess
    import sys
    import tempfile
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    rng = np.random.default_rng(5)
    x = rng.standard_normal(6 * lat.n_nodes)
    fixed = rng.random((lat.n_nodes, 6)) < 0.2
    m = (~fixed).ravel().astype(float)
    for tile_nodes in (0, 16):
        with _device(L, spmv_kernel=3, palette=1, tile_nodes=tile_nodes) as dev:
            dev.assemble()
            assert _rel(dev.spmv(x).ravel(), K @ x) < 1e-11        # (palette records are compared on 40 mantissa bits)
            dev.set_bc(fixed)
            assert _rel(dev.spmv_free(x).ravel(), m * (K @ (m * x))) < 1e-11
            y_lds = dev.spmv(x).ravel()
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "x.npy"), x)
        code = (f"import sys, numpy as np; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
                f"import test_gpu_parity as T\n"
                f"_, L = T._sim({str(golden_dir)!r}, {name!r})\n"
                f"x = np.load({os.path.join(tmp, 'x.npy')!r})\n"
                f"with T._device(L, spmv_kernel=3, palette=1, tile_nodes=16) as dev:\n"
                f"    dev.assemble(); np.save({os.path.join(tmp, 'y.npy')!r}, dev.spmv(x).ravel())\n")
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, PL_TILE_LDS="0"), timeout=300)
        assert _rel(np.load(os.path.join(tmp, "y.npy")), y_lds) < 1e-13


ALL_CELLS = ["auxetic_2x2x2", "bcc_2x2x2", "bccz_2x2x2", "cubic_2x2x2", "diamond_2x2x2", "hybrid1_1x1x1_periodic",
             "hybrid2_2x2x2", "hybrid3_2x2x2", "hybrid4_1x1x1_periodic", "hybrid5_2x2x2", "kelvin_2x2x2", "octahedron_2x2x2",
             "octahedronyz_2x2x2", "octahedronz_2x2x2", "octet_2x2x2", "octetext_2x2x2", "original2_2x2x2", "original_2x2x2"]


@pytest.mark.parametrize("name", ALL_CELLS)
def test_tile_kernels_on_every_unit_cell(golden_dir, name):
    """All 18 unit cells of the reference (reference-dumped 2 x 2 x 2 / periodic 1 x 1 x 1 lattices), small tiles: the
    LDS-resident K*p in its palette form (palette = 1) and in its streaming form (palette = 0: records streamed, strut
    vectors from the LDS direction table) against the per-node gather kernel (spmv_kernel = 2, none of the tile machinery),
    plain and masked, and through a multi-level solve."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    rng = np.random.default_rng(21)
    x = rng.standard_normal(6 * lat.n_nodes)
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == lat.node_xyz[:, 0].min()] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == lat.node_xyz[:, 0].max(), 2] = -1e-3
    with _device(L, spmv_kernel=2) as dev:
        dev.assemble()
        y_ref = dev.spmv(x).ravel()
        dev.set_bc(fixed, None, f)
        yf_ref = dev.spmv_free(x).ravel()
        dev.assemble()
        u_ref, st = dev.solve(rtol=1e-11, max_iter=100000)
        assert st["converged"] == 1
    for palette in (1, 0):
        with _device(L, spmv_kernel=3, palette=palette, tile_nodes=16, precond=3) as dev:
            dev.assemble()
            assert _rel(dev.spmv(x).ravel(), y_ref) < 1e-11
            dev.set_bc(fixed, None, f)
            assert _rel(dev.spmv_free(x).ravel(), yf_ref) < 1e-11
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=100000)
            assert st["converged"] == 1 and _rel(u.ravel(), u_ref.ravel()) < 1e-7


def test_irregular_lattice_takes_the_gather_kernel(golden_dir):
    """Node coordinates jittered at random: every strut has its own end-to-end vector and its own record, so neither the
    direction palette (<= 256 distinct vectors, compared bit for bit) nor the LDS record palette applies and the tile K*p
    falls back to the kernel that gathers x rows and records from global memory - same oracle parity, with and without the
    palette attempt, and through a solve."""
    _, L = _sim(golden_dir, "bcc_4x4x4")
    lat, pen = L.lattice, L.penalized
    rng = np.random.default_rng(11)
    xyz = lat.node_xyz + 1e-3 * rng.standard_normal(lat.node_xyz.shape)
    sc = _oracle_scalars(L)
    K = O.assemble_condensed(xyz, lat.beam_conn, sc)
    x = rng.standard_normal(6 * lat.n_nodes)
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == 4.0, 2] = -1e-3
    free = np.flatnonzero(fixed.ravel() == 0)
    import scipy.sparse.linalg as spla
    u_ref = np.zeros(6 * lat.n_nodes)
    u_ref[free] = spla.spsolve(K.tocsr()[free][:, free].tocsc(), f.ravel()[free])
    for palette in (0, 1):
        with _capi.HipLattice(xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, spmv_kernel=3,
                              palette=palette, precond=3, tile_nodes=32) as dev:
            dev.assemble()
            assert _rel(dev.spmv(x).ravel(), K @ x) < 1e-11
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-10, max_iter=5000)
            assert st["converged"] == 1 and _rel(u.ravel(), u_ref) < 1e-7


@pytest.mark.parametrize("geom,condense,precision", [("Octet", -1, 0), ("BCC", 1, 0), ("Octet", -1, 1), ("BCC", 1, 1)])
def test_warm_start_of_a_design_loop(geom, condense, precision):
    """opts.warm_start = 1: the second solve of a handle starts from the first one's solution.  On a system changed by a few
    per cent (radii) it converges to the same displacements as a cold start (1e-7) in fewer iterations; on the identical
    system it needs none to speak of; a changed Dirichlet set is honoured (the old solution is masked); with and without node
    elimination; in fp64 and with the fp32 inner solver (precision = 1: the refinement starts at the previous solution)."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 12
    lat = LA.generate((1, 1, 1), (n, n, n), [geom], [0.05 if geom == "BCC" else 0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == float(n), 2] = -1e-3
    rng = np.random.default_rng(2)
    r2 = lat.beam_radius * (1.0 + 0.03 * rng.standard_normal(lat.n_beams))
    res = {}
    for warm in (0, 1):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                              palette=1, tile_nodes=64, coarse_max_dofs=600, condense=condense, warm_start=warm,
                              precision=precision) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u1, st1 = dev.solve(rtol=1e-9, max_iter=20000)
            assert st1["precision_used"] == precision
            u1b, st1b = dev.solve(rtol=1e-9, max_iter=20000)          # the identical system again
            dev.update_radii(r2)
            dev.assemble()
            u2, st2 = dev.solve(rtol=1e-9, max_iter=20000)
            fixed2 = fixed.copy()
            fixed2[lat.node_xyz[:, 0] == float(n), 0] = 1                  # loaded face may no longer move in x
            dev.set_bc(fixed2, None, f)
            dev.assemble()
            u3, st3 = dev.solve(rtol=1e-9, max_iter=20000)
            assert all(s["converged"] == 1 for s in (st1, st1b, st2, st3))
            assert np.all(u3[fixed2 != 0] == 0.0)
            # zero right-hand side on a handle that holds a previous solution: the answer is zero, not that solution
            # (round-3 advisor finding: the early return used to leave the warm start in x)
            dev.set_bc(fixed2, None, np.zeros_like(f))
            u0, st0 = dev.solve(rtol=1e-9, max_iter=20000)
            assert st0["converged"] == 1 and st0["iterations"] == 0 and np.all(u0 == 0.0)
            res[warm] = (u1, u1b, u2, u3, st1["iterations"], st1b["iterations"], st2["iterations"], st3["iterations"])
    cold, warm = res[0], res[1]
    for k in range(4):
        assert _rel(warm[k], cold[k]) < 1e-7
    # the first solve of a handle has nothing to start from.  (Two handles assemble the dense level with atomics in a
    # different order: a residual that lands on the threshold can take one iteration more on one of them; precision = 1
    # counts whole check intervals of the inner solves, whose lengths follow the observed decay.)
    assert abs(warm[4] - cold[4]) <= (1 if precision == 0 else 0.2 * cold[4]), (warm[4:], cold[4:])
    assert warm[5] <= 2                            # identical system: the previous solution IS the solution
    if precision == 0:
        assert cold[5] == cold[4]                  # (same handle, same assembled operator: the same count)
    # perturbed radii: fewer iterations (the fp32 inner solver cuts what is left into stages of equal depth: a start two
    # decades nearer shortens every stage a little instead of saving a whole one)
    assert warm[6] < (0.9 if precision == 0 else 1.0) * cold[6], (warm[4:], cold[4:])
    if condense == 1:
        assert cold[4] > 0


def test_spmv_edge_cases(golden_dir):
    _, L = _sim(golden_dir, "bcc_2x2x2")
    lat = L.lattice
    with _device(L) as dev:
        dev.assemble()
        assert np.all(dev.spmv(np.zeros(6 * lat.n_nodes)) == 0.0)
        # rigid-body motions are in the null space of the unconstrained operator
        for om in np.eye(3):
            x = np.c_[np.cross(om, lat.node_xyz), np.tile(om, (lat.n_nodes, 1))]
            y = dev.spmv(x)
            assert np.abs(y).max() < 1e-9
        t = np.c_[np.ones((lat.n_nodes, 3)), np.zeros((lat.n_nodes, 3))]
        assert np.abs(dev.spmv(t)).max() < 1e-10
        # symmetry and linearity
        rng = np.random.default_rng(0)
        a, b = rng.standard_normal((2, 6 * lat.n_nodes))
        Ka, Kb = dev.spmv(a).ravel(), dev.spmv(b).ravel()
        assert abs(b @ Ka - a @ Kb) < 1e-11 * abs(b @ Ka)
        assert _rel(dev.spmv(2.0 * a - 3.0 * b).ravel(), 2.0 * Ka - 3.0 * Kb) < 1e-13


@pytest.mark.parametrize("name", ["bcc_2x2x2", "bccoctet_2x2x2"])
def test_bsr_assembly_matches_oracle(golden_dir, name):
    import scipy.sparse as sp
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    with _device(L) as dev:
        dev.assemble()
        nr, nb = dev.assemble_bsr(False)
        assert (nr, nb) == (lat.n_nodes, lat.n_nodes + 2 * lat.n_beams)
        rowptr, col, vals = dev.get_bsr()
        A = sp.bsr_matrix((vals, col, rowptr), shape=K.shape).tocsr()
        A.sum_duplicates()
        assert abs(A - K).max() < 1e-11 * abs(K).max()
        x = np.random.default_rng(5).standard_normal(K.shape[0])
        assert _rel(dev.spmv_bsr(x).ravel(), K @ x) < 1e-13
        # dolfinx Dirichlet treatment: constrained rows/cols zero, unit diagonal
        dev.set_bc(L.fixed_DOF)
        dev.assemble_bsr(True)
        rowptr, col, vals = dev.get_bsr()
        A = sp.bsr_matrix((vals, col, rowptr), shape=K.shape).tocsr()
        fx = L.fixed_DOF.ravel()
        Kref = K.tolil()
        Kref[fx, :] = 0.0
        Kref[:, fx] = 0.0
        Kref[np.flatnonzero(fx), np.flatnonzero(fx)] = 1.0
        assert abs(A - Kref.tocsr()).max() < 1e-11 * abs(K).max()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("name", ["bcc_2x2x2", "bcc_4x4x4", "bcc_6x3x3_flexion", "bcc_3x2x2_gradradius",
                                  "bcchybrid1_2x2x2", "bcchybrid4_2x2x2", "bcchybrid1hybrid4_3x2x1_size"])
def test_solve_matches_reference_faithful_direct_solve(golden_dir, name, kernel):
    """GPU PCG on condensed struts vs sparse-direct solve of the reference-faithful sub-meshed model built from
    the segments / BCs the reference itself produced (golden state).  Bar: 1e-6 relative L2 (we get ~1e-9)."""
    g, L = _sim(golden_dir, name)
    L._device = _device(L, spmv_kernel=kernel)
    xsol, model = solve_FEM_FenicsX(L)
    keep = ~g["beam_dup"]
    h = 0.05 * L.cell_size_x                                  # latticeGeneration.find_mesh_size (lattice_generation.py:50-60)
    K, nv = O.assemble_submeshed(g["node_xyz"], g["beam_conn"][keep], g["beam_radius"][keep], E, NU, h)
    n0, N = len(g["node_xyz"]), L.lattice.n_nodes
    fixed = np.zeros((nv, 6), bool)
    ubar = np.zeros((nv, 6))
    ff = np.zeros((nv, 6))
    fixed[:n0] = g["node_fixed"] != 0
    ubar[:n0] = g["node_ubar"]
    ff[:n0, :3] = g["node_force"][:, :3]
    uall = O.solve_dirichlet(K, fixed, ubar, ff).reshape(-1, 6)
    assert _rel(model.u, uall[:N]) < 1e-7
    assert model.stats["converged"] == 1
    # xsol = free dofs of cell-boundary nodes in the reference's order
    free = ~L.fixed_DOF
    expect = np.concatenate([uall[n][free[n]] for n in L._boundary_visit_order])
    assert _rel(xsol, expect) < 1e-7
    # penalisation points (node_mod): FE vertices in the reference, recovered here in closed form along every condensed
    # strut (pl_node_mod) - same node numbering as the reference, same displacements and rotations
    assert len(L.nodes) == n0
    umod = np.array([p.displacement_vector for p in L.nodes[N:]])
    assert umod.shape == (n0 - N, 6) and _rel(umod, uall[N:n0]) < 1e-7
    assert np.allclose(model.domain.geometry.x, g["node_xyz"]) and len(model.domain.topology.cells) == keep.sum()
    # reactions: R = K u on constrained nodes, times the reference's per-cell accumulation
    Rref = (K @ uall.ravel()).reshape(-1, 6)[:N]
    nodes = L.fixed_DOF.any(axis=1)
    mult = np.bincount(L.lattice.cell_node_idx, minlength=N)
    assert _rel(L.reaction_force_vector[nodes], mult[nodes, None] * Rref[nodes]) < 1e-6
    L._device.close()


@pytest.mark.parametrize("geom,name", [("BCC", "bcc"), ("Hybrid1", "hybrid1"), ("Hybrid4", "hybrid4")])
def test_schur_complement_matches_dolfinx_golden(golden_dir, geom, name):
    """get_schur_complement on the GPU vs the reference's committed dolfinx/PETSc Schur complements."""
    sg = np.load(os.path.join(golden_dir, f"schur_{geom}.npz"))
    for r, G in list(zip(sg["radius_values"].ravel(), sg["schur_matrices"]))[::2]:
        preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                               "radii": [float(r)], "geom_types": [geom]},
                  # the BCC dataset was generated with joint penalisation, Hybrid1/Hybrid4 without
                  "simulation_parameters": {"enable": geom == "BCC", "material": "VeroClear", "periodicity": True}}
        L = LatticeSim(preset)
        S = get_schur_complement(L)
        L._device.close()
        assert S.shape == G.shape
        assert _rel(S, G) < 1e-8, (geom, r)


@pytest.mark.parametrize("name", ["bcc_2x2x2", "bcchybrid1_2x2x2", "bcc_3x2x2_gradradius"])
def test_schur_complement_of_one_cell_of_a_multi_cell_lattice(golden_dir, name):
    """get_schur_complement(lattice, cell_index) (utils_schur.py:22-41 with BeamModel(..., cell_index)): the cell's own
    struts and nodes - penalised as they are in the WHOLE lattice - condensed on its boundary nodes.  Held to the oracle:
    assembly of exactly those struts (closed-form condensation of their penalised, sub-meshed segments, which
    tests/test_oracle_golden.py holds to the assembled sub-mesh) and a dense static condensation.  1e-8 relative
    Frobenius."""
    from pylatticedso_amd.utils_schur import node_order_to_simulate
    _, L = _sim(golden_dir, name)
    lat, pen = L.lattice, L.penalized
    with pytest.raises(ValueError):
        get_schur_complement(L)                                   # several cells and no cell_index: as the reference
    for cell in (0, lat.n_cells - 1):
        S = get_schur_complement(L, cell)
        struts = np.unique(lat.cell_beam_idx[lat.cell_beam_ptr[cell]:lat.cell_beam_ptr[cell + 1]])
        nodes = np.unique(lat.beam_conn[struts])
        loc = np.full(lat.n_nodes, -1)
        loc[nodes] = np.arange(len(nodes))
        sc = np.array([O.condensed_beam(r, l, n, E, NU) for r, l, n in
                       zip(lat.beam_radius[struts], pen.seg_len[struts], pen.seg_nsub[struts])])
        K = O.assemble_condensed(lat.node_xyz[nodes], loc[lat.beam_conn[struts]], sc)
        order = loc[node_order_to_simulate(L, cell)]
        bd = (6 * order[:, None] + np.arange(6)).ravel()
        Sref = O.schur_complement(K, bd)
        assert S.shape == Sref.shape == (6 * len(order),) * 2
        assert _rel(S, Sref) < 1e-8, (name, cell)
        assert _rel(S, S.T) < 1e-9


def test_schur_dataset_construction_flow(golden_dir, tmp_path, monkeypatch):
    """The reference's construct_schur_complement_dataset.py flow on the GPU: one LatticeSim, reset_cell_with_new_radii
    per sample, get_schur_complement, save / load of the dataset - against the reference's dolfinx dataset."""
    from pylatticedso_amd import utils_schur as US
    sg = np.load(os.path.join(golden_dir, "schur_BCC.npz"))
    preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                           "radii": [0.02], "geom_types": ["BCC"]},
              "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": True}}
    L = LatticeSim(preset)
    radii, mats = [], []
    for r in sg["radius_values"].ravel():
        L.reset_cell_with_new_radii([float(r)])
        assert np.allclose(L.lattice.beam_radius, r)
        mats.append(get_schur_complement(L))
        radii.append([float(r)])
    for S, G in zip(mats, sg["schur_matrices"]):
        assert _rel(S, G) < 1e-8
    monkeypatch.setattr(US, "define_path_schur_complement", lambda lat: str(tmp_path / "Schur_complement_BCC.npz"))
    US.save_schur_complement_npz(L, radii, mats)
    back = US.load_schur_complement_dataset(L)
    assert len(back) == len(radii) and _rel(back[(float(radii[2][0]),)], mats[2]) == 0.0
    with pytest.raises(ValueError):
        L.reset_cell_with_new_radii([0.05, 0.05])


def test_sensitivity_matches_finite_difference(golden_dir):
    _, L = _sim(golden_dir, "bcc_2x2x2")
    lat, pen = L.lattice, L.penalized
    rng = np.random.default_rng(2)
    u, lam = rng.standard_normal((2, lat.n_nodes, 6))
    with _device(L) as dev:
        dev.assemble()
        s_uu = dev.sens(u)
        s_lu = dev.sens(u, lam)

    def quad(radius, a, b):
        sc = np.array([O.condensed_beam(r, l, n, E, NU) for r, l, n in zip(radius, pen.seg_len, pen.seg_nsub)])
        K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, sc)
        return a.ravel() @ (K @ b.ravel())

    for b in [0, 7, lat.n_beams - 1]:
        h = 1e-6 * lat.beam_radius[b]
        rp, rm = lat.beam_radius.copy(), lat.beam_radius.copy()
        rp[b] += h
        rm[b] -= h
        assert abs(s_uu[b] - (quad(rp, u, u) - quad(rm, u, u)) / (2 * h)) < 1e-6 * abs(s_uu[b])
        assert abs(s_lu[b] - (quad(rp, lam, u) - quad(rm, lam, u)) / (2 * h)) < 1e-6 * abs(s_lu[b])


def test_update_radii_reuses_topology(golden_dir):
    _, L = _sim(golden_dir, "bcc_2x2x2")
    lat = L.lattice
    x = np.random.default_rng(1).standard_normal(6 * lat.n_nodes)
    with _device(L) as dev:
        dev.assemble()
        y1 = dev.spmv(x)
        dev.update_radii(lat.beam_radius * 1.3)
        with pytest.raises(_capi.PlError):
            dev.spmv(x)                       # must re-assemble first
        dev.assemble()
        y2 = dev.spmv(x)
        assert not np.allclose(y1, y2)
        dev.update_radii(lat.beam_radius)
        dev.assemble()
        assert np.array_equal(dev.spmv(x), y1)


def test_error_paths(golden_dir):
    _, L = _sim(golden_dir, "bcc_2x2x2")
    lat, pen = L.lattice, L.penalized
    bad = lat.beam_conn.copy()
    bad[0, 0] = lat.n_nodes + 5
    with pytest.raises(_capi.PlError) as e:
        _capi.HipLattice(lat.node_xyz, bad, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
    assert e.value.code == _capi.PL_ERR_ARG
    with pytest.raises(_capi.PlError):
        _capi.HipLattice(lat.node_xyz, lat.beam_conn, -lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
    with _device(L) as dev:
        with pytest.raises(_capi.PlError) as e:
            dev.solve()
        assert e.value.code == _capi.PL_ERR_STATE
        dev.assemble()
        with pytest.raises(_capi.PlError):
            dev.solve()                                        # no BCs yet
        dev.set_bc(L.fixed_DOF, L.displacement_vector, L.applied_force)
        u, st = dev.solve(rtol=1e-12, max_iter=3, raise_on_noconv=False)
        assert st["converged"] == 0 and st["iterations"] == 3
        with pytest.raises(_capi.PlError) as e:
            dev.solve(rtol=1e-12, max_iter=3)
        assert e.value.code == _capi.PL_ERR_NOCONV
        # zero load, zero prescribed displacement -> zero solution, converged in 0 iterations
        dev.set_bc(L.fixed_DOF)
        u, st = dev.solve()
        assert np.all(u == 0.0) and st["iterations"] == 0 and st["converged"] == 1


def test_ddm_golden_cross_check(golden_dir):
    """Reference DDM solve (RBF Schur surrogate, CG tol 1e-6) vs the GPU FEM solve of the same cantilever.  The
    reference's own two solvers differ at this level: its BCC Schur dataset was generated on a PERIODIC single cell
    (every corner joint penalised), whereas its FEM path leaves the single-strut corners of the free surfaces
    un-penalised, and the surrogate/CG tolerances are 1e-6.  Hence a loose 5e-3 bar (measured 2.3e-3); this is a
    plausibility cross-check, the tight parity tests are the ones above."""
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x4x4.npz"))
    preset = json.loads(str(g["preset_json"]))
    preset["simulation_parameters"].pop("DDM")
    L = LatticeSim(preset)
    xsol, model = solve_FEM_FenicsX(L)
    L._device.close()
    assert len(xsol) == len(g["xsol"])
    assert _rel(xsol, g["xsol"]) < 5e-3


def test_neighbour_exchange_plumbing_with_a_self_peer(golden_dir):
    """pl_dist_set_peers on the one GPU of the test box: a single-rank communicator whose only "neighbour" is the rank
    itself, so every grouped ncclSend / ncclRecv pair delivers the rank's own packed rows and the shared rows of K*x
    come back doubled - which pins the pack -> send/recv -> add chain (order by global interface id, fp64 and fp32
    row types) without a second GPU.  The two-rank arithmetic itself is covered on CPU by tests/test_partition_gloo.py."""
    _, L = _sim(golden_dir, "bcc_4x4x4")
    lat = L.lattice
    x = np.random.default_rng(1).standard_normal(6 * lat.n_nodes)
    shared = np.flatnonzero(lat.node_xyz[:, 1] == 2.0)
    rng = np.random.default_rng(5)
    gid = rng.permutation(len(shared))                           # global ids in an order unrelated to the local one
    with _device(L) as ref:
        ref.assemble()
        y0 = ref.spmv(x)
    with _device(L) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, gid, len(shared),
                      shared_peer=np.zeros(len(shared), np.int32))
        dev.assemble()
        y1 = dev.spmv(x)
    expect = y0.copy()
    expect[shared] *= 2.0
    assert _rel(y1, expect) < 1e-14
    with _device(L) as dev:
        with pytest.raises(_capi.PlError):
            dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, gid, len(shared),
                          shared_peer=np.full(len(shared), 3, np.int32))          # peer outside the communicator


def test_rccl_path_with_single_rank_communicator(golden_dir):
    """The multi-GPU code path (RCCL communicator, interface pack / all-reduce / unpack, weighted dot products)
    driven with world = 1 on the one GPU of the test box: results must equal the plain path."""
    _, L = _sim(golden_dir, "bcc_4x4x4")
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    with _device(L) as ref:
        ref.set_bc(L.fixed_DOF, None, f)
        ref.assemble()
        u0, st0 = ref.solve(rtol=1e-11)
    shared = np.flatnonzero(lat.node_xyz[:, 1] == 2.0)          # pretend the plane y = 2 is a slab interface
    with _device(L) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
        dev.set_bc(L.fixed_DOF, None, f)
        dev.assemble()
        u1, st1 = dev.solve(rtol=1e-11)
        x = np.random.default_rng(0).standard_normal(6 * lat.n_nodes)
        y1 = dev.spmv(x)
    # same with the two-level preconditioner and an explicit global grid (what bench.py passes for N > 1)
    grid = (lat.bbox[0::2], lat.bbox[1::2], lat.n_nodes)
    with _device(L, precond=2, tile_nodes=32, coarse_max_dofs=600, grid=grid) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
        dev.set_bc(L.fixed_DOF, None, f)
        dev.assemble()
        u2, st2 = dev.solve(rtol=1e-11)
    # (on a 4 x 4 x 4 lattice the coarse space buys nothing - the iteration benefit is asserted on larger lattices in
    # test_tile_level_reduces_iterations_octet16 - here only the RCCL plumbing of the coarse level is checked)
    assert _rel(u2, u0) < 1e-8 and st2["converged"] == 1 and st2["iterations"] < 1.2 * st0["iterations"]
    # ... and with the tile level, which skips the tiles that hold shared nodes
    with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, grid=grid) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
        dev.set_bc(L.fixed_DOF, None, f)
        dev.assemble()
        u3, st3 = dev.solve(rtol=1e-11)
    assert _rel(u3, u0) < 1e-8 and st3["converged"] == 1
    # ... and in the single-reduction form (one all-reduce per iteration: opts.cg_form = 1), with the neighbour exchange
    # as well (the rank is its own neighbour: shared rows come back doubled, so the weights are 1/2 and the assembled
    # operator is the plain one only if every shared row is halved first - here only the all-reduce variant is compared)
    with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, grid=grid, cg_form=1) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
        dev.set_bc(L.fixed_DOF, None, f)
        dev.assemble()
        u5, st5 = dev.solve(rtol=1e-11)
    assert _rel(u5, u0) < 1e-8 and st5["converged"] == 1 and int(st5["cg_form_used"]) == 1
    assert st5["iterations"] <= st3["iterations"] + 2
    # ... and with the rank-local dense level under a coarser global one (precond = 4, what bench.py uses for N > 1):
    # shared nodes are left out of the local level, nothing of it is communicated
    with _device(L, precond=4, tile_nodes=32, coarse_max_dofs=100, local_max_dofs=600, grid=grid) as dev:
        dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
        dev.set_bc(L.fixed_DOF, None, f)
        dev.assemble()
        u4, st4 = dev.solve(rtol=1e-11)
    assert _rel(u4, u0) < 1e-8 and st4["converged"] == 1
    with _device(L) as ref:
        ref.assemble()
        y0 = ref.spmv(x)
    assert _rel(y1, y0) < 1e-14
    assert _rel(u1, u0) < 1e-9
    assert abs(st1["iterations"] - st0["iterations"]) <= 2


def test_full_size_properties_config2():
    """BASELINE.json configs[1] (50^3 Octet, 3.03 M struts): size-independent properties of the device operator
    and of the solve - symmetry, linearity, rigid-body null space, true residual, Clapeyron (f.u = u.K.u)."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 50
    lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    assert lat.n_beams == 3030000 and lat.n_nodes == 515151
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == float(n)
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    rng = np.random.default_rng(11)
    for kernel in (3, 2):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              spmv_kernel=kernel) as dev:
            dev.assemble()
            a, b = rng.standard_normal((2, 6 * lat.n_nodes))
            Ka, Kb = dev.spmv(a).ravel(), dev.spmv(b).ravel()
            assert abs(b @ Ka - a @ Kb) < 1e-10 * abs(b @ Ka)
            assert _rel(dev.spmv(a - 2.0 * b).ravel(), Ka - 2.0 * Kb) < 1e-12
            om = np.array([0.3, -0.2, 0.5])
            rigid = np.c_[np.cross(om, lat.node_xyz) + [1.0, 2.0, 3.0], np.tile(om, (lat.n_nodes, 1))]
            assert np.abs(dev.spmv(rigid)).max() < 1e-7 * np.abs(Ka).max()
            if kernel == 3:
                dev.set_bc(fixed, None, f)
                u, st = dev.solve(rtol=1e-8, max_iter=20000)
                assert st["converged"] == 1
                R = dev.spmv(u)
                res = np.where(fixed != 0, 0.0, f - R)
                assert np.linalg.norm(res) / np.linalg.norm(f) < 5e-8          # true residual, not the recurrence
                assert abs((f * u).sum() - 2.0 * dev.energy(u)) < 1e-7 * abs((f * u).sum())
                assert np.all(u[fixed != 0] == 0.0)
                u_ref = u
            else:
                dev.set_bc(fixed, None, f)
                u2, _ = dev.solve(rtol=1e-8, max_iter=20000)
                assert _rel(u2, u_ref) < 1e-6                                    # two independent kernels agree


def test_full_size_properties_config3():
    """BASELINE.json configs[2] (100^3 BCC, r = 0.05: 8 M struts, 2.03 M nodes) on one GPU with the bench settings
    (multi-level preconditioner, record palette): operator symmetry, rigid-body null space, true residual and
    Clapeyron's theorem for the cantilever solve."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 100
    lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    assert lat.n_beams == 8_000_000 and lat.n_nodes == 2_030_301
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == float(n)
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    rng = np.random.default_rng(12)
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                          precond=3, palette=1) as dev:
        dev.assemble()
        a, b = rng.standard_normal((2, 6 * lat.n_nodes))
        Ka, Kb = dev.spmv(a).ravel(), dev.spmv(b).ravel()
        assert abs(b @ Ka - a @ Kb) < 1e-10 * abs(b @ Ka)
        om = np.array([0.3, -0.2, 0.5])
        rigid = np.c_[np.cross(om, lat.node_xyz) + [1.0, 2.0, 3.0], np.tile(om, (lat.n_nodes, 1))]
        assert np.abs(dev.spmv(rigid)).max() < 1e-7 * np.abs(Ka).max()
        dev.set_bc(fixed, None, f)
        dev.assemble()                       # the coarse levels need the Dirichlet set
        u, st = dev.solve(rtol=1e-8, max_iter=50000)
        # BCC is bipartite: the automatic node elimination takes the 10^6 cell centres out (793 -> 503 iterations; 474 with
        # the strain modes of the tile level, 334 with those of the dense level: the round-1 verdict asked for <= 400)
        assert st["converged"] == 1 and int(st["precond_used"]) == 3 and st["iterations"] < 400
        assert int(st["condensed_nodes"]) == 1_000_000
        res = np.where(fixed != 0, 0.0, f - dev.spmv(u))
        assert np.linalg.norm(res) / np.linalg.norm(f) < 5e-8
        assert abs((f * u).sum() - 2.0 * dev.energy(u)) < 1e-7 * abs((f * u).sum())
        assert np.all(u[fixed != 0] == 0.0)


def test_full_size_properties_config5():
    """BASELINE.json configs[4] AS NAMED on one GPU: 200 x 200 x 50 cells of BCC + Octet (r = 0.04 / 0.03), 64.2 M struts,
    fp32 matrix-free PCG with fp64 refinement (opts.precision = 1): converged against the TRUE fp64 residual, Clapeyron's
    theorem, clamped face at rest, and the same solve in fp64 arithmetic and storage within 1e-6 (the bar of BASELINE.json)."""
    from pylatticedso_amd import lattice_arrays as LA
    lat = LA.generate((1, 1, 1), (200, 200, 50), ["BCC", "Octet"], [0.04, 0.03])
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
    assert lat.n_beams > 64_000_000
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == 200.0
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    sols = {}
    for precision in (1, 0):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=3, palette=1, precision=precision) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-8, max_iter=5000)
            assert st["converged"] == 1 and int(st["precision_used"]) == precision and st["iterations"] < 500
            assert st["rel_residual"] <= 1.0000001e-8
            if precision == 1:
                res = np.where(fixed != 0, 0.0, f - dev.spmv(u))
                assert np.linalg.norm(res) / np.linalg.norm(f) < 5e-8         # the residual the solver reports is the true one
                assert abs((f * u).sum() - 2.0 * dev.energy(u)) < 1e-6 * abs((f * u).sum())
            assert np.all(u[fixed != 0] == 0.0)
            sols[precision] = u
    assert _rel(sols[1], sols[0]) < 1e-6


def test_two_rank_workload_emulated_on_one_gpu():
    """The 2-GPU weak-scaling workload of bench.py (50 x 100 x 50 Octet, 6.05 M struts) solved on ONE GPU through the
    multi-rank code path (single-rank RCCL communicator) with the slab interface plane declared shared, so that the
    rank-local levels leave those nodes out exactly as two ranks would: same solution as the plain single-GPU solve,
    and the iteration count DESIGN.md section 8 quotes for N = 2 (191; the all-reduced dense level is coarser than at
    N = 1, 155)."""
    from pylatticedso_amd import lattice_arrays as LA
    lat = LA.generate((1, 1, 1), (50, 100, 50), ["Octet"], [0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == 50.0
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    shared = np.flatnonzero(lat.node_xyz[:, 1] == 50.0)
    assert len(shared) == 5101
    out = []
    for emulate in (False, True):
        # (what a rank of a multi-GPU run uses: rigid-body modes in the dense level, whose operator is summed over
        # ranks; the tile level keeps its strain modes, on the rank's own nodes)
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=3, palette=1, coarse_modes=6) as dev:
            if emulate:
                dev.dist_init(0, 1, _capi.HipLattice.dist_unique_id(), shared, np.arange(len(shared)), len(shared))
            dev.set_bc(fixed, None, f)
            dev.assemble()
            out.append(dev.solve(rtol=1e-8, max_iter=5000))
    (u0, s0), (u1, s1) = out
    assert s0["converged"] == 1 and s1["converged"] == 1
    assert _rel(u1, u0) < 1e-6
    print("iterations without / with the interface plane declared shared:", s0["iterations"], s1["iterations"])
    assert s0["iterations"] <= 170 and s1["iterations"] <= 175          # measured: 152 and 155


@pytest.mark.parametrize("name", ["bcc_4x4x4", "bcc_6x3x3_flexion"])
def test_multilevel_preconditioners_same_solution(golden_dir, name):
    """precond = 2 (Jacobi + rigid-body coarse space) must give the same displacements as Jacobi-PCG."""
    _, L = _sim(golden_dir, name)
    f = np.zeros((L.lattice.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    res = {}
    for pc in (1, 2, 3):
        with _device(L, precond=pc, tile_nodes=32, coarse_max_dofs=600) as dev:
            dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
            dev.assemble()
            res[pc] = dev.solve(rtol=1e-11)
    (u1, s1), (u2, s2), (u3, s3) = res[1], res[2], res[3]
    assert s2["converged"] == 1 and s3["converged"] == 1
    assert _rel(u2, u1) < 1e-8 and _rel(u3, u1) < 1e-8
    # lattices this small have next to nothing for a coarse space to capture (the gain is asserted on a 16^3 lattice
    # below); here the coarse levels must simply not hurt
    assert s2["iterations"] < 1.1 * s1["iterations"]


@pytest.mark.parametrize("name", ["bccoctet_2x2x2", "bcc_6x3x3_flexion", "octet_3x2x2_size",
                                  "bcchybrid1hybrid4_3x2x1_size"])
def test_strain_mode_levels_match_oracle(golden_dir, name):
    """The 12-mode block levels (opts.tile_modes = 12, opts.coarse_modes = 12: rigid + uniform strains per tile and per
    aggregate) forced on the small reference lattices, prescribed displacements and node elimination included: the
    displacements are the oracle's direct solve."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    for cond in (-1, 1):
        with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, condense=cond, tile_modes=12, coarse_modes=12) as dev:
            dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
            assert st["converged"] == 1 and int(st["precond_used"]) == 3
            assert _rel(u, uref) < 1e-8


def test_strain_modes_of_the_dense_level():
    """opts.coarse_modes = 12 (rigid + uniform strains per aggregate, fewer aggregates) against 6: same displacements on a
    bending-dominated lattice with its cell centres eliminated and on a stretch-dominated one."""
    from pylatticedso_amd import lattice_arrays as LA
    for geom, n, r in ((["BCC"], 20, [0.05]), (["Octet"], 20, [0.03])):
        lat = LA.generate((1, 1, 1), (n, n, n), geom, r)
        pen = LA.penalize(lat, LA.compute_lzone(lat))
        fixed = np.zeros((lat.n_nodes, 6), np.uint8)
        fixed[lat.node_xyz[:, 0] == 0.0] = 1
        tgt = lat.node_xyz[:, 0] == float(n)
        f = np.zeros((lat.n_nodes, 6))
        f[tgt, 2] = -0.1 / tgt.sum()
        out = {}
        for cm in (6, 12):
            with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                                  precond=3, coarse_modes=cm, coarse_max_dofs=768) as dev:
                dev.set_bc(fixed, None, f)
                dev.assemble()
                out[cm] = dev.solve(rtol=1e-10, max_iter=20000)
                assert out[cm][1]["converged"] == 1
                res = np.where(fixed != 0, 0.0, f - dev.spmv(out[cm][0]))
                assert np.linalg.norm(res) <= 2e-10 * np.linalg.norm(f)
        assert _rel(out[12][0], out[6][0]) < 1e-7
        assert out[12][1]["iterations"] < 1.05 * out[6][1]["iterations"]


def test_strain_modes_of_the_tile_level():
    """opts.tile_modes: 12 (rigid + uniform strains per tile, the default) against 6 (rigid only): same displacements,
    fewer iterations on a bending-dominated and on a stretch-dominated lattice."""
    from pylatticedso_amd import lattice_arrays as LA
    for geom, n, r in ((["BCC"], 16, [0.05]), (["Octet"], 16, [0.03])):
        lat = LA.generate((1, 1, 1), (n, n, n), geom, r)
        pen = LA.penalize(lat, LA.compute_lzone(lat))
        fixed = np.zeros((lat.n_nodes, 6), np.uint8)
        fixed[lat.node_xyz[:, 0] == 0.0] = 1
        tgt = lat.node_xyz[:, 0] == float(n)
        f = np.zeros((lat.n_nodes, 6))
        f[tgt, 2] = -0.1 / tgt.sum()
        out = {}
        for tm in (6, 12):
            with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                                  precond=3, condense=-1, tile_modes=tm) as dev:
                dev.set_bc(fixed, None, f)
                dev.assemble()
                out[tm] = dev.solve(rtol=1e-10, max_iter=20000)
                res = np.where(fixed != 0, 0.0, f - dev.spmv(out[tm][0]))
                assert np.linalg.norm(res) <= 2e-10 * np.linalg.norm(f)
        assert _rel(out[12][0], out[6][0]) < 1e-7
        assert out[12][1]["iterations"] < out[6][1]["iterations"]


def test_tile_level_reduces_iterations_octet16():
    """precond = 3 adds the tile level (rigid-body modes of every 256-node K*p tile, 6 x 6 block solves) to the
    two-level preconditioner: same displacements, fewer iterations once the tiles are large (with tiny tiles the
    tile spaces nearly repeat the Jacobi level and the additive sum can cost iterations - hence opt-in per case)."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 16
    lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == float(n)
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    out = {}
    for pc in (1, 2, 3):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=pc) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            out[pc] = dev.solve(rtol=1e-10, max_iter=20000)
    assert all(st["converged"] == 1 for _, st in out.values())
    assert _rel(out[2][0], out[1][0]) < 1e-7 and _rel(out[3][0], out[1][0]) < 1e-7
    assert out[3][1]["iterations"] < out[2][1]["iterations"] < out[1][1]["iterations"]
    # precond = 4: with a global level forced to be coarse (as it is on many GPUs), the rank-local dense level wins
    # iterations back
    res = {}
    for pc in (3, 4):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=pc, coarse_max_dofs=100, tile_modes=6) as dev:   # (precond 4 has 6-mode tiles)
            dev.set_bc(fixed, None, f)
            dev.assemble()
            res[pc] = dev.solve(rtol=1e-10, max_iter=20000)
    assert _rel(res[4][0], out[1][0]) < 1e-7 and res[4][1]["converged"] == 1
    # (a gain of up to 15 % when the bricks were cut independently of the aggregates; now that they nest it is a wash on
    # one GPU - the level exists for many-rank runs, where the all-reduced level has to coarsen)
    assert res[4][1]["iterations"] < 1.1 * res[3][1]["iterations"]


@pytest.mark.parametrize("name", ["bcc_4x4x4", "bcc_6x3x3_flexion", "bccoctet_2x2x2", "bcchybrid1hybrid4_3x2x1_size"])
def test_condensed_pcg_matches_oracle(golden_dir, name):
    """opts.condense = 1: an independent set of nodes is eliminated exactly inside the PCG (CG on the Schur complement
    of the others, two K*p passes per iteration); the solution - including the eliminated nodes - must be the oracle's,
    prescribed displacements and loads on eliminated nodes included."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    f[lat.n_nodes // 2, :3] += [1e-3, -2e-3, 5e-4]                 # a load on an interior node (condensed or not)
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    its = {}
    for cond in (-1, 1):
        with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, condense=cond) as dev:
            dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
            assert st["converged"] == 1 and _rel(u, uref) < 1e-8
            assert (st["condensed_nodes"] > 0) == (cond > 0)
            its[cond] = st["iterations"]
            if cond > 0:                                                  # a second right-hand side on the same handle
                dev.set_bc(L.fixed_DOF, None, 2.0 * f)
                u2, _ = dev.solve(rtol=1e-11, max_iter=20000)
                uref2 = O.solve_dirichlet(K, L.fixed_DOF, 0 * ubar, np.where(L.fixed_DOF, 0.0, 2.0 * f)).reshape(-1, 6)
                assert _rel(u2, uref2) < 1e-8
    assert its[1] <= its[-1] + 2          # (a few dozen iterations on these small lattices: +-1 is noise)


@pytest.mark.parametrize("name,graded", [("bcc_4x4x4", False), ("bcc_6x3x3_flexion", True), ("bccoctet_2x2x2", False),
                                         ("octet_3x2x2_size", True), ("bcchybrid1hybrid4_3x2x1_size", False)])
@pytest.mark.parametrize("condense", [-1, 1])
@pytest.mark.parametrize("modes", [(6, 6), (12, 6), (12, 12)])
def test_short_iteration_matches_ordinary_form_and_oracle(golden_dir, name, graded, condense, modes):
    """opts.short_iteration (pl_small.h): dense level through its explicit inverse inside the z-kernel, search direction formed
    in the K*p launch - 3 (4) launches per iteration instead of 5 (6).  Same preconditioner and recurrences: the oracle's
    solution, the ordinary form's iteration count (+-2: A_c^-1 is rounded to fp32 once more), with and without node
    elimination, palette and streaming records (graded radii), 6- and 12-mode levels, prescribed displacements, a second
    right-hand side and a warm start on the same handle."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    if graded:
        rng = np.random.default_rng(5)
        lat.beam_radius[:] = lat.beam_radius * (1.0 + 0.2 * rng.random(lat.n_beams))
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    f[lat.n_nodes // 2, :3] += [1e-3, -2e-3, 5e-4]
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    uref2 = O.solve_dirichlet(K, L.fixed_DOF, 0 * ubar, np.where(L.fixed_DOF, 0.0, 2.0 * f)).reshape(-1, 6)
    its = {}
    for short in (-1, 1):
        with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, condense=condense, palette=1, tile_modes=modes[0],
                     coarse_modes=modes[1], short_iteration=short, warm_start=1) as dev:
            dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
            assert st["converged"] == 1 and _rel(u, uref) < 1e-8
            assert int(st["short_iteration_used"]) == (1 if short == 1 else 0)
            assert (st["condensed_nodes"] > 0) == (condense > 0)
            # (graded radii: more than 256 distinct records - on a small lattice the LDS-resident streaming form, which the
            # short iteration needs, instead of the gather kernel with a large palette)
            assert int(st["kp_form"]) == ((2 if short == 1 else 3) if graded else 1)
            res = np.where(L.fixed_DOF, 0.0, f - dev.spmv(u))
            assert np.linalg.norm(res) < 1e-9 * np.linalg.norm(f)
            dev.set_bc(L.fixed_DOF, None, 2.0 * f)            # second right-hand side, warm start from the first solution
            u2, st2 = dev.solve(rtol=1e-11, max_iter=20000)
            assert st2["converged"] == 1 and _rel(u2, uref2) < 1e-8
            its[short] = (st["iterations"], st2["iterations"])
    assert abs(its[1][0] - its[-1][0]) <= 2 and abs(its[1][1] - its[-1][1]) <= 3, its


@pytest.mark.parametrize("name,graded", [("bcc_4x4x4", False), ("bcc_6x3x3_flexion", True), ("bccoctet_2x2x2", False),
                                         ("octet_3x2x2_size", True)])
@pytest.mark.parametrize("modes", [(6, 6), (12, 12)])
def test_persistent_pcg_matches_oracle(golden_dir, name, graded, modes):
    """opts.short_iteration = 2 (pl_persist.h, experimental): the whole PCG loop - single-reduction CG, multi-level
    preconditioner, no node elimination - as ONE persistent launch with hand-offs between the tiles' workgroups.  The oracle's
    solution, iteration counts near the multi-launch solver's, a second solve on the same handle."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    if graded:
        rng = np.random.default_rng(5)
        lat.beam_radius[:] = lat.beam_radius * (1.0 + 0.2 * rng.random(lat.n_beams))
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    f[lat.n_nodes // 2, :3] += [1e-3, -2e-3, 5e-4]
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    its = {}
    for short in (-1, 2):
        with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, condense=-1, palette=1, tile_modes=modes[0],
                     coarse_modes=modes[1], short_iteration=short) as dev:
            dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-10, max_iter=5000)
            assert st["converged"] == 1, st
            assert int(st["short_iteration_used"]) == (2 if short == 2 else 0)
            assert _rel(u, uref) < 1e-7
            u2, st2 = dev.solve(rtol=1e-10, max_iter=5000)
            assert st2["converged"] == 1 and _rel(u2, u) < 1e-8
            its[short] = st["iterations"]
    assert abs(its[2] - its[-1]) <= max(3, its[-1] // 10), its


def test_short_iteration_is_the_default_on_small_lattices_only():
    """Automatic choice (opts.short_iteration = 0): on where every tile can read its rows of A_c^-1 once per iteration
    (n_tiles x modes x dofs x 4 B <= 16 MB), off on the headline-sized lattice; never with the fp32 solver modes or the
    single-reduction form."""
    n = 10
    lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == float(n), 2] = -1e-3
    used, sols = {}, {}
    for key, kw in {"auto": {}, "off": {"short_iteration": -1}, "precision1": {"precision": 1}, "cg1": {"cg_form": 1}}.items():
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                              palette=1, **kw) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            sols[key], st = dev.solve(rtol=1e-10, max_iter=20000)
            assert st["converged"] == 1
            used[key] = (int(st["short_iteration_used"]), st["iterations"])
    assert used["auto"][0] == 1 and used["off"][0] == 0 and used["precision1"][0] == 0 and used["cg1"][0] == 0
    assert abs(used["auto"][1] - used["off"][1]) <= 2
    for key in ("off", "precision1", "cg1"):
        assert _rel(sols[key], sols["auto"]) < 1e-7


@pytest.mark.parametrize("name", ["bcc_2x2x2", "bcc_4x4x4", "bcc_6x3x3_flexion", "bccoctet_2x2x2", "octet_2x2x2_pull",
                                  "bcchybrid1hybrid4_3x2x1_size"])
def test_dense_factor_preconditioner_on_small_lattices(golden_dir, name):
    """opts.precond = 5 (what LatticeSim.device_model asks for below 400 nodes): the dense Cholesky factor of P K P as the
    preconditioner - the oracle's solution in one or two PCG steps, prescribed displacements and a second Dirichlet set on the
    same handle included; the explicit BSR matrix the caller asked for is left as asked."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    f[lat.n_nodes // 2, :3] += [1e-3, -2e-3, 5e-4]
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    with _device(L, precond=5) as dev:
        dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
        dev.assemble()
        dev.assemble_bsr(False)
        u, st = dev.solve(rtol=1e-10, max_iter=100)
        assert st["converged"] == 1 and int(st["precond_used"]) == 5 and st["iterations"] <= 3, st
        assert _rel(u, uref) < 1e-8
        rng = np.random.default_rng(3)
        xr = rng.standard_normal((lat.n_nodes, 6))
        assert _rel(dev.spmv_bsr(xr), dev.spmv(xr)) < 1e-12           # the caller's BSR (without bcs) is intact
        fixed2 = L.fixed_DOF.copy()
        fixed2[lat.n_nodes - 1] = True                                # a new Dirichlet set on the assembled handle
        dev.set_bc(fixed2, None, f)
        u2, st2 = dev.solve(rtol=1e-10, max_iter=100)
        uref2 = O.solve_dirichlet(K, fixed2, 0 * ubar, np.where(fixed2, 0.0, f)).reshape(-1, 6)
        assert int(st2["precond_used"]) == 5 and st2["iterations"] <= 3 and _rel(u2, uref2) < 1e-8
    # the drop-in layer picks it for small lattices
    xsol, model = solve_FEM_FenicsX(L)
    assert int(model.stats["precond_used"]) == 5 and model.stats["iterations"] <= 3
    L._device.close()


def test_dense_factor_preconditioner_falls_back_on_a_mechanism():
    """A lattice that is not held (no Dirichlet dof): P K P is singular, the factorisation reports it and the solve runs as
    Jacobi PCG (precond_used = 1) instead of failing."""
    lat = LA.generate((1, 1, 1), (2, 2, 2), ["BCC"], [0.05])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=5) as dev:
        f = np.zeros((lat.n_nodes, 6))
        f[0, 0], f[-1, 0] = 1e-3, -1e-3                               # self-equilibrated load: solvable by CG
        dev.set_bc(np.zeros((lat.n_nodes, 6), bool), None, f)
        dev.assemble()
        u, st = dev.solve(rtol=1e-8, max_iter=5000, raise_on_noconv=False)
        assert int(st["precond_used"]) in (1, 5)
        if int(st["precond_used"]) == 5:                              # (rounding made the factorisation go through)
            assert st["converged"] == 1


def test_bfloat16_storage_of_the_dense_level():
    """opts.coarse_storage = 16: the inverse factor of the dense level in bfloat16 (half the bytes of the two triangular
    GEMVs per iteration; automatic from 1 024 dofs).  Same solution to the solver tolerance - the preconditioner only has
    to be a fixed SPD operator - and no more than a few per cent more iterations than with fp32 storage; BCC with node
    elimination and Octet."""
    for geom, radius, n in (("Octet", 0.03, 14), ("BCC", 0.05, 16)):
        lat = LA.generate((1, 1, 1), (n, n, n), [geom], [radius])
        pen = LA.penalize(lat, LA.compute_lzone(lat))
        fixed = np.zeros((lat.n_nodes, 6), np.uint8)
        fixed[lat.node_xyz[:, 0] == 0.0] = 1
        tgt = lat.node_xyz[:, 0] == float(n)
        f = np.zeros((lat.n_nodes, 6))
        f[tgt, 2] = -0.1 / tgt.sum()
        out = {}
        for st in (32, 16):
            with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                                  tile_nodes=64, coarse_max_dofs=1536, coarse_storage=st) as dev:
                dev.set_bc(fixed, None, f)
                dev.assemble()
                out[st] = dev.solve(rtol=1e-10, max_iter=20000)
        assert out[16][1]["converged"] == 1 and _rel(out[16][0], out[32][0]) < 1e-7
        assert out[16][1]["iterations"] <= 1.05 * out[32][1]["iterations"] + 2, (geom, out[16][1]["iterations"],
                                                                                 out[32][1]["iterations"])
    with pytest.raises(_capi.PlError):
        _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                         coarse_storage=8)


@pytest.mark.parametrize("kernel", [1, 2])
def test_node_elimination_is_off_with_kernels_that_ignore_its_masks(golden_dir, kernel):
    """Round-2 advisor finding: with an explicit spmv_kernel = 1 / 2 the K*p kernels ignore the elimination masks, so the
    solve must not run the elimination prologue either (it rewrote r_v -= K_vc K_cc^-1 b_c and never back-substituted:
    silently wrong whenever an eliminated node carried load).  condense = 1 and a body load on every interior node."""
    _, L = _sim(golden_dir, "bcc_4x4x4")
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    inside = np.all((lat.node_xyz > 0.0) & (lat.node_xyz < 4.0), axis=1)
    f[inside, :3] += [1e-3, -2e-3, 5e-4]                 # the cell centres (what gets eliminated on BCC) carry load
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    uref = O.solve_dirichlet(K, L.fixed_DOF, 0.0 * f, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    with _device(L, spmv_kernel=kernel, precond=3, tile_nodes=32, coarse_max_dofs=600, condense=1) as dev:
        dev.set_bc(L.fixed_DOF, None, f)
        dev.assemble()
        assert dev.time_kernel(3, 2) > 0.0               # (the timed iteration is the one the solve runs: no elimination)
        u, st = dev.solve(rtol=1e-11, max_iter=20000)
        assert st["converged"] == 1 and st["condensed_nodes"] == 0
        assert _rel(u, uref) < 1e-8


@pytest.mark.parametrize("modes", [(6, 6), (12, 6), (12, 12)])
@pytest.mark.parametrize("name", ["bccoctet_2x2x2", "bcc_6x3x3_flexion", "octet_3x2x2_size",
                                  "bcchybrid1hybrid4_3x2x1_size"])
def test_single_reduction_pcg_matches_oracle(golden_dir, name, modes):
    """opts.cg_form = 1 (Chronopoulos-Gear recurrences, the dense level's residual carried by recurrence - one all-reduce
    per iteration on several GPUs): same solution as the oracle's direct solve and as the ordinary form, prescribed
    displacements included; the iteration count may differ by the one-iteration lag of the residual norm.  With rigid-body
    levels and (round 4) with the uniform-strain modes on the tile level alone and on both levels."""
    _, L = _sim(golden_dir, name)
    tile_modes, coarse_modes = modes
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    res = {}
    for form in (0, 1):
        with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, condense=-1, cg_form=form, tile_modes=tile_modes,
                     coarse_modes=coarse_modes) as dev:
            dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
            assert st["converged"] == 1 and int(st["cg_form_used"]) == form
            assert _rel(u, uref) < 1e-8
            res[form] = (u, st["iterations"])
            if form == 1:                                                # a second solve on the same handle
                dev.set_bc(L.fixed_DOF, None, 2.0 * f)
                u2, _ = dev.solve(rtol=1e-11, max_iter=20000)
                uref2 = O.solve_dirichlet(K, L.fixed_DOF, 0 * ubar, np.where(L.fixed_DOF, 0.0, 2.0 * f)).reshape(-1, 6)
                assert _rel(u2, uref2) < 1e-8
    assert _rel(res[1][0], res[0][0]) < 1e-8
    assert res[1][1] <= res[0][1] * 1.03 + 2


@pytest.mark.parametrize("tile_modes", [6, 12])
def test_single_reduction_pcg_at_scale(tile_modes):
    """24^3 Octet cantilever with the bench's solver settings: both CG forms reach the same displacements (1e-8) and
    the single-reduction form needs at most 3 % + 1 more iterations - with rigid-body levels
    and with the strain modes the ordinary form runs with by default (round 4: the single-reduction form has them too)."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 24
    lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == float(n)
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    out = {}
    for form in (0, 1):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=3, palette=1, cg_form=form, tile_modes=tile_modes) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-10, max_iter=5000)
            assert st["converged"] == 1 and int(st["cg_form_used"]) == form
            r = np.where(fixed, 0.0, f - dev.spmv(u))
            assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(f)
            out[form] = (u, st["iterations"])
    assert _rel(out[1][0], out[0][0]) < 1e-8
    assert out[1][1] <= out[0][1] * 1.03 + 1


@pytest.mark.parametrize("precision", [1, 2])
@pytest.mark.parametrize("name", ["bccoctet_2x2x2", "bcc_6x3x3_flexion", "bcchybrid1hybrid4_3x2x1_size"])
def test_fp32_solver_modes_match_oracle(golden_dir, name, precision):
    """opts.precision = 1 (fp32 inner PCG + fp64 refinement) and 2 (fp32 search direction and K*p, fp64 iterate and
    residual) against the CPU oracle (sparse-direct solve of the condensed model): BASELINE.json's bar, 1e-6 relative
    L2 on the displacements; rtol refers to the TRUE fp64 residual in these modes."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    ubar = np.where(L.fixed_DOF, L.displacement_vector, 0.0)
    uref = O.solve_dirichlet(K, L.fixed_DOF, ubar, np.where(L.fixed_DOF, 0.0, f)).reshape(-1, 6)
    with _device(L, precond=3, tile_nodes=32, coarse_max_dofs=600, precision=precision) as dev:
        dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
        dev.assemble()
        u, st = dev.solve(rtol=1e-9, max_iter=20000)
        assert st["converged"] == 1 and int(st["precision_used"]) == precision and st["restarts"] >= 1
        assert _rel(u, uref) < 1e-6
        res = np.where(L.fixed_DOF, 0.0, f - dev.spmv(u))
        b = np.where(L.fixed_DOF, 0.0, f - dev.spmv(ubar))
        assert np.linalg.norm(res) <= 2e-9 * np.linalg.norm(b)
    # a Jacobi handle ignores the request and says so
    with _device(L, precond=1, precision=precision) as dev:
        dev.set_bc(L.fixed_DOF, L.displacement_vector, f)
        dev.assemble()
        u1, st1 = dev.solve(rtol=1e-10, max_iter=20000)
        assert int(st1["precision_used"]) == 0 and _rel(u1, uref) < 1e-6


def test_fp32_solver_quarter_of_config5():
    """A quarter of BASELINE.json configs[4] (100 x 100 x 50 BCC + Octet, r = [0.04, 0.03], 16 M struts) with the fp32
    inner PCG + fp64 refinement and with the mixed mode: true residual, Clapeyron, and agreement with the fp64 solve."""
    from pylatticedso_amd import lattice_arrays as LA
    lat = LA.generate((1, 1, 1), (100, 100, 50), ["BCC", "Octet"], [0.04, 0.03])
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    tgt = lat.node_xyz[:, 0] == 100.0
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    out = {}
    for precision in (0, 1, 2):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=3, palette=1, precision=precision) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-8, max_iter=20000)
            assert st["converged"] == 1 and int(st["precision_used"]) == precision
            res = np.where(fixed != 0, 0.0, f - dev.spmv(u))
            assert np.linalg.norm(res) / np.linalg.norm(f) < 5e-8
            assert abs((f * u).sum() - 2.0 * dev.energy(u)) < 1e-7 * abs((f * u).sum())
            out[precision] = (u, st)
            print(f"precision {precision}: {st['iterations']} iterations, {st['restarts']:.0f} inner solves, "
                  f"{st['ms_solve']:.1f} ms, true rel. residual {np.linalg.norm(res) / np.linalg.norm(f):.2e}")
    assert _rel(out[1][0], out[0][0]) < 1e-6 and _rel(out[2][0], out[0][0]) < 1e-6


@pytest.mark.parametrize("n", [5, 64, 100, 700, 1000])      # (1000: band wider than the LDS ring - the MFMA walk)
def test_device_dense_spd_solver(n):
    """Blocked Cholesky + inverse factor on the device (coarse solver of the two-level preconditioner) vs numpy."""
    rng = np.random.default_rng(n)
    Q = rng.standard_normal((n, n))
    A = Q @ Q.T + n * np.eye(n)
    b = rng.standard_normal(n)
    x, quad = _capi.debug_spd_solve(A, b)
    ref = np.linalg.solve(A, b)
    assert _rel(x, ref) < 1e-11
    assert abs(quad - b @ ref) < 1e-11 * abs(b @ ref)
    # the storage the preconditioner uses: W = L^-1 rounded to fp32 (x = W32^T W32 b, fp64 accumulation)
    x32, quad32 = _capi.debug_spd_solve(A, b, fp32_factor=True)
    assert _rel(x32, ref) < 2e-6
    assert abs(quad32 - b @ ref) < 2e-6 * abs(b @ ref)
    with pytest.raises(_capi.PlError):
        _capi.debug_spd_solve(-A, b)


@pytest.mark.parametrize("name", ["bcc_4x4x4", "octet_3x2x2_size", "bcc_3x2x2_gradradius"])
def test_record_palette_matches_plain_records(golden_dir, name):
    """palette = 1 (2-byte ids into a table of distinct records, compared on 40 mantissa bits) vs per-strut records."""
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    x = np.random.default_rng(4).standard_normal(6 * lat.n_nodes)
    with _device(L, spmv_kernel=3) as dev:
        dev.assemble()
        y0 = dev.spmv(x)
    with _device(L, spmv_kernel=3, palette=1) as dev:
        dev.assemble()
        y1 = dev.spmv(x)
        dev.update_radii(lat.beam_radius * 1.1)      # palette is rebuilt with the records
        dev.assemble()
        y2 = dev.spmv(x)
    assert _rel(y1, y0) < 1e-11
    assert _rel(y2, y0) > 1e-3


@pytest.mark.parametrize("name", ["bcc_2x2x2", "bccoctet_2x2x2", "octet_3x2x2_size", "bcc_3x2x2_gradradius"])
def test_device_lzone_matches_reference_golden(golden_dir, name):
    """pl_lzone against the penalisation lengths the reference itself computed (golden base_beam_lzone), and against
    the numpy restatement."""
    g, L = _sim(golden_dir, name)
    lat = L.lattice
    dev = _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius)
    host = LA.compute_lzone(lat, False)
    assert dev.shape == host.shape == (lat.n_beams, 2)
    assert np.allclose(dev, host, rtol=1e-13, atol=1e-18)
    # reference: Lzone per base beam, matched through the end-point coordinates
    key = lambda a, b: tuple(np.round(np.r_[a, b], 9))
    ref = {key(g["base_node_xyz"][a], g["base_node_xyz"][b]): lz
           for (a, b), lz in zip(g["base_beam_conn"], g["base_beam_lzone"])}
    hit = 0
    for (a, b), lz in zip(lat.beam_conn, dev):
        k1, k2 = key(lat.node_xyz[a], lat.node_xyz[b]), key(lat.node_xyz[b], lat.node_xyz[a])
        if k1 in ref:
            assert np.allclose(lz, ref[k1], rtol=1e-12, atol=1e-15)
            hit += 1
        elif k2 in ref:
            assert np.allclose(lz[::-1], ref[k2], rtol=1e-12, atol=1e-15)
            hit += 1
    assert hit == lat.n_beams


def test_device_lzone_large_and_errors():
    lat = LA.generate((1, 1, 1), (24, 24, 24), ["BCC", "Octet"], [0.04, 0.03])
    import time
    t0 = time.perf_counter()
    host = LA.compute_lzone(lat, False)
    t1 = time.perf_counter()
    dev = _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius)
    t2 = time.perf_counter()
    assert np.allclose(dev, host, rtol=1e-13, atol=1e-18) and (dev >= 0).all()
    print(f"L_zone of {lat.n_beams} struts: numpy {t1 - t0:.2f} s, device (incl. transfers) {t2 - t1:.3f} s")
    # a strut alone at its nodes has no neighbour: 0; collinear continuation: 1e-7
    xyz = np.array([[0.0, 0, 0], [1, 0, 0], [2, 0, 0], [5, 5, 5], [6, 5, 5]])
    conn = np.array([[0, 1], [1, 2], [3, 4]], np.int32)
    lz = _capi.lzone(xyz, conn, np.array([0.1, 0.2, 0.3]))
    assert np.array_equal(lz, [[0.0, 1e-7], [1e-7, 0.0], [0.0, 0.0]])
    bad = conn.copy()
    bad[0, 0] = 9
    with pytest.raises(_capi.PlError):
        _capi.lzone(xyz, bad, np.array([0.1, 0.2, 0.3]))


@pytest.mark.parametrize("name", ["bcc_4x4x4", "octet_3x2x2_size", "bccoctet_2x2x2", "kelvin_2x2x2"])
def test_row_form_of_the_operator_matches_oracle(golden_dir, name, monkeypatch):
    """K*p by rows (pl_rows.h, opt-in with PL_ROWS=1: one lane per node, every strut evaluated once per end from a palette
    that holds both orientations, records through scalar registers, no atomics) against the ORACLE's assembled K - plain,
    masked with p.Ap, and through solves with and without node elimination (whose two passes are row selections there) -
    and bit for bit against itself (fixed summation order)."""
    monkeypatch.setenv("PL_ROWS", "1")
    _, L = _sim(golden_dir, name)
    lat = L.lattice
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, _oracle_scalars(L))
    rng = np.random.default_rng(17)
    x = rng.standard_normal(6 * lat.n_nodes)
    fixed = rng.random((lat.n_nodes, 6)) < 0.15
    m = (~fixed).ravel().astype(float)
    for tile_nodes in (0, 16):
        with _device(L, spmv_kernel=3, palette=1, tile_nodes=tile_nodes) as dev:
            dev.assemble()
            y = dev.spmv(x).ravel()
            assert _rel(y, K @ x) < 1e-11                     # (palette records are compared on 40 mantissa bits)
            assert np.array_equal(y, dev.spmv(x).ravel())       # no atomics: the same bits every time
            dev.set_bc(fixed)
            assert _rel(dev.spmv_free(x).ravel(), m * (K @ (m * x))) < 1e-11
    # solves: cantilever, Jacobi and multi-level, node elimination forced on and off
    fx = np.zeros((lat.n_nodes, 6), np.uint8)
    fx[lat.node_xyz[:, 0] == lat.node_xyz[:, 0].min()] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == lat.node_xyz[:, 0].max(), 2] = -1e-3
    u_ref = O.solve_dirichlet(K, fx.astype(bool), np.zeros_like(f), f).ravel()
    for kw in (dict(precond=1), dict(precond=3, tile_nodes=16, coarse_max_dofs=96, condense=1),
               dict(precond=3, tile_nodes=16, coarse_max_dofs=96, condense=-1)):
        with _device(L, palette=1, **kw) as dev:
            dev.set_bc(fx, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
            assert st["converged"] == 1 and _rel(u.ravel(), u_ref) < 1e-7, kw
    monkeypatch.setenv("PL_ROWS", "0")
    with _device(L, spmv_kernel=3, palette=1) as dev:           # the tile kernel on the same handle settings
        dev.assemble()
        assert _rel(dev.spmv(x).ravel(), y) < 1e-12


@pytest.mark.parametrize("env", [{}, {"PL_BSR_CUMASK": "0"}, {"PL_TRTRI_ROWS": "0"}, {"PL_TRTRI_ROWS": "1", "PL_BSR_CUMASK": "7"}])
def test_assembly_schedules_give_the_same_operator_and_solution(env, monkeypatch):
    """Round 4 moved two pieces of pl_assemble beside the dense level's factorisation chain: the explicit K (BSR) fill on a
    CU-masked stream (the part that fits; the rest behind the chain) and the inverse factor in row ranges on a second stream.
    With either switched off (the round-3 order), with single-row ranges, and with the fill squeezed onto an eighth of the chip,
    the assembled blocks are bit-identical and the solve takes the same path to the same displacements.  24^3 Octet: a dense
    level of several 64-dof blocks, so that there ARE links and ranges."""
    from pylatticedso_amd import lattice_arrays as LA
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 24
    lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == float(n), 2] = -1e-3
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                          palette=1) as dev:
        dev.set_bc(fixed, None, f)
        dev.assemble()
        dev.assemble_bsr(with_bc=True)                   # a fill of its own, on the main stream ...
        _, col, vals = dev.get_bsr()
        dev.assemble()                                   # ... and (the handle now wants K) inside pl_assemble, beside the chain
        _, col2, vals2 = dev.get_bsr()
        assert np.array_equal(vals, vals2) and np.array_equal(col, col2)
        m = 1.0 - fixed
        x = m * np.random.default_rng(3).standard_normal((lat.n_nodes, 6))
        assert _rel(m * dev.spmv_bsr(x), dev.spmv_free(x)) < 1e-12      # (K with boundary conditions: free rows / columns)
        u, st = dev.solve(rtol=1e-10, max_iter=5000)
        assert st["converged"] == 1
        r = np.where(fixed, 0.0, f - dev.spmv(u))
        assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(f)
        # the dense level is the same operator whatever the launch schedule: same iteration count (+- the one a last-bit
        # difference in the atomically assembled coarse operator can cost)
        assert abs(st["iterations"] - 133) <= 3, st["iterations"]


@pytest.mark.parametrize("cell,geom,condense", [(1.0, "Octet", 0), (0.5, "BCC", 1), (0.7, "Octet", 0)])
def test_float_lever_arms_of_the_vector_kernels_change_nothing(cell, geom, condense, monkeypatch):
    """Round 5: where every node's position relative to its aggregate's reference point is a float exactly (cell sizes 1, 1/2 ...:
    checked node by node at pl_create), k_pcg_update_tile / k_pcg_direction_* read it as 12 bytes per node instead of 24.  The
    numbers are the same, so the preconditioner is: same iteration count (+- what a last-bit difference of the atomically
    assembled coarse operator can cost) and the same displacements as with the table switched off (PL_NO_REL32); a cell size of
    0.7 has no exact table and takes the fp64 positions as before."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 14
    lat = LA.generate((cell, cell, cell), (n, n, n), [geom], [0.04 * cell])
    pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[np.isclose(lat.node_xyz[:, 0], n * cell), 2] = -1e-3
    out = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("PL_NO_REL32", "1")
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                              palette=1, condense=condense) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-10, max_iter=5000)
            assert st["converged"] == 1
            r = np.where(fixed, 0.0, f - dev.spmv(u))
            assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(f)
            out.append((u, st["iterations"]))
    assert abs(out[0][1] - out[1][1]) <= 3, (out[0][1], out[1][1])
    assert _rel(out[0][0].ravel(), out[1][0].ravel()) < 1e-8


@pytest.mark.parametrize("condense", [0, 1])
def test_extrapolated_warm_start_on_a_smooth_design_path(condense):
    """opts.warm_start = 2 / 3 (round 5): the solve starts from the linear / quadratic extrapolation of the handle's last two /
    three solutions.  Along a smooth path of radii (a design loop) every solve reaches the displacements of a cold start (1e-7)
    and the path as a whole takes fewer iterations than with warm_start = 1, which takes fewer than cold starts."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 12
    lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == float(n), 2] = -1e-3
    mid = 0.5 * (lat.node_xyz[lat.beam_conn[:, 0]] + lat.node_xyz[lat.beam_conn[:, 1]])
    shape = np.sin(2 * np.pi * mid[:, 0] / n) * np.cos(2 * np.pi * mid[:, 1] / n)
    path = [lat.beam_radius * (1.0 + 0.012 * k * shape) for k in range(7)]
    total, sols = {}, {}
    for warm in (0, 1, 2, 3, 4):
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                              palette=1, tile_nodes=64, coarse_max_dofs=600, condense=condense, warm_start=warm) as dev:
            dev.set_bc(fixed, None, f)
            its, us = [], []
            for r in path:
                dev.update_radii(r)
                dev.assemble()
                u, st = dev.solve(rtol=1e-9, max_iter=20000)
                assert st["converged"] == 1
                its.append(int(st["iterations"]))
                us.append(u)
            total[warm], sols[warm] = its, us
    for warm in (1, 2, 3, 4):
        for k in range(len(path)):
            assert _rel(sols[warm][k], sols[0][k]) < 1e-7, (warm, k)
    tail = {w: sum(v[3:]) for w, v in total.items()}          # (from the fourth solve on all of them have their history)
    assert tail[2] < tail[1] < tail[0], total
    assert tail[3] < tail[1], total
    # 4 = the Galerkin start: the best combination of the last (up to) four solutions for the current system - it contains
    # the candidates of 1, 2 and 3
    assert tail[4] <= tail[2], total


def test_failed_palette_attempts_back_off_and_a_success_brings_the_palette_back():
    """Round 5: on a graded lattice every strut has its own record, so the record-palette attempt of pl_assemble fails; after a
    failure the next 1, 3, 7, 15 assemblies go without an attempt (the streaming form of K*p meanwhile: the same operator), and
    when the radii become uniform again an attempt at the end of the current pause succeeds and the palette form returns.  Every
    solve on the way reaches its residual."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 10
    lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == float(n), 2] = -1e-3
    rng = np.random.default_rng(4)
    graded = lat.beam_radius * (1.0 + 0.2 * rng.random(lat.n_beams))
    forms = []
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                          palette=1) as dev:
        dev.set_bc(fixed, None, f)

        def step(r):
            dev.update_radii(r)
            dev.assemble()
            u, st = dev.solve(rtol=1e-9, max_iter=20000)
            assert st["converged"] == 1
            res = np.where(fixed, 0.0, f - dev.spmv(u))
            assert np.linalg.norm(res) <= 2e-9 * np.linalg.norm(f)
            forms.append(int(st["kp_form"]))
        step(lat.beam_radius)                      # uniform: palette form
        assert forms[-1] == 1
        for _ in range(4):                         # graded: attempt fails, pause, attempt fails, pause ...
            step(graded)
        assert all(k != 1 for k in forms[1:])
        for _ in range(20):                        # uniform again: the palette returns once the pause is over
            step(lat.beam_radius)
            if forms[-1] == 1:
                break
        assert forms[-1] == 1, forms
        step(lat.beam_radius)                      # ... and stays
        assert forms[-1] == 1


@pytest.mark.parametrize("condense", [0, 1])
def test_galerkin_start_with_repeated_scaled_and_unrelated_right_hand_sides(condense):
    """opts.warm_start = 4 is a projection onto the handle's stored solutions: the identical system again and a scaled load are
    answered (almost) at once - the stored vectors are dependent then, which the pivot floor of the small Cholesky absorbs -,
    an unrelated load costs what a cold start costs (never more), and every answer reaches its residual."""
    from pylatticedso_amd import lattice_arrays as LA
    n = 12
    lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed = np.zeros((lat.n_nodes, 6), np.uint8)
    fixed[lat.node_xyz[:, 0] == 0.0] = 1
    f = np.zeros((lat.n_nodes, 6))
    f[lat.node_xyz[:, 0] == float(n), 2] = -1e-3
    g = np.zeros_like(f)
    g[lat.node_xyz[:, 0] == float(n), 1] = 1e-3
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, precond=3,
                          palette=1, warm_start=4, condense=condense, tile_nodes=64, coarse_max_dofs=600) as dev:
        its = []
        for load in (f, f, f, 2.0 * f, 2.0 * f, g, g):
            dev.set_bc(fixed, None, load)
            dev.assemble()
            u, st = dev.solve(rtol=1e-9, max_iter=20000)
            assert st["converged"] == 1 and np.isfinite(u).all()
            r = np.where(fixed, 0.0, load - dev.spmv(u))
            assert np.linalg.norm(r) <= 2e-9 * np.linalg.norm(load)
            its.append(int(st["iterations"]))
    cold = its[0]
    assert cold > 100 and max(its[1:5]) <= 3, its           # the same system, the same load scaled
    assert its[5] <= cold + 3 and its[6] <= 3, its          # an unrelated load: a cold start's count, then known
