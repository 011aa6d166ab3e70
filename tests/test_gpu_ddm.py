"""Domain-decomposition path on the GPU (pl_create_ddm + the solve_DDM mirror) against solves produced by the
REFERENCE's own solve_DDM (tests/golden/ddm_*.npz: xsol, the Schur matrix it used, BC state)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from pylatticedso_amd import _capi                     # noqa: E402
from pylatticedso_amd.lattice_sim import LatticeSim   # noqa: E402


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.mark.parametrize("name", ["bcc_4x2x2", "bcc_4x4x4", "bcc_6x3x3"])
def test_solve_ddm_reproduces_reference_solution(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"ddm_{name}.npz"))
    preset = json.loads(str(g["preset_json"]))
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    assert not L.is_penalized                       # surrogate mode: penalisation lives in the Schur matrices
    # the preset asks for the RBF surrogate: the host mirror evaluates it from the reference's reduced basis and must
    # arrive at the matrix the reference evaluated
    assert L.schur_complements.shape == (1, 48, 48) and _rel(L.schur_complements[0], g["schur"]) < 1e-10
    xsol, info, idx, b = L.solve_DDM()
    assert info == int(g["info"]) == 0
    assert len(xsol) == len(g["xsol"]) and len(b) == len(g["b"])
    assert abs(np.linalg.norm(b) - np.linalg.norm(g["b"])) < 1e-12 * np.linalg.norm(g["b"])
    # both are CG iterates stopped at ||r|| <= 1e-6 ||b||: matching solutions and the same iteration count up to a
    # few steps (the device scatter sums with f64 atomics, whose order - hence the last bits of every K*p - changes
    # from run to run; over ~200 unpreconditioned iterations that moves the stopping step by 0-3)
    assert abs(L.iteration - int(g["iterations"])) <= max(3, int(g["iterations"]) // 40)
    assert _rel(xsol, g["xsol"]) < 1e-5
    # the reference's solution satisfies OUR operator to its own stopping tolerance (operator + RHS parity)
    dev = L.ddm_model()
    bn = L._boundary_nodes_by_index()
    u_ref = np.zeros((len(bn), 6))
    free = ~L.fixed_DOF[bn]
    order = np.concatenate([6 * L.index_boundary[n] + np.flatnonzero(~L.fixed_DOF[n]) for n in L._boundary_visit_order])
    u_ref.ravel()[order] = g["xsol"]
    res = np.where(free, L.applied_force[bn] - dev.spmv(u_ref), 0.0)
    assert np.linalg.norm(res) <= 1.05e-6 * np.linalg.norm(b)


def test_ddm_exact_schur_agrees_with_fem(golden_dir):
    """DDM with exact (device-condensed) cell Schur complements == FEM solve of the same penalised lattice, on the
    cell-boundary nodes (the reference's compare_FEM_DDM.py experiment)."""
    from pylatticedso_amd.utils_simulation import solve_FEM_FenicsX
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))
    preset = json.loads(str(g["preset_json"]))
    preset["simulation_parameters"]["DDM"]["schur_complement_computation"] = {"type": "exact"}
    preset["simulation_parameters"]["DDM"]["max_iterations"] = 5000
    Ld = LatticeSim(preset, enable_domain_decomposition_solver=True)
    assert Ld.is_penalized
    # every cell of this lattice has the same radii -> one representative matrix, as in the reference's grouping
    assert Ld.schur_complements.shape == (1, 48, 48)
    xs_d, info, _, _ = Ld.solve_DDM()
    p2 = json.loads(json.dumps(preset))
    p2["simulation_parameters"].pop("DDM")
    Lf = LatticeSim(p2)
    xs_f, _ = solve_FEM_FenicsX(Lf)
    # the representative-cell Schur (first cell: corner cell with un-penalised outer joints) is reused for interior
    # cells, exactly as the reference does - hence a modelling difference, not a solver error
    assert info == 0 and _rel(xs_d, xs_f) < 5e-2


@pytest.mark.parametrize("kind", ["nearest_neighbor", "linear", "RBF"])
def test_ddm_surrogate_kinds_on_a_graded_lattice(golden_dir, kind):
    """Cells with different radii (one Schur matrix per distinct radius, evaluated in one batch) with every surrogate
    kind: the device DDM solution satisfies the assembled condensed system built on the host from the same cell
    matrices."""
    preset = json.loads(str(np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))["preset_json"]))
    preset["simulation_parameters"]["DDM"]["schur_complement_computation"] = {"type": kind, "precision_greedy": 1e-6}
    graded = json.loads(json.dumps(preset))
    graded["gradient"] = {"radii": {"rule": "linear", "direction_x": True, "direction_y": False, "direction_z": False,
                                    "parameter_x": 0.25, "parameter_y": 0.0, "parameter_z": 0.0}}
    Lg = LatticeSim(graded, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    # the preset's radius gradient scales the struts but not Cell.radii, the argument of the reference's surrogates
    # (cell.py:86,407-412; lattice_sim.py:846-919): one matrix for all cells, as in the reference
    assert len({tuple(np.round(r, 8)) for r in Lg.lattice.cell_radii}) > 1 and Lg.schur_complements.shape[0] == 1
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    L.set_cell_radii(0.04 + 0.0125 * L.lattice.cell_pos[:, 0])
    n_distinct = len({tuple(np.round(r, 8)) for r in L._cell_parameter_radii()})
    assert n_distinct == 4 and L.schur_complements.shape == (n_distinct, 48, 48)
    xsol, info, idx, b = L.solve_DDM()
    assert info == 0
    # host assembly of sum_c B_c^T S_c B_c on the free dofs, in the ordering of xsol
    cb = L.cell_boundary_nodes()
    nb = L.max_index_boundary + 1
    K = np.zeros((6 * nb, 6 * nb))
    for c in range(L.lattice.n_cells):
        dofs = (6 * L.index_boundary[cb[c]][:, None] + np.arange(6)).ravel()
        K[np.ix_(dofs, dofs)] += L.schur_complements[L.cell_schur_index[c]]
    order = np.concatenate([6 * L.index_boundary[n] + np.flatnonzero(~L.fixed_DOF[n]) for n in L._boundary_visit_order])
    Kff = K[np.ix_(order, order)]
    assert np.linalg.norm(Kff @ xsol - b) <= 1.05e-6 * np.linalg.norm(b)
    assert _rel(xsol, np.linalg.solve(Kff, b)) < 1e-3


@pytest.mark.parametrize("case,iters", [("uniform_4x2x2_exact", 1), ("uniform_4x2x2_nearest_reference", 1),
                                        ("varied_6x3x3_exact", 1), ("varied_6x3x3_nearest_reference", 10)])
def test_preconditioned_ddm_reproduces_reference(golden_dir, case, iters):
    """The reference's assembled-Schur preconditioner (build_preconditioner, lattice_sim.py:1351-1415), factorised on
    the device: same solution and the same number of CG iterations as the reference's SuperLU-preconditioned solve."""
    g = np.load(os.path.join(golden_dir, "ddm_preconditioned.npz"))
    assert int(g[f"{case}_iterations"]) == iters and int(g[f"{case}_info"]) == 0
    L = LatticeSim(json.loads(str(g[f"{case}_preset_json"])), enable_domain_decomposition_solver=True,
                   data_roots=[golden_dir])
    if case.startswith("varied"):
        pos = L.lattice.cell_pos
        L.set_cell_radii(0.034 + 0.011 * pos[:, 0] + 0.002 * pos[:, 2])
        ref = {tuple(p): r for p, r in zip(g[f"{case}_cell_pos"].tolist(), g[f"{case}_cell_radii"].ravel())}
        assert np.allclose([ref[tuple(p)] for p in pos.tolist()], L._cell_parameter_radii().ravel(), atol=1e-14)
    xsol, info, _, b = L.solve_DDM()
    assert L._ddm_precond == 2 and info == 0
    assert np.allclose(b, g[f"{case}_b"], rtol=1e-9, atol=1e-12 * np.abs(g[f"{case}_b"]).max())
    assert L.iteration == iters
    # both stop at ||r|| <= 1e-6 ||b||; with the exact preconditioner the first step already lands on the solution
    assert _rel(xsol, g[f"{case}_xsol"]) < (1e-9 if iters == 1 else 2e-6)


def test_mean_preconditioner_and_the_jacobi_fallback(golden_dir, capsys, monkeypatch):
    """preconditioner_type "mean" (what the reference's DDM presets name): its Schur_complement_mean_*.npz is not in
    the reference's checkout, so the mean of the radius dataset stands in.  Beyond the dense limit the device falls
    back to the node blocks of the assembled matrix (block Jacobi), says so once and lifts the iteration cap those
    presets carry (optimization_DDM_surrogate: 10)."""
    import pylatticedso_amd.lattice_sim as LS
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))
    preset = json.loads(str(g["preset_json"]))
    ddm = preset["simulation_parameters"]["DDM"]
    ddm.update(enable_preconditioner=True, preconditioner_type="mean", max_iterations=400)
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    xsol, info, _, _ = L.solve_DDM()
    assert L._ddm_precond == 2 and info == 0 and _rel(xsol, g["xsol"]) < 1e-5
    assert L.iteration < int(g["iterations"]) // 4        # plain CG of the golden: ~200
    assert "exceed" not in capsys.readouterr().out
    # no dataset anywhere: the mean over the lattice's own cell matrices, re-evaluated when the cells change
    ddm["max_iterations"] = 200
    L2 = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    L2._mean_of_own_cells = True
    _, info2, _, _ = L2.solve_DDM()
    assert info2 == 0 and L2.iteration == 1 and L2.used_schur_preconditioner is None     # one matrix: exact
    L2.set_cell_radii(0.04 + 0.0125 * L2.lattice.cell_pos[:, 0])
    x2, info2, _, b2 = L2.solve_DDM()
    assert info2 == 0 and 1 < L2.iteration < 60
    # Jacobi fallback
    monkeypatch.setattr(LS, "DDM_DENSE_MAX", 100)
    ddm["max_iterations"] = 10
    L3 = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    xsol3, info3, _, _ = L3.solve_DDM()
    L3.solve_DDM()
    # (round 5: above the dense limit the node blocks carry a dense level - here one aggregate of 45 nodes, 12 global modes)
    assert L3._ddm_precond == 4 and info3 == 0 and L3.iteration > 10 and _rel(xsol3, g["xsol"]) < 1e-5
    assert int(L3.ddm_model().last_stats["precond_used"]) == 4
    assert L3.iteration < int(g["iterations"])            # ... and beat the plain CG count
    assert capsys.readouterr().out.count("exceed") == 1
    ddm.pop("preconditioner_type")
    with pytest.raises(ValueError):
        LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    ddm["preconditioner_type"] = "something"
    with pytest.raises(NotImplementedError):
        LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])


def test_node_block_jacobi_of_the_ddm_operator(golden_dir):
    """opts.precond = 3 on a DDM handle: CG preconditioned by the inverted 6 x 6 diagonal blocks of sum_c B^T S B (what
    solve_DDM asks for above the dense limit).  Same solution as plain and Jacobi CG, fewer iterations than Jacobi, the blocks
    follow a changed Dirichlet set, and the result equals numpy's block-preconditioned operator applied by hand."""
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))
    preset = json.loads(str(g["preset_json"]))
    preset["geometry"]["number_of_cells"] = dict(x=8, y=4, z=4)
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    L.set_cell_radii(0.034 + 0.03 * L.lattice.cell_pos[:, 0] / 7.0)
    L.ddm_model()                                         # (fills schur_complements / index tables)
    cb = L.cell_boundary_nodes()
    n_nodes = L.max_index_boundary + 1
    nodes = L.index_boundary[cb]
    bn = L._boundary_nodes_by_index()
    fixed = L.fixed_DOF[bn]
    f = L.applied_force[bn]
    out = {}
    for pre in (0, 1, 3):
        with _capi.HipLattice.ddm(n_nodes, nodes, L.schur_complements, L.cell_schur_index, precond=pre) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-10, max_iter=20000)
            assert st["converged"] == 1 and int(st["precond_used"]) == pre
            out[pre] = (u, st["iterations"])
            if pre == 3:
                # a changed Dirichlet set: the blocks are rebuilt, constrained dofs stay where they are
                fixed2 = fixed.copy()
                fixed2[np.argmax(f[:, 2] != 0)] = 1
                dev.set_bc(fixed2, None, f)
                dev.assemble()
                u2, st2 = dev.solve(rtol=1e-10, max_iter=20000)
                assert st2["converged"] == 1 and np.all(u2[fixed2 != 0] == 0.0)
                r2 = np.where(fixed2, 0.0, f - dev.spmv(u2))
                assert np.linalg.norm(r2) <= 2e-10 * np.linalg.norm(np.where(fixed2, 0.0, f))
    assert _rel(out[3][0], out[0][0]) < 1e-7 and _rel(out[1][0], out[0][0]) < 1e-7
    assert out[3][1] < 0.9 * out[1][1] < 0.9 * out[0][1]


@pytest.mark.parametrize("tag,kw", [
    ("plain", dict(M=False, maxiter=200, tol=1e-10, mintol=1e-14, restart_every=500000, alpha_max=100)),
    ("jacobi", dict(M=True, maxiter=200, tol=1e-10, mintol=1e-14, restart_every=500000, alpha_max=100)),
    ("clamped", dict(M=False, maxiter=25, tol=1e-10, mintol=1e-14, restart_every=7, alpha_max=0.01)),
    ("dirstop", dict(M=False, maxiter=200, tol=1e-14, mintol=2e-4, restart_every=500000, alpha_max=100)),
    ("restart_jacobi", dict(M=True, maxiter=40, tol=1e-9, mintol=1e-14, restart_every=5, alpha_max=100)),
    ("tiny_step", dict(M=False, maxiter=12, tol=1e-10, mintol=1e-14, restart_every=500000, alpha_max=5e-7)),
])
def test_device_cg_has_the_reference_semantics(golden_dir, tag, kw):
    """The device CG against runs of the reference's conjugate_gradient_solver (cg_trace.npz, dumped from the running
    reference): alpha clamp, restart (on the previous z with a preconditioner, on the updated residual without), the
    direction-norm stop and the info codes 0 / 1 / 2 (conjugate_gradient_solver.py:79,96-109).  The 60 x 60 SPD matrix
    is handed over as the Schur complement of one 10-node "cell"; its Jacobi preconditioner is precond = 1."""
    g = np.load(os.path.join(golden_dir, "cg_trace.npz"))
    A, b = g["A"], g["b"]
    dev = _capi.HipLattice.ddm(10, np.arange(10)[None, :], A, np.zeros(1, np.int32), precond=1 if kw["M"] else 0,
                               alpha_max=kw["alpha_max"], mintol=kw["mintol"], restart_every=kw["restart_every"],
                               check_every=1)
    with dev:
        dev.set_bc(np.zeros((10, 6), bool), None, b.reshape(10, 6))
        dev.assemble()
        x, st = dev.solve(rtol=kw["tol"], max_iter=kw["maxiter"], raise_on_noconv=False)
    assert int(st["info"]) == int(g[f"{tag}_info"])
    assert st["iterations"] == len(g[f"{tag}_trace"])
    assert np.allclose(x.ravel(), g[f"{tag}_x"], rtol=1e-8, atol=1e-12)
    if tag == "dirstop":
        assert int(st["stop_reason"]) == 1 and st["converged"] == 1


@pytest.mark.parametrize("nb", [2, 8, 12, 14, 20, 26, 27])      # (27 = the library's limit: every corner, edge and face node + one)
def test_cell_product_on_the_matrix_pipe_for_every_cell_size(nb, monkeypatch):
    """The DDM operator sum_c B_c^T S_c B_c for cells with nb boundary nodes (m = 6 nb dofs): 8 = BCC, 12 = Hybrid1,
    26 = the reference's BCC + Hybrid1 (+ Hybrid4) hybrids (optimization/Cantilever_L_beam.json) - all on
    v_mfma_f64_16x16x4_f64 since round 5 (output columns of a tile dealt over several waves when S^T does not fit one
    wave's registers) - against numpy, and against the generic kernels (PL_DDM_MFMA=0) of the same library."""
    rng = np.random.default_rng(nb)
    n_nodes, n_cells, n_S = 60, 45, 3
    m = 6 * nb
    cell_nodes = np.stack([rng.choice(n_nodes, nb, replace=False) for _ in range(n_cells)]).astype(np.int32)
    A = rng.standard_normal((n_S, m, m))
    S = A @ A.transpose(0, 2, 1) + m * np.eye(m)
    cell_S = rng.integers(0, n_S, n_cells).astype(np.int32)
    x = rng.standard_normal((n_nodes, 6))
    x[cell_nodes[3]] = 0.0                       # one cell at rest: the reference's skip rule (lattice_sim.py:1239)
    y_ref = np.zeros((n_nodes, 6))
    for c in range(n_cells):
        xc = x[cell_nodes[c]].ravel()
        if xc.sum() == 0.0:
            continue
        np.add.at(y_ref, cell_nodes[c], (S[cell_S[c]] @ xc).reshape(nb, 6))
    ys = {}
    for mfma in ("1", "0"):
        monkeypatch.setenv("PL_DDM_MFMA", mfma)
        with _capi.HipLattice.ddm(n_nodes, cell_nodes, S, cell_S) as dev:
            dev.set_bc(np.zeros((n_nodes, 6), bool), None, np.zeros((n_nodes, 6)))
            dev.assemble()
            ys[mfma] = dev.spmv(x)
            assert int(dev.time_kernel(0, 2) > 0)
    assert _rel(ys["1"], y_ref) < 1e-13
    assert _rel(ys["0"], y_ref) < 1e-13


@pytest.mark.parametrize("cells", [(12, 6, 6), (8, 8, 8)])
def test_two_level_preconditioner_of_the_ddm_operator(golden_dir, cells):
    """opts.precond = 4 on a DDM handle (round 5): the node blocks of precond = 3 plus a dense level of 12 modes (rigid +
    uniform strain) per aggregate of boundary nodes, A_c = Z^T P G P Z assembled from the cell matrices on the device.  Same
    solution as plain CG and as the node blocks alone, clearly fewer iterations than the node blocks, pl_stats_t.precond_used
    says 4; the level follows a changed Dirichlet set; a handle without node positions refuses to assemble."""
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))
    preset = json.loads(str(g["preset_json"]))
    preset["geometry"]["number_of_cells"] = dict(x=cells[0], y=cells[1], z=cells[2])
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    L.set_cell_radii(0.034 + 0.03 * L.lattice.cell_pos[:, 0] / (cells[0] - 1.0))
    L.ddm_model()
    cb = L.cell_boundary_nodes()
    n_nodes = L.max_index_boundary + 1
    nodes = L.index_boundary[cb]
    bn = L._boundary_nodes_by_index()
    fixed = L.fixed_DOF[bn]
    f = L.applied_force[bn]
    xyz = L.lattice.node_xyz[bn]
    out = {}
    for pre in (0, 3, 4):
        with _capi.HipLattice.ddm(n_nodes, nodes, L.schur_complements, L.cell_schur_index, precond=pre,
                                  node_xyz=xyz if pre == 4 else None) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-10, max_iter=20000)
            assert st["converged"] == 1 and int(st["precond_used"]) == pre
            r = np.where(fixed, 0.0, f - dev.spmv(u))
            assert np.linalg.norm(r) <= 2e-10 * np.linalg.norm(np.where(fixed, 0.0, f))
            out[pre] = (u, st["iterations"])
            if pre == 4:
                fixed2 = fixed.copy()
                fixed2[np.argmax(f[:, 2] != 0)] = 1
                dev.set_bc(fixed2, None, f)
                dev.assemble()
                u2, st2 = dev.solve(rtol=1e-10, max_iter=20000)
                assert st2["converged"] == 1 and int(st2["precond_used"]) == 4 and np.all(u2[fixed2 != 0] == 0.0)
                r2 = np.where(fixed2, 0.0, f - dev.spmv(u2))
                assert np.linalg.norm(r2) <= 2e-10 * np.linalg.norm(np.where(fixed2, 0.0, f))
                # a second solve on the same factorisation, other load
                f3 = np.zeros_like(f)
                f3[np.isclose(xyz[:, 0], xyz[:, 0].max()), 1] = 1e-3
                dev.set_bc(fixed, None, f3)
                dev.assemble()
                u3, st3 = dev.solve(rtol=1e-10, max_iter=20000)
                assert st3["converged"] == 1
                r3 = np.where(fixed, 0.0, f3 - dev.spmv(u3))
                assert np.linalg.norm(r3) <= 2e-10 * np.linalg.norm(np.where(fixed, 0.0, f3))
    assert _rel(out[4][0], out[0][0]) < 1e-7 and _rel(out[3][0], out[0][0]) < 1e-7
    assert out[4][1] < 0.8 * out[3][1], (out[4][1], out[3][1])
    with pytest.raises(ValueError):
        _capi.HipLattice.ddm(n_nodes, nodes, L.schur_complements, L.cell_schur_index, precond=4)
    with pytest.raises(ValueError):
        with _capi.HipLattice.ddm(n_nodes, nodes, L.schur_complements, L.cell_schur_index, precond=3) as dev:
            dev.set_ddm_geometry(xyz[:-1])                  # one position per node


def test_solve_ddm_above_the_dense_limit_uses_the_two_level_preconditioner(golden_dir, capsys):
    """LatticeSim.solve_DDM with enable_preconditioner beyond PL_DDM_DENSE_MAX boundary dofs: the handle is created with
    precond = 4 and the positions of the boundary nodes; the solution equals the one the node blocks alone give
    (DDM_LARGE_PRECOND = 3) to the CG tolerance, in fewer iterations."""
    import pylatticedso_amd.lattice_sim as LS
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))
    preset = json.loads(str(g["preset_json"]))
    preset["geometry"]["number_of_cells"] = dict(x=16, y=14, z=12)          # 17 * 15 * 13 * 6 = 19 890 boundary dofs
    preset["simulation_parameters"]["DDM"].update(enable_preconditioner=True, preconditioner_type="exact")
    res = {}
    for pre in (4, 3):
        LS.DDM_LARGE_PRECOND = pre
        try:
            L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
            xsol, info, idx, b = L.solve_DDM()
        finally:
            LS.DDM_LARGE_PRECOND = 4
        assert info == 0 and L._ddm_precond == pre and int(L.ddm_model().last_stats["precond_used"]) == pre
        res[pre] = (xsol, L.iteration)
    assert "dense level on aggregates" in capsys.readouterr().out
    assert _rel(res[4][0], res[3][0]) < 2e-5                  # (both stop at the reference's 1e-6)
    assert res[4][1] < 0.8 * res[3][1], (res[4][1], res[3][1])


@pytest.mark.parametrize("geoms,radii", [(["Hybrid1"], [0.04]), (["BCC", "Hybrid1"], [0.05, 0.03])])
def test_two_level_preconditioner_on_cells_with_edge_and_face_nodes(golden_dir, geoms, radii, monkeypatch):
    """precond = 4 on cells whose boundary is more than the eight corners (Hybrid1: 12 boundary nodes, m = 72; BCC + Hybrid1
    hybrids: 20): exact (device-condensed) Schur complements, the coarse operator assembled from them cell by cell, the node
    aggregates cut from the positions of corner, edge and face nodes alike.  solve_DDM reaches what the node blocks alone reach
    (DDM_LARGE_PRECOND = 3), in fewer iterations."""
    import pylatticedso_amd.lattice_sim as LS
    g = np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))
    preset = json.loads(str(g["preset_json"]))
    preset["geometry"].update(geom_types=geoms, radii=radii, number_of_cells=dict(x=8, y=4, z=4))
    preset["simulation_parameters"]["DDM"].update(enable_preconditioner=True, preconditioner_type="exact", max_iterations=20000,
                                                  schur_complement_computation={"type": "exact"})
    monkeypatch.setattr(LS, "DDM_DENSE_MAX", 100)
    res = {}
    for pre in (4, 3):
        monkeypatch.setattr(LS, "DDM_LARGE_PRECOND", pre)
        L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
        xsol, info, _, _ = L.solve_DDM()
        st = L.ddm_model().last_stats
        assert info == 0 and int(st["precond_used"]) == pre, (info, st["precond_used"])
        res[pre] = (xsol, L.iteration, L.cell_boundary_nodes().shape[1])
    assert res[4][2] == (12 if geoms == ["Hybrid1"] else 20)
    assert _rel(res[4][0], res[3][0]) < 2e-5
    assert res[4][1] < 0.8 * res[3][1], (res[4][1], res[3][1])
