"""Surrogate Schur complements (reduced basis + nearest / linear / thin-plate-spline coefficients) against outputs of
the running reference (tests/golden/surrogate_bcc.npz, made by make_golden.py::gen_surrogate from the reference's own
reduced basis of the BCC cell) and against the reference's dolfinx dataset."""
import os

import numpy as np
import pytest

from pylatticedso_amd.schur_surrogate import SchurSurrogate, ThinPlateSpline, reduced_basis_file_name

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "surrogate_bcc.npz"))


def _surrogate(kind):
    return SchurSurrogate.load(["BCC"], 1e-6, kind, search_dirs=[GOLD])


def test_file_name_rule():
    # greedy_algorithm.py:214-233
    assert reduced_basis_file_name(["BCC"], 1e-6) == "reduced_basis_BCC_tol_1e-6.npz"
    assert reduced_basis_file_name(["BCC", "Hybrid4"], 1e-3) == "reduced_basis_BCC_Hybrid4_tol_1e-3.npz"
    with pytest.raises(FileNotFoundError):
        SchurSurrogate.load(["Octet"], 1e-6, "RBF", search_dirs=[GOLD])
    with pytest.raises(NotImplementedError):
        SchurSurrogate.load(["BCC"], 1e-6, "kriging", search_dirs=[GOLD])


@pytest.mark.parametrize("kind", ["nearest_neighbor", "linear", "RBF"])
def test_surrogate_matrices_match_reference(gold, kind):
    """S(r) inside the training range, at training points, and outside it (clamped / extrapolated) - same matrices as
    LatticeSim.get_schur_complement_from_reduced_basis_batch of the reference."""
    S = _surrogate(kind).schur_batch(gold["radii"])
    ref = gold[f"S_{kind}"]
    assert S.shape == ref.shape == (len(gold["radii"]), 48, 48)
    for q in range(len(ref)):
        assert _rel(S[q], ref[q]) < 1e-10, (kind, float(gold["radii"][q, 0]))


def test_rbf_gradient_matches_reference_and_finite_differences(gold):
    sur = _surrogate("RBF")
    for q, r in enumerate(gold["radii"][:5, 0]):
        dS = sur.schur_gradients([r])[0]
        assert _rel(dS, gold["dS_RBF"][q]) < 1e-9
        h = 1e-6
        Sp, Sm = sur.schur_batch([[r + h], [r - h]])
        assert _rel(dS, (Sp - Sm) / (2 * h)) < 1e-5


def test_linear_gradient_is_the_reference_finite_difference(gold):
    sur = _surrogate("linear")
    for q, r in enumerate(gold["radii"][:5, 0]):
        got, ref = sur.schur_gradients([r])[0], gold["dS_linear_fd"][q]
        # a difference quotient over 2e-6 of numbers ~1e3: compare on the matrix scale
        assert np.abs(got - ref).max() < 1e-6 * np.abs(ref).max()


def test_rbf_surrogate_reproduces_the_dolfinx_dataset(golden_dir=GOLD):
    """The reduced basis was built (tolerance 1e-6) from the reference's dolfinx Schur complements: the surrogate must
    give those matrices back at the training radii - and these are the matrices oracle/ and pl_schur reproduce."""
    d = np.load(os.path.join(GOLD, "schur_BCC.npz"))
    sur = _surrogate("RBF")
    S = sur.schur_batch(d["radius_values"])
    for q in range(len(S)):
        assert _rel(S[q], d["schur_matrices"][q]) < 1e-5


def test_thin_plate_spline_nd():
    """2-parameter cells (hybrid geometries): interpolation property, exactness on affine data, analytic gradient."""
    rng = np.random.default_rng(5)
    X = rng.uniform(0.01, 0.1, (14, 2))
    Y = np.c_[np.sin(20 * X[:, 0]) + X[:, 1] ** 2, 3.0 + 2.0 * X[:, 0] - 5.0 * X[:, 1]]
    tps = ThinPlateSpline(X, Y)
    assert np.abs(tps.evaluate(X) - Y).max() < 1e-9
    q = rng.uniform(0.01, 0.1, (5, 2))
    assert np.abs(tps.evaluate(q)[:, 1] - (3.0 + 2.0 * q[:, 0] - 5.0 * q[:, 1])).max() < 1e-9
    g = tps.gradient(q)
    h = 1e-7
    for d in range(2):
        e = np.zeros(2)
        e[d] = h
        fd = (tps.evaluate(q + e) - tps.evaluate(q - e)) / (2 * h)
        assert np.abs(g[:, d, :] - fd).max() < 1e-4 * max(1.0, np.abs(fd).max())


def test_linear_surrogate_nd_falls_back_to_nearest_outside_the_hull():
    rng = np.random.default_rng(2)
    pts = rng.uniform(0.0, 1.0, (12, 2))
    alpha = np.c_[1.0 + pts[:, 0] + 2.0 * pts[:, 1], pts[:, 0] - pts[:, 1]]       # affine -> exact inside the hull
    sur = SchurSurrogate(np.eye(4)[:, :2], alpha.T, pts, "linear")
    inside = pts[:3].mean(axis=0, keepdims=True)
    a = sur.alphas(inside)
    assert np.allclose(a, [[1.0 + inside[0, 0] + 2.0 * inside[0, 1], inside[0, 0] - inside[0, 1]]])
    far = np.array([[5.0, 5.0]])
    i0 = np.argmin(np.linalg.norm(pts - far, axis=1))
    assert np.allclose(sur.alphas(far), alpha[i0][None])


# ---- greedy reduced-basis construction (tests/golden/greedy_bcc.npz: the reference's own greedy run on its dolfinx
# ---- dataset of the BCC cell) -------------------------------------------------------------------------------------
@pytest.mark.parametrize("tol", [1e-3, 1e-6])
def test_greedy_reduced_basis_matches_reference(tol, tmp_path):
    from pylatticedso_amd.greedy_algorithm import reduce_basis_greedy
    g = np.load(os.path.join(GOLD, "greedy_bcc.npz"))
    data = {tuple(r): m for r, m in zip(g["radius_values"], g["schur_matrices"])}
    tag = f"{tol:.0e}"
    main, coef, fields, basis, alpha, matP, norms = reduce_basis_greedy(dict(data), tol, "rb_test", verbose=0,
                                                                        root=str(tmp_path))
    assert list(main) == list(g[f"mainelem_{tag}"])
    assert basis.shape == g[f"basis_{tag}"].shape
    assert np.abs(basis - g[f"basis_{tag}"]).max() < 1e-9
    assert _rel(alpha, g[f"alpha_{tag}"]) < 1e-9
    assert _rel(coef, g[f"reducedcoef_{tag}"]) < 1e-8
    assert _rel(matP, g[f"matP_{tag}"]) < 1e-9 and _rel(norms, g[f"norms_{tag}"]) < 1e-12
    # (one-pass Gram-Schmidt, as in the reference: the last vectors normalise residuals of ~1e-6)
    assert np.abs(basis.T @ basis - np.eye(basis.shape[1])).max() < 1e-6
    # the file it wrote is what the surrogate modes load
    d = np.load(tmp_path / "data" / "outputs" / "schur_complement" / "reduced_basis" / "rb_test.npz")
    assert set(d.files) == {"basis_reduced_ortho", "alpha_ortho", "list_elements"}
    s2 = SchurSurrogate(d["basis_reduced_ortho"], d["alpha_ortho"], d["list_elements"], "RBF")
    S = s2.schur_batch(g["radius_values"])
    worst = max(_rel(S[q], g["schur_matrices"][q]) for q in range(len(S)))
    assert worst < (5e-3 if tol == 1e-3 else 1e-5)


def test_greedy_1e6_reproduces_the_committed_reduced_basis():
    """Same dataset, same tolerance -> the reduced basis the reference ships (up to the sign of a basis vector)."""
    from pylatticedso_amd.greedy_algorithm import reduce_basis_greedy
    g = np.load(os.path.join(GOLD, "greedy_bcc.npz"))
    ref = np.load(os.path.join(GOLD, "reduced_basis_BCC_tol_1e-6.npz"))
    data = {tuple(r): m for r, m in zip(g["radius_values"], g["schur_matrices"])}
    basis, alpha = reduce_basis_greedy(data, 1e-6, None, verbose=0)[3:5]
    assert basis.shape == ref["basis_reduced_ortho"].shape
    sign = np.sign((basis * ref["basis_reduced_ortho"]).sum(axis=0))
    assert np.abs(basis * sign - ref["basis_reduced_ortho"]).max() < 1e-8
    assert _rel(alpha * sign[:, None], ref["alpha_ortho"]) < 1e-8


def test_preconditioner_dataset_lookup_and_cell_parameter_radii(golden_dir):
    """Host side of the reference's DDM preconditioner (lattice_sim.py:1312-1329,1351-1415): which dataset each
    preconditioner_type reads, the fall-backs of "mean", and Cell.radii (un-graded) as the key of everything."""
    import json
    import os
    from pylatticedso_amd.lattice_sim import LatticeSim
    preset = json.loads(str(np.load(os.path.join(golden_dir, "ddm_bcc_4x2x2.npz"))["preset_json"]))
    ddm = preset["simulation_parameters"]["DDM"]
    ddm.update(enable_preconditioner=True, preconditioner_type="nearest_reference")
    preset["gradient"] = {"radii": {"rule": "linear", "direction_x": True, "direction_y": False, "direction_z": False,
                                    "parameter_x": 0.25, "parameter_y": 0.0, "parameter_z": 0.0}}
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    # the preset gradient scales the struts, not Cell.radii: one surrogate matrix, parameter radii all 0.05
    assert len({tuple(np.round(r, 8)) for r in L.lattice.cell_radii}) > 1
    assert np.allclose(L._cell_parameter_radii(), 0.05) and L.schur_complements.shape[0] == 1
    path = L._define_preconditioner_approximation()
    assert path.endswith("Schur_complement_BCC.npz") and L.used_schur_preconditioner["schur_matrices"].shape == (10, 48, 48)
    L.preconditioner_type, L.used_schur_preconditioner = "mean", None
    path = L._define_preconditioner_approximation()          # no Schur_complement_mean_BCC.npz: mean of the dataset
    d = np.load(os.path.join(golden_dir, "Schur_complement_BCC.npz"))
    assert path.endswith("Schur_complement_BCC.npz")
    assert np.allclose(L.used_schur_preconditioner["schur_matrices"], d["schur_matrices"].mean(axis=0))
    L.data_roots, L.used_schur_preconditioner = [], None
    os.environ.pop("PYLATTICE_DATA_ROOT", None)
    assert L._define_preconditioner_approximation() is None and L._mean_of_own_cells     # nothing on disk
    L.preconditioner_type = "exact"
    assert L._define_preconditioner_approximation() is None
    L.preconditioner_type = "nearest_reference"
    with pytest.raises(FileNotFoundError):
        L._define_preconditioner_approximation()
    ddm["preconditioner_type"] = "nonsense"
    with pytest.raises(NotImplementedError):
        LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    # per-cell radii the way LatticeOpti sets them
    ddm["preconditioner_type"] = "exact"
    preset.pop("gradient")
    L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[golden_dir])
    L.set_cell_radii(0.04 + 0.0125 * L.lattice.cell_pos[:, 0])
    assert L.schur_complements.shape[0] == 4 and np.allclose(np.unique(L._cell_parameter_radii()), [0.04, 0.0525, 0.065, 0.0775])
