"""Oracle's periodic homogenisation (oracle.homogenize_submeshed, restating homogenization_cell.py) against closed
forms, invariants and - since round 5 - REFERENCE-HELD data: the reference commits no homogenisation output and cannot run
here (dolfinx_mpc), but the periodic homogenised matrix of a cell follows from its Schur complement on the boundary nodes
alone, and the reference commits those (tests/golden/homogenized_from_schur.npz, made by
tests/golden/make_homogenization_fixture.py).  tests/test_gpu_homogenization.py holds the device path to both."""
import os

import numpy as np
import pytest

from oracle import timoshenko_oracle as O

E, NU = 1013.0, 0.3


def _simple_cubic(r, h=0.05):
    """One cell whose 12 edges are struts of radius r (un-penalised), corners tagged as Point.tag_point does."""
    corners = np.array([[i, j, k] for i in (0.0, 1.0) for j in (0.0, 1.0) for k in (0.0, 1.0)])
    segs = [(a, b) for a in range(8) for b in range(a + 1, 8) if np.abs(corners[a] - corners[b]).sum() == 1.0]
    tags = np.arange(1000, 1008)
    K, nv = O.assemble_submeshed(corners, np.array(segs), np.full(len(segs), r), E, NU, h)
    V = O.submesh_vertices(corners, np.array(segs), h)
    assert len(V) == nv
    return K, V, tags


def test_simple_cubic_cell_has_the_closed_form_moduli():
    r = 0.04
    K, V, tags = _simple_cubic(r)
    C, C_raw, _ = O.homogenize_submeshed(K, V, tags)
    EA = E * np.pi * r ** 2
    # four struts along every axis, uniform axial strain, no Poisson coupling between orthogonal struts
    assert np.allclose(np.diag(C)[:3], 4 * EA, rtol=1e-11)
    off = C - np.diag(np.diag(C))
    assert np.abs(off).max() < 1e-9 * EA
    # tensorial shear strain 1 (gamma = 2): every strut of the two families sees a unit relative transverse
    # displacement with clamped-clamped ends (the shared periodic joint cannot rotate by symmetry):
    # sigma_xy = 4 * 12 EI / (L^3 (1 + Phi)),  Phi = 12 EI / (kappa G A L^2)  (the four struts normal to the face)
    EI = E * np.pi * r ** 4 / 4
    GA = O.KAPPA * E / (2 * (1 + NU)) * np.pi * r ** 2
    shear = 4 * 12 * EI / (1 + 12 * EI / GA)
    assert np.allclose(np.diag(C)[3:], shear, rtol=5e-3)          # P1 sub-elements: O(h^2) off the exact beam
    assert np.linalg.norm(C_raw - C_raw.T) < 1e-10 * np.linalg.norm(C)


@pytest.mark.parametrize("name", ["bcc", "hybrid4"])
def test_reference_cells_cubic_symmetry_and_hill_mandel(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"lattice_{name}_1x1x1_periodic.npz"))
    keep = ~g["beam_dup"]
    K, nv = O.assemble_submeshed(g["node_xyz"], g["beam_conn"][keep], g["beam_radius"][keep], E, NU, 0.05)
    V = O.submesh_vertices(g["node_xyz"], g["beam_conn"][keep], 0.05)
    C, C_raw, u_tots = O.homogenize_submeshed(K, V, g["node_tag"])
    assert np.linalg.norm(C_raw - C_raw.T) < 1e-10 * np.linalg.norm(C)
    assert np.linalg.eigvalsh(C).min() > 0
    # cubic material: three independent constants
    assert np.allclose(np.diag(C)[:3], C[0, 0], rtol=1e-10) and np.allclose(np.diag(C)[3:], C[3, 3], rtol=1e-10)
    assert np.allclose([C[0, 1], C[0, 2], C[1, 2]], C[0, 1], rtol=1e-10)
    assert np.abs(C[:3, 3:]).max() < 1e-9 * C[0, 0]
    # Hill-Mandel: boundary-reaction stresses == strain-energy products of the total fields (shear rows: eps_ij and
    # eps_ji both work, hence the factor 2)
    en = np.array([[u_tots[i].ravel() @ (K @ u_tots[j].ravel()) for j in range(6)] for i in range(6)])
    scale = np.array([1, 1, 1, 2, 2, 2.0])
    assert np.allclose(en, C * scale[:, None], rtol=1e-9, atol=1e-9 * C[0, 0])


@pytest.mark.parametrize("name", ["bcc", "hybrid1", "hybrid4"])
def test_periodic_partners_by_coordinates_reproduce_the_tag_groups(golden_dir, name):
    """Host mirror: nodes matched modulo the cell size fall into the reference's corner / edge / face tag groups."""
    from pylatticedso_amd.homogenization_cell import imposed_displacement, periodic_masters
    g = np.load(os.path.join(golden_dir, f"lattice_{name}_1x1x1_periodic.npz"))
    xyz, tag = g["node_xyz"], g["node_tag"]
    master, boundary = periodic_masters(xyz, np.zeros(3), np.ones(3))
    assert np.array_equal(boundary, tag > 0)
    assert np.array_equal(master[~boundary], np.flatnonzero(~boundary))
    for group in O.CORNER_TAGS + O.EDGE_TAGS + O.FACE_TAGS:
        members = np.flatnonzero(np.isin(tag, group))
        if len(members):
            m = np.flatnonzero(tag == group[0])
            assert len(m) == 1 and np.all(master[members] == m[0])
    # the six macro strains: symmetric gradient of w_k is the k-th unit strain (shear tensorial)
    w = np.stack([imposed_displacement(k, xyz)[:, :3] for k in range(1, 7)])
    A = np.c_[xyz, np.ones(len(xyz))]
    for k, (i, j) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]):
        grad = np.linalg.lstsq(A, w[k], rcond=None)[0][:3].T          # d w_a / d x_b
        eps = 0.5 * (grad + grad.T)
        expect = np.zeros((3, 3))
        expect[i, j] = expect[j, i] = 1.0
        assert np.allclose(eps, expect, atol=1e-12)
    with pytest.raises(ValueError):
        imposed_displacement(7, xyz)


def test_engineering_constants_and_directional_modulus_of_an_isotropic_solid():
    """Post-processing of HomogenizedCell on a matrix with a known answer: isotropic C (tensorial-shear convention of
    the reference: C44 = 2 G) -> Ex = E, nu, G, and the same directional modulus in every direction."""
    from pylatticedso_amd.homogenization_cell import HomogenizedCell, directional_modulus
    Em, nu = 10.0, 0.3
    lam, G = Em * nu / ((1 + nu) * (1 - 2 * nu)), Em / (2 * (1 + nu))
    C = np.zeros((6, 6))
    C[:3, :3] = lam
    C[np.arange(3), np.arange(3)] = lam + 2 * G
    C[np.arange(3, 6), np.arange(3, 6)] = 2 * G
    H = HomogenizedCell.__new__(HomogenizedCell)
    H.homogenizeMatrix = C
    H.convert_to_orthotropic_form()
    H.compute_errors()
    M = H.orthotropicMatrix
    assert np.allclose([M[0, 0], M[1, 1], M[2, 2]], Em) and np.allclose([M[3, 3], M[4, 4], M[5, 5]], G)
    assert np.allclose([M[0, 1], M[0, 2], M[1, 2]], nu) and H._symmetryError == 0.0
    S = H.get_S_orthotropic()
    assert np.allclose(S[:3, :3] @ C[:3, :3], np.eye(3)) and np.allclose(np.diag(S)[3:], 1 / G)
    for theta, phi in [(90, 0), (90, 45), (54.7356, 45), (30, 200), (0, 0)]:
        assert np.isclose(np.linalg.norm(directional_modulus(S, theta, phi)), Em, rtol=1e-12)


def test_oracle_homogenisation_equals_what_the_reference_schur_complements_give(golden_dir):
    """The pin: homogenize_submeshed on the sub-meshed model of the reference's own 1x1x1 periodic BCC cell (r = 0.05,
    joint penalisation on - the settings of the reference's Schur dataset) against the matrix derived from the reference's
    committed dolfinx / PETSc Schur complement of that cell.  Two independent routes (full sub-meshed K with periodic
    constraints vs. the reference's condensed 48 x 48 operator) to 1e-10."""
    fx = np.load(os.path.join(golden_dir, "homogenized_from_schur.npz"))
    g = np.load(os.path.join(golden_dir, "lattice_bcc_1x1x1_periodic.npz"))
    keep = ~g["beam_dup"]
    K, _ = O.assemble_submeshed(g["node_xyz"], g["beam_conn"][keep], g["beam_radius"][keep], E, NU, 0.05)
    V = O.submesh_vertices(g["node_xyz"], g["beam_conn"][keep], 0.05)
    C, C_raw, _ = O.homogenize_submeshed(K, V, g["node_tag"])
    i = int(np.argmin(np.abs(fx["BCC_radius"] - 0.05)))
    assert fx["BCC_radius"][i] == 0.05
    assert np.linalg.norm(C - fx["BCC_C"][i]) < 1e-10 * np.linalg.norm(C)


def test_fixture_is_what_its_script_makes(golden_dir):
    """The committed fixture is reproducible from the committed Schur goldens (no reference needed), cubic for the three
    cells, positive definite, and monotone in the radius."""
    fx = np.load(os.path.join(golden_dir, "homogenized_from_schur.npz"))
    for geom in ("BCC", "Hybrid1", "Hybrid4"):
        sg = np.load(os.path.join(golden_dir, f"schur_{geom}.npz"))
        assert np.array_equal(sg["radius_values"].ravel(), fx[f"{geom}_radius"])
        for S, C in zip(sg["schur_matrices"], fx[f"{geom}_C"]):
            C2, _ = O.homogenize_from_schur(S, sg["boundary_node_xyz"])
            assert np.linalg.norm(C2 - C) < 1e-12 * np.linalg.norm(C)
            assert np.linalg.eigvalsh(C).min() > 0
            assert np.allclose(np.diag(C)[:3], C[0, 0], rtol=1e-8) and np.allclose(np.diag(C)[3:], C[3, 3], rtol=1e-8)
        assert np.all(np.diff(fx[f"{geom}_C"][:, 0, 0]) > 0)
