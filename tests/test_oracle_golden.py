"""The CPU oracle against the reference's golden data (no GPU).

* Schur_complement_{BCC,Hybrid1,Hybrid4}.npz are the reference's committed dolfinx/PETSc outputs
  (data/outputs/schur_complement/, produced by examples/simulation/construct_schur_complement_dataset.py).
  BCC was generated WITH joint penalisation, Hybrid1/Hybrid4 WITHOUT (established numerically: the
  un-penalised sub-meshed model reproduces them to 1e-13) - so together they pin the element, the
  gmsh subdivision rule and the penalised-segment handling.
* cg_trace.npz holds iterates of the reference's conjugate_gradient_solver on a fixed SPD matrix.
"""
import os

import numpy as np
import pytest

from oracle import timoshenko_oracle as O

E, NU = 1013.0, 0.3  # VeroClear (src/pyLatticeDesign/materials/VeroClear.json)


def _boundary_dofs(node_xyz, bxyz):
    ids = [int(np.argmin(np.linalg.norm(node_xyz - p, axis=1))) for p in bxyz]
    assert all(np.linalg.norm(node_xyz[i] - p) < 1e-12 for i, p in zip(ids, bxyz))
    return np.concatenate([6 * i + np.arange(6) for i in ids])


def test_schur_bcc_penalised_r005(golden_dir):
    st = np.load(os.path.join(golden_dir, "lattice_bcc_1x1x1_periodic.npz"))
    sg = np.load(os.path.join(golden_dir, "schur_BCC.npz"))
    keep = ~st["beam_dup"]
    K, _ = O.assemble_submeshed(st["node_xyz"], st["beam_conn"][keep], st["beam_radius"][keep], E, NU, 0.05)
    S = O.schur_complement(K, _boundary_dofs(st["node_xyz"], sg["boundary_node_xyz"]))
    i = list(np.round(sg["radius_values"].ravel(), 3)).index(0.05)
    G = sg["schur_matrices"][i]
    assert np.linalg.norm(S - G) / np.linalg.norm(G) < 1e-11


@pytest.mark.parametrize("geom,name", [("Hybrid1", "hybrid1"), ("Hybrid4", "hybrid4")])
def test_schur_unpenalised_all_radii(golden_dir, geom, name):
    st = np.load(os.path.join(golden_dir, f"lattice_{name}_1x1x1_periodic.npz"))
    sg = np.load(os.path.join(golden_dir, f"schur_{geom}.npz"))
    bd = _boundary_dofs(st["base_node_xyz"], sg["boundary_node_xyz"])
    for r, G in zip(sg["radius_values"].ravel(), sg["schur_matrices"]):
        rad = np.full(len(st["base_beam_conn"]), r)
        K, _ = O.assemble_submeshed(st["base_node_xyz"], st["base_beam_conn"], rad, E, NU, 0.05)
        S = O.schur_complement(K, bd)
        assert np.linalg.norm(S - G) / np.linalg.norm(G) < 1e-11, (geom, r)


def test_subdivision_rule_is_pinned(golden_dir):
    """round()/floor() subdivision rules must NOT reproduce the golden (guards the gmsh rule)."""
    st = np.load(os.path.join(golden_dir, "lattice_hybrid1_1x1x1_periodic.npz"))
    sg = np.load(os.path.join(golden_dir, "schur_Hybrid1.npz"))
    bd = _boundary_dofs(st["base_node_xyz"], sg["boundary_node_xyz"])
    G = sg["schur_matrices"][2]
    rad = np.full(len(st["base_beam_conn"]), sg["radius_values"].ravel()[2])
    orig = O.gmsh_subdivisions
    try:
        O.gmsh_subdivisions = lambda L, h: max(1, int(np.floor(L / h)))
        K, _ = O.assemble_submeshed(st["base_node_xyz"], st["base_beam_conn"], rad, E, NU, 0.05)
        S = O.schur_complement(K, bd)
        assert np.linalg.norm(S - G) / np.linalg.norm(G) > 1e-7
    finally:
        O.gmsh_subdivisions = orig


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_closed_form_condensation_matches_submeshed_chain(seed):
    rng = np.random.default_rng(seed)
    xa = rng.standard_normal(3)
    d = rng.standard_normal(3)
    L = 0.6 + 0.5 * rng.random()
    d *= L / np.linalg.norm(d)
    r = 0.02 + 0.06 * rng.random()
    l1, l2 = (0.03 + 0.1 * rng.random(), 0.0 if seed == 2 else 0.03 + 0.1 * rng.random())
    lm = L - l1 - l2
    h = 0.05
    t = d / L
    if l2 > 0:
        nodes = np.array([xa, xa + d, xa + t * l1, xa + t * (l1 + lm)])
        segs, rad = np.array([[0, 2], [2, 3], [3, 1]]), np.array([1.5 * r, r, 1.5 * r])
    else:
        nodes = np.array([xa, xa + d, xa + t * l1])
        segs, rad = np.array([[0, 2], [2, 1]]), np.array([1.5 * r, r])
    K, _ = O.assemble_submeshed(nodes, segs, rad, E, NU, h)
    S = O.schur_complement(K, np.arange(12))
    n = [O.gmsh_subdivisions(x, h) if x > 0 else 0 for x in (l1, lm, l2)]
    sc = O.condensed_beam(r, (l1, lm, l2), n, E, NU)
    K12 = O.beam_matrix(sc, d)
    assert np.linalg.norm(S - K12) / np.linalg.norm(S) < 1e-11
    xa6, xb6 = rng.standard_normal(6), rng.standard_normal(6)
    fa, fb = O.beam_apply(sc, d, xa6, xb6)
    ref = K12 @ np.r_[xa6, xb6]
    assert np.allclose(np.r_[fa, fb], ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    # rigid-body modes carry no force
    for om in np.eye(3):
        qa = np.r_[np.zeros(3), om]
        qb = np.r_[np.cross(om, d), om]
        assert np.abs(K12 @ np.r_[qa, qb]).max() < 1e-9 * np.abs(K12).max()


@pytest.mark.parametrize("tag,kw", [
    ("plain", dict(M=False, maxiter=200, tol=1e-10, mintol=1e-14, restart_every=500000, alpha_max=100)),
    ("jacobi", dict(M=True, maxiter=200, tol=1e-10, mintol=1e-14, restart_every=500000, alpha_max=100)),
    ("clamped", dict(M=False, maxiter=25, tol=1e-10, mintol=1e-14, restart_every=7, alpha_max=0.01)),
    ("dirstop", dict(M=False, maxiter=200, tol=1e-14, mintol=2e-4, restart_every=500000, alpha_max=100)),
    ("restart_jacobi", dict(M=True, maxiter=40, tol=1e-9, mintol=1e-14, restart_every=5, alpha_max=100)),
    ("tiny_step", dict(M=False, maxiter=12, tol=1e-10, mintol=1e-14, restart_every=500000, alpha_max=5e-7)),
])
def test_reference_cg_trace(golden_dir, tag, kw):
    g = np.load(os.path.join(golden_dir, "cg_trace.npz"))
    kw = dict(kw)
    M = g["Minv"] if kw.pop("M") else None
    trace = []
    x, info = O.reference_cg(g["A"], g["b"], M=M, callback=lambda xk: trace.append(xk.copy()), **kw)
    assert info == int(g[f"{tag}_info"])
    assert len(trace) == len(g[f"{tag}_trace"])
    assert np.allclose(np.array(trace), g[f"{tag}_trace"], rtol=1e-9, atol=1e-13)
    assert np.allclose(x, g[f"{tag}_x"], rtol=1e-9, atol=1e-13)
