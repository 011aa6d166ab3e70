"""The plain-C oracle (cpu_baseline port) against the numpy oracle, which is pinned by the reference's goldens."""
import json
import os

import numpy as np
import scipy.sparse.linalg as spla

from oracle import c_oracle, timoshenko_oracle as O
from pylatticedso_amd.lattice_sim import LatticeSim

E, NU = 1013.0, 0.3


def _case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"lattice_{name}.npz"))
    L = LatticeSim(json.loads(str(g["preset_json"])))
    return L


def test_c_condense_and_spmv_match_numpy_oracle(golden_dir):
    L = _case(golden_dir, "bccoctet_2x2x2")
    lat, pen = L.lattice, L.penalized
    sc = c_oracle.condense(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
    ref = np.array([O.condensed_beam(r, l, n, E, NU) for r, l, n in zip(lat.beam_radius, pen.seg_len, pen.seg_nsub)])
    assert np.allclose(sc, ref, rtol=1e-13)
    assert np.allclose(c_oracle.condense_unique(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU), sc, rtol=0)
    K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, ref)
    x = np.random.default_rng(1).standard_normal(6 * lat.n_nodes)
    y = c_oracle.spmv(lat.node_xyz, lat.beam_conn, sc, x).ravel()
    assert np.linalg.norm(y - K @ x) / np.linalg.norm(K @ x) < 1e-13


def test_c_pcg_matches_direct_solve_of_submeshed_model(golden_dir):
    """End-to-end: C Jacobi-PCG on condensed struts == sparse-direct solve of the reference-faithful sub-meshed
    model (every gmsh sub-node an unknown), on the reference's own simulation_beam_flexion boundary conditions."""
    g = np.load(os.path.join(golden_dir, "lattice_bcc_6x3x3_flexion.npz"))
    L = _case(golden_dir, "bcc_6x3x3_flexion")
    lat, pen = L.lattice, L.penalized
    sc = c_oracle.condense(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
    f = np.zeros((lat.n_nodes, 6))
    f[:, :3] = L.applied_force[:, :3]
    u, it, rel = c_oracle.pcg(lat.node_xyz, lat.beam_conn, sc, L.fixed_DOF, L.displacement_vector, f, rtol=1e-12)
    assert it > 0 and rel < 1e-12
    # faithful model straight from the reference's dumped segments
    keep = ~g["beam_dup"]
    K, nv = O.assemble_submeshed(g["node_xyz"], g["beam_conn"][keep], g["beam_radius"][keep], E, NU, 0.05)
    fixed = np.zeros((nv, 6), bool)
    ubar = np.zeros((nv, 6))
    ff = np.zeros((nv, 6))
    n0 = len(g["node_xyz"])
    fixed[:n0] = g["node_fixed"] != 0
    ubar[:n0] = g["node_ubar"]
    ff[:n0, :3] = g["node_force"][:, :3]
    uref = O.solve_dirichlet(K, fixed, ubar, ff).reshape(-1, 6)[:lat.n_nodes]
    assert np.linalg.norm(u - uref) / np.linalg.norm(uref) < 1e-8
