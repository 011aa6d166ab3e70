"""Periodic homogenisation of one cell ON THE DEVICE (get_homogenized_properties: strut records, K*w, and the six solves
under periodic constraints through pl_set_periodic + the library's PCG) against the oracle's restatement of the reference
procedure on the sub-meshed model built from the reference's own segments, and against the matrices the reference's own
committed Schur complements give (tests/golden/homogenized_from_schur.npz)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import timoshenko_oracle as O                                      # noqa: E402
from pylatticedso_amd.homogenization_cell import HomogenizedCell, directional_modulus   # noqa: E402
from pylatticedso_amd.lattice_sim import LatticeSim                            # noqa: E402
from pylatticedso_amd.utils_simulation import get_homogenized_properties     # noqa: E402

E, NU = 1013.0, 0.3


def _oracle(g):
    keep = ~g["beam_dup"]
    K, _ = O.assemble_submeshed(g["node_xyz"], g["beam_conn"][keep], g["beam_radius"][keep], E, NU, 0.05)
    V = O.submesh_vertices(g["node_xyz"], g["beam_conn"][keep], 0.05)
    return O.homogenize_submeshed(K, V, g["node_tag"])


@pytest.mark.parametrize("name", ["bcc", "hybrid4", "bcchybrid1", "bcchybrid4"])
def test_homogenized_matrix_matches_oracle(golden_dir, name, capsys):
    g = np.load(os.path.join(golden_dir, f"lattice_{name}_1x1x1_periodic.npz"))
    L = LatticeSim(json.loads(str(g["preset_json"])))
    S, analysis = get_homogenized_properties(L)
    out = capsys.readouterr().out
    assert "Homogenized matrix" in out and "Symmetry error" in out
    C, C_raw, u_tots = _oracle(g)
    H = analysis.homogenizeMatrix
    assert np.linalg.norm(H - C) < 1e-9 * np.linalg.norm(C)
    assert analysis._symmetryError < 1e-10
    # total displacement fields on the lattice nodes (up to nothing: same anchor, same periodic groups)
    xyz = L.lattice.node_xyz
    ids = [int(np.argmin(np.abs(g["node_xyz"] - p).max(axis=1))) for p in xyz]
    for k in range(6):
        ref = u_tots[k][ids]
        assert np.linalg.norm(analysis.saveDataToExport[k] - ref) < 1e-8 * np.linalg.norm(ref)
    # engineering constants: S = inverse of the cubic stiffness with tensorial shear (G = C44 / 2)
    Hinv = np.linalg.inv(C)
    assert np.isclose(S[0, 0], Hinv[0, 0], rtol=1e-8) and np.isclose(1 / S[3, 3], C[3, 3] / 2, rtol=1e-8)
    assert np.isclose(analysis.orthotropicMatrix[0, 0], 1 / Hinv[0, 0], rtol=1e-8)
    e100 = np.linalg.norm(directional_modulus(S, 90.0, 0.0))
    assert np.isclose(e100, 1 / Hinv[0, 0], rtol=1e-8)


def test_example_preset_and_errors():
    repo_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    L = LatticeSim("simulation/hybrid_cell_simulation")
    S, analysis = get_homogenized_properties(L)
    H = analysis.homogenizeMatrix
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > 0
    assert np.allclose(np.diag(H)[:3], H[0, 0], rtol=1e-8) and np.allclose(np.diag(H)[3:], H[3, 3], rtol=1e-8)
    assert len(analysis.saveDataToExport) == 6 and analysis.saveDataToExport[0].shape == (L.lattice.n_nodes, 6)
    # periodicity of the fluctuation: u_tot - w equal on partner nodes
    from pylatticedso_amd.homogenization_cell import imposed_displacement
    fl = analysis.saveDataToExport[3] - imposed_displacement(4, L.lattice.node_xyz)
    assert np.allclose(fl, fl[analysis._master], atol=1e-12)
    p = json.load(open(os.path.join(repo_root, "data/inputs/preset_lattice/simulation/hybrid_cell_simulation.json")))
    p["geometry"]["number_of_cells"]["x"] = 2
    with pytest.raises(ValueError):
        get_homogenized_properties(LatticeSim(p))
    with pytest.raises(ValueError):
        HomogenizedCell(LatticeSim(p))


@pytest.mark.parametrize("geom,penalised", [("BCC", True), ("Hybrid1", False), ("Hybrid4", False)])
def test_homogenized_matrix_matches_the_reference_schur_complements(golden_dir, geom, penalised):
    """device <-> reference-derived fixture at every radius of the committed Schur goldens (the BCC dataset was generated
    with joint penalisation, Hybrid1 / Hybrid4 without: tests/test_gpu_parity.py::test_schur_complement_matches_dolfinx_golden)."""
    fx = np.load(os.path.join(golden_dir, "homogenized_from_schur.npz"))
    for r, C in zip(fx[f"{geom}_radius"], fx[f"{geom}_C"]):
        preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                               "radii": [float(r)], "geom_types": [geom]},
                  "simulation_parameters": {"enable": penalised, "material": "VeroClear", "periodicity": True}}
        L = LatticeSim(preset)
        analysis = HomogenizedCell(L)
        analysis.prepare_simulation()
        analysis.apply_dirichlet_for_homogenization()
        analysis.periodic_boundary_condition()
        H = analysis.solve_full_homogenization()
        # (six device solves; a case whose load -P^T K w vanishes by symmetry - the normal strains of the BCC cell - takes none)
        assert analysis.solver == "device" and len(analysis.pcg_iterations) == 6 and max(analysis.pcg_iterations) > 0
        assert int(analysis.device.last_stats["precond_used"]) == 1
        assert np.linalg.norm(H - C) < 2e-8 * np.linalg.norm(C), (geom, r, np.linalg.norm(H - C) / np.linalg.norm(C))
        L._device.close()


def test_device_and_host_solvers_of_the_periodic_problem_agree(golden_dir):
    """The six periodic solves on the device (PCG on Q K Q, pl_set_periodic) against the dense host factorisation of the
    reduced matrix used until round 4: same total displacement fields, same matrix."""
    g = np.load(os.path.join(golden_dir, "lattice_bcchybrid1_1x1x1_periodic.npz"))
    out = {}
    for solver in ("device", "host"):
        L = LatticeSim(json.loads(str(g["preset_json"])))
        a = HomogenizedCell(L, solver=solver)
        a.prepare_simulation()
        a.apply_dirichlet_for_homogenization()
        a.periodic_boundary_condition()
        out[solver] = (a.solve_full_homogenization(), a.saveDataToExport)
        L._device.close()
    assert np.linalg.norm(out["device"][0] - out["host"][0]) < 1e-9 * np.linalg.norm(out["host"][0])
    for ud, uh in zip(out["device"][1], out["host"][1]):
        assert np.linalg.norm(ud - uh) < 1e-8 * np.linalg.norm(uh)
