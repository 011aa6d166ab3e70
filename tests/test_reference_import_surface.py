"""The reference's import paths resolve (src/pyLatticeSim, src/pyLatticeDesign, src/pyLatticeOpti) and presets are
looked up under data/inputs/preset_lattice like utils.open_lattice_parameters does (utils.py:111-130)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "src"))


def test_imports_and_preset_lookup():
    from pyLatticeSim.lattice_sim import LatticeSim
    from pyLatticeSim.utils_simulation import solve_FEM_FenicsX          # noqa: F401
    from pyLatticeSim.utils_schur import get_schur_complement            # noqa: F401
    from pyLatticeSim.export_simulation_results import exportSimulationResults   # noqa: F401
    from pyLatticeDesign.plotting_lattice import LatticePlotting         # noqa: F401
    from pyLatticeOpti.lattice_opti import LatticeOpti                   # noqa: F401
    L = LatticeSim("simulation/simulation_beam_flexion")
    # 432 struts -> 1 288 penalised segments, as in the reference: len(lattice.beams) counts the
    # segments once the joints are penalised, 166 design nodes + 856 penalisation points
    assert (L.lattice.n_cells, L.lattice.n_beams, L.lattice.n_nodes) == (54, 432, 166)
    assert (L.get_number_cells(), L.get_number_beams(), L.get_number_nodes()) == (54, 1288, 166 + 856)
    assert L.material_name == "VeroClear" and L.is_penalized
    assert int((L.penalized.seg_len > 0).sum()) == 1288
    with pytest.raises(FileNotFoundError):
        LatticeSim("simulation/does_not_exist")


@pytest.mark.gpu
def test_simulation_example_end_to_end(tmp_path):
    from pyLatticeSim.lattice_sim import LatticeSim
    from pyLatticeSim.utils_simulation import solve_FEM_FenicsX
    from pyLatticeSim.export_simulation_results import exportSimulationResults
    L = LatticeSim("simulation/simulation_beam_flexion")
    sol, model = solve_FEM_FenicsX(L)
    assert len(sol) == 572 and np.isfinite(sol).all()
    # prescribed displacement is honoured and the loaded edge moves in +Y
    assert np.allclose(L.displacement_vector[L.fixed_DOF], np.where(L.fixed_DOF, model._ubar, 0)[L.fixed_DOF])
    ex = exportSimulationResults(model, "t", out_dir=str(tmp_path))
    ex.export_displacement_rotation()
    ex.export_reaction_force()
    path = ex.export_finalize()
    txt = open(path).read().split("\n")
    # the file holds the whole penalised model: 166 lattice nodes + 856 penalisation points, 1 288 segments
    assert "POINTS 1022 double" in txt and "LINES 1288 3864" in txt and "VECTORS displacement double" in txt
    i0 = txt.index("VECTORS displacement double") + 1
    disp = np.array([[float(v) for v in ln.split()] for ln in txt[i0:i0 + 1022]])
    assert np.allclose(disp[:166], model.u[:, :3], rtol=1e-10, atol=1e-14)
    assert np.abs(disp[166:]).max() > 0.1 * np.abs(disp[:166]).max()       # node_mod points carry real displacements
    radius = np.array([float(v) for v in txt[txt.index("SCALARS radius double 1") + 2:][:1288]])
    assert set(np.round(radius, 12)) == {0.1, 0.15}
    # the same data as VTU + PVD (what the reference's dolfinx VTKFile writes), plus the reference's per-element (DG0) fields:
    # local frame, section forces and moments in it
    ex2 = exportSimulationResults(model, "t2", out_dir=str(tmp_path))
    ex2.full_export()
    import xml.etree.ElementTree as ET
    pvd = ET.parse(ex2.pvd_path).getroot()
    vtu_name = pvd.find("Collection/DataSet").attrib["file"]
    piece = ET.parse(str(tmp_path / vtu_name)).getroot().find("UnstructuredGrid/Piece")
    assert piece.attrib == {"NumberOfPoints": "1022", "NumberOfCells": "1288"}
    cell = {d.attrib["Name"]: np.array(d.text.split(), float).reshape(1288, -1) for d in piece.find("CellData")}
    assert {"radius", "beam_mod", "type_beam", "Moment", "Forces", "a1", "a2", "t"} <= set(cell)
    tt, a1, a2 = cell["t"], cell["a1"], cell["a2"]
    assert np.allclose(np.einsum("ij,ij->i", tt, a1), 0, atol=1e-12) and np.allclose(np.cross(tt, a1), a2, atol=1e-12)
    # section forces against the oracle's element: for every design strut the end force of the condensed element, rotated
    # into the local frame of its middle segment
    from oracle import timoshenko_oracle as O
    from pylatticedso_amd.views import _tables
    lat, pen, t = L.lattice, L.penalized, _tables(L)
    mid = np.flatnonzero(~t.beam_mod)                                   # the un-penalised (middle) segments
    par = t.beam_parent[mid]
    worst = 0.0
    for k in range(0, len(mid), 37):
        b = par[k]
        A, B = lat.beam_conn[b]
        sc = O.condensed_beam(lat.beam_radius[b], pen.seg_len[b], pen.seg_nsub[b], L.young_modulus, L.poisson_ratio)
        d = lat.node_xyz[B] - lat.node_xyz[A]
        fA, fB = O.beam_apply(sc, d, model.u[A], model.u[B])
        sgn = np.sign(tt[mid[k]] @ d)
        F_loc = sgn * np.array([fB[:3] @ tt[mid[k]], fB[:3] @ a1[mid[k]], fB[:3] @ a2[mid[k]]])
        worst = max(worst, np.abs(F_loc - cell["Forces"][mid[k]]).max() / max(np.abs(cell["Forces"]).max(), 1e-300))
    assert worst < 1e-9
    # free lattice nodes are in equilibrium: the section forces of the segments meeting there cancel (global components)
    Fg = cell["Forces"][:, :1] * tt + cell["Forces"][:, 1:2] * a1 + cell["Forces"][:, 2:3] * a2
    net = np.zeros((1022, 3))
    np.add.at(net, t.beam_conn[:, 1], Fg)           # +t side of the cut at the segment's second node ...
    np.add.at(net, t.beam_conn[:, 0], -Fg)          # ... -t side at its first
    free = ~L.fixed_DOF[:, :3].any(axis=1) & ~(L.applied_force[:, :3] != 0).any(axis=1)
    # (to the solver's tolerance: ||r|| <= 1e-8 ||b|| leaves nodal residuals of that order, 1.0e-8 ... 1.3e-8 of the largest
    # section force from run to run - the LDS accumulation order of K*p is not fixed)
    assert np.abs(net[:166][free]).max() < 1e-7 * np.abs(Fg).max()
