"""Lazy Cell / Beam / Point views (pylatticedso_amd/views.py) against the object graph the reference itself built
(tests/golden/lattice_*.npz dumped from the running reference: nodes by index, beams by index, per-cell membership)."""
import glob
import json
import os

import numpy as np
import pytest

from pylatticedso_amd.lattice_sim import LatticeSim

CASES = sorted(os.path.basename(f)[len("lattice_"):-4]
               for f in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "lattice_*.npz")))


def g_has_copies(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"lattice_{name}.npz"))["beam_dup"].any()


def _load(golden_dir, name, **kw):
    g = np.load(os.path.join(golden_dir, f"lattice_{name}.npz"))
    return g, LatticeSim(json.loads(str(g["preset_json"])), **kw)


@pytest.mark.parametrize("compat", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_nodes_match_the_reference_object_graph(golden_dir, name, compat):
    g, L = _load(golden_dir, name, reference_compat=compat)
    nodes = L.nodes
    assert len(nodes) == L.get_number_nodes() == len(g["node_xyz"])
    xyz = np.array([p.coordinates for p in nodes])
    assert np.array_equal(xyz, g["node_xyz"])                      # same index -> same point, bit-exact (pen points too)
    assert np.array_equal([p.node_mod for p in nodes], g["node_mod"])
    assert [p.index for p in nodes] == list(range(len(nodes)))
    N = L.lattice.n_nodes
    # (cubichybrid1: the reference's boundary conditions on that lattice differ from run to run, tests/test_host_lattice.py)
    if "periodic" not in name and name != "cubichybrid1_2x2x2" and (
            compat or not (g["node_fixed"][N:].any() or np.any(g["node_force"][N:] != 0))):
        assert np.array_equal(np.array([p.fixed_DOF for p in nodes]).astype(np.int8), g["node_fixed"])
        assert np.allclose(np.array([p.applied_force for p in nodes]), g["node_force"], rtol=1e-15, atol=0)
        assert np.array_equal(np.array([p.displacement_vector for p in nodes]), g["node_ubar"])
    # (penalisation points of struts lying in a cell face - Octet, Cubic, Kelvin, Auxetic, ... - get a boundary index in the
    # reference, whose per-cell copies of such struts the fixture flags as beam_dup: the shared-strut corner of DESIGN.md
    # section 2, defect 1; design nodes are compared everywhere)
    m = slice(0, N) if g["beam_dup"].any() and not compat else slice(None)
    assert np.array_equal([p.index_boundary is not None for p in nodes][m], (g["node_index_boundary"] >= 0)[m])
    if not name.startswith("sym_"):      # (the reference tags a symmetric lattice before its twins exist)
        assert np.array_equal([-1 if p.tag is None else p.tag for p in nodes[:N]], g["node_tag"][:N])


@pytest.mark.parametrize("name", CASES)
def test_beams_and_cells_match_the_reference_object_graph(golden_dir, name):
    # struts shared by several cells: the reference keeps one penalised copy per owner cell - the
    # reference_compat model; the default model has every segment once
    if name.startswith("sym_"):
        pytest.skip("apply_symmetry's per-cell copies of un-penalised struts: one entry with a count here "
                    "(lattice.extras['design_mult'], held to the reference's beam list by tests/test_host_lattice.py)")
    g, L = _load(golden_dir, name, reference_compat=bool(g_has_copies(golden_dir, name)))
    beams = L.beams
    assert len(beams) == L.get_number_beams() == len(g["beam_conn"])
    conn = np.array([[b.point1.index, b.point2.index] for b in beams])
    # same order (Beam.index); point1 / point2 may be swapped: the built-in unit cells are generated from their
    # crystallographic description, not copied from the reference's geometry tables, so a strut may run the other way
    assert np.array_equal(np.sort(conn, axis=1), np.sort(g["beam_conn"], axis=1))
    assert np.array_equal([b.radius for b in beams], g["beam_radius"])
    assert np.array_equal([b.beam_mod for b in beams], g["beam_mod"])
    assert np.array_equal([b.type_beam for b in beams], g["beam_type"])
    assert np.array_equal([b.length for b in beams], g["beam_length"])
    assert np.array_equal([b.cell_belongings[0].index for b in beams], g["beam_cell0"])
    cells = L.cells
    assert len(cells) == len(g["cell_pos"])
    for c, cell in enumerate(cells):
        assert cell.pos == list(g["cell_pos"][c]) and cell.coordinate == list(g["cell_coord"][c])
        # which of the identical copies of a segment sits in which owner cell is decided by id()-hashed set order in
        # the reference: compare through the first index of every group of copies
        first = np.arange(len(conn))
        same = (np.sort(conn, axis=1)[1:] == np.sort(conn, axis=1)[:-1]).all(axis=1) & (g["beam_radius"][1:] == g["beam_radius"][:-1])
        for i in np.flatnonzero(same) + 1:
            first[i] = first[i - 1]
        mine = sorted(first[b._i] for b in cell.beams_cell)
        ref = sorted(first[g["cell_beam_idx"][g["cell_beam_ptr"][c]:g["cell_beam_ptr"][c + 1]]])
        assert mine == ref
        assert len(cell.points_cell) == g["cell_npoints"][c] and len(cell.beams_cell) == g["cell_nbeams"][c]
        if c > 3:
            break


def test_view_behaviour():
    L = LatticeSim({"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 2, "y": 1, "z": 1},
                                 "radii": [0.05], "geom_types": ["BCC"]},
                    "simulation_parameters": {"enable": False, "material": "VeroClear"}})
    assert (len(L.cells), len(L.beams), len(L.nodes)) == (2, 16, 14) and L.get_number_beams() == 16
    c0 = L.cells[0]
    assert c0.pos == [0, 0, 0] and c0.size == [1.0, 1.0, 1.0] and c0.center_point == [0.5, 0.5, 0.5]
    assert c0.radii == [0.05] and c0.geom_types == ["BCC"] and len(c0.corner_coordinates) == 8
    assert c0.boundary_box == [0.0, 1.0, 0.0, 1.0, 0.0, 1.0] and c0.volume == 1.0
    b = c0.beams_cell[0]
    assert b.length == round(np.sqrt(0.75), 4) and np.isclose(b.volume, np.pi * 0.05 ** 2 * b.length)
    assert abs(c0.relative_density - 8 * np.pi * 0.05 ** 2 * b.length) < 1e-12
    assert b in L.beams and b == L.beams[b._i] and b.cell_belongings == [c0]
    # Tests/Beam_test.py / Point_test.py style checks
    p, q = b.point1, b.point2
    assert p != q and p == L.nodes[p.index] and hash(p) == hash(L.nodes[p.index])
    assert np.isclose(p.distance_to(q), np.sqrt(0.75)) and (q - p) == [q.x - p.x, q.y - p.y, q.z - p.z]
    mid = [pt for pt in L.nodes if pt.coordinates == (1.0, 0.0, 0.0)][0]
    assert len(mid.cell_belongings) == 2 and len(mid.connected_beams) == 2 and mid.is_on_boundary(L.get_lattice_boundary_box())
    centre = [pt for pt in L.nodes if pt.coordinates == (0.5, 0.5, 0.5)][0]
    assert len(centre.connected_beams) == 8 and centre.index_boundary is None and centre.tag is None
    assert b.is_point_on_beam(centre) is False
    # rows are live: writing through a view changes the simulation arrays, like the reference's lists
    mid.fix_DOF([0, 2])
    mid.set_applied_force([1.5], [1])
    mid.displacement_vector[2] = 0.25
    assert L.fixed_DOF[mid.index].tolist() == [True, False, True, False, False, False]
    assert L.applied_force[mid.index, 1] == 1.5 and L.displacement_vector[mid.index, 2] == 0.25
    assert mid.deformed_coordinates == (1.0, 0.0, 0.0 + 0.25 * 5.0)
    mid.set_reaction_force([1, 2, 3, 4, 5, 6])
    mid.set_reaction_force([1, 2, 3, 4, 5, 6])
    assert L.reaction_force_vector[mid.index].tolist() == [2, 4, 6, 8, 10, 12]
    assert np.isclose(mid.calculate_point_energy(), (6 + 0) * 0.25)
    with pytest.raises(ValueError):
        mid.set_reaction_force([1, 2])
    with pytest.raises(IndexError):
        L.cells[2]
    # a new radius for one cell goes through the simulation (Cell.change_beam_radius)
    L.cells[1].change_beam_radius([0.08])
    assert L.cells[1].radii == [0.08] and L.cells[0].radii == [0.05]
    assert sorted(set(np.round(L.lattice.beam_radius, 12))) == [0.05, 0.08]
