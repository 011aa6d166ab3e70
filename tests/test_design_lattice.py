"""Design-side API surface: ``pyLatticeDesign.lattice.Lattice`` behaves as the reference's Tests/Lattice_test.py expects
(JSON file given by absolute path, cell counts, sizes, bounding box, counts, relative density, repr), every import line
of the reference's examples/simulation and examples/optimization scripts resolves under src/ (the kriging density
surrogate, pyLatticeOpti.surrogate_model_relative_densities, is out of scope), and the small helpers those scripts
import work.  CPU only."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "src"))

BCC2 = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 2, "y": 2, "z": 2},
                     "radii": [0.05], "geom_types": ["BCC"]}}
BCC1 = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": 1, "y": 1, "z": 1},
                     "radii": [0.05], "geom_types": ["BCC"]}}


@pytest.fixture
def json_file(tmp_path):
    def make(cfg):
        p = tmp_path / "lattice.json"
        p.write_text(json.dumps(cfg))
        return str(p)
    return make


def test_lattice_like_the_reference_tests(json_file):
    from pyLatticeDesign.lattice import Lattice
    lat = Lattice(json_file(BCC2))                         # Tests/Lattice_test.py:55-76
    assert len(lat.cells) == 8 and (lat.num_cells_x, lat.num_cells_y, lat.num_cells_z) == (2, 2, 2)
    assert (lat.size_x, lat.size_y, lat.size_z) == (2.0, 2.0, 2.0)
    assert (lat.x_min, lat.x_max, lat.y_min, lat.y_max, lat.z_min, lat.z_max) == (0.0, 2.0, 0.0, 2.0, 0.0, 2.0)
    small = Lattice(json_file(BCC1))                       # :79-121
    nb, nn, rd = small.get_number_beams(), small.get_number_nodes(), small.get_relative_density()
    assert isinstance(nb, int) and isinstance(nn, int) and (nb, nn) == (8, 9)
    assert isinstance(rd, float) and 0 < rd < 1
    assert abs(rd - 8 * np.pi * 0.05 ** 2 * round(np.sqrt(0.75), 4)) < 1e-12      # Cell.relative_density, beam.py:135
    assert "Lattice" in repr(small) and "1.0" in repr(small)
    assert small == Lattice(json_file(BCC1)) and not (small == lat)
    # the base class carries no simulation layer even when the file has one
    cfg = dict(BCC2, simulation_parameters={"enable": True, "material": "VeroClear"},
               boundary_conditions={"Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X"], "Value": [0]}}})
    plain = Lattice(json_file(cfg))
    assert not plain.is_penalized and not plain.fixed_DOF.any() and plain.get_number_beams() == 64
    assert plain.get_beam_radius_min_max() == (0.05, 0.05)


def test_every_import_of_the_reference_examples_resolves():
    from pyLatticeSim.lattice_sim import LatticeSim                                             # noqa: F401
    from pyLatticeDesign.plotting_lattice import LatticePlotting
    from pyLatticeSim.utils_simulation import solve_FEM_FenicsX, get_homogenized_properties       # noqa: F401
    from pyLatticeSim.export_simulation_results import exportSimulationResults                   # noqa: F401
    from pyLatticeOpti.lattice_opti import LatticeOpti                                           # noqa: F401
    from pyLatticeDesign.lattice import Lattice                                                   # noqa: F401
    from pyLatticeSim.utils_schur import (get_schur_complement, save_schur_complement_npz,       # noqa: F401
                                          load_schur_complement_dataset)
    from pyLatticeSim.utils import create_homogenization_figure                                  # noqa: F401
    from pyLatticeSim.greedy_algorithm import reduce_basis_greedy, find_name_file_reduced_basis   # noqa: F401
    from pyLatticeDesign.utils import save_JSON_to_Grasshopper, save_lattice_object              # noqa: F401
    assert hasattr(LatticePlotting, "visualize_lattice") and hasattr(LatticePlotting, "subplot_lattice_hybrid_geometries")
    assert hasattr(LatticeOpti, "optimize_lattice") and hasattr(LatticeOpti, "reset_penalized_beams")


def test_helpers_of_the_examples(tmp_path, monkeypatch):
    from pyLatticeDesign.lattice import Lattice
    from pyLatticeDesign.utils import function_penalization_Lzone, save_JSON_to_Grasshopper, save_lattice_object
    from pyLatticeSim.utils import clear_directory, create_homogenization_figure, directional_modulus, directional_modulus_grid
    import pylatticedso_amd.design_utils as DU
    monkeypatch.setattr(DU, "_OUT", tmp_path)
    lat = Lattice(BCC2)
    path = save_JSON_to_Grasshopper(lat, "t")[0]
    obj = json.load(open(path))
    assert len(obj["radii"]) == 64 and len(obj["nodesX"]) == 128 and obj["maxX"] == 2.0
    assert abs(obj["relativeDensity"] - lat.get_relative_density()) < 1e-15
    import pickle
    st = pickle.load(open(save_lattice_object(lat, "obj"), "rb"))
    assert st["beam_conn"].shape == (64, 2) and st["geom_types"] == ["BCC"]
    assert len(os.listdir(tmp_path)) == 2
    clear_directory(str(tmp_path))
    assert os.listdir(tmp_path) == []
    # Tests/Utils_test.py: sign and the two special cases of the penalisation length
    assert function_penalization_Lzone(0.05, 90.0) == pytest.approx(0.05) and function_penalization_Lzone(0.05, 175) == 1e-7
    assert function_penalization_Lzone(0.05, 0.0) == 0.0
    # directional modulus of an isotropic compliance is E in every direction; the grid form equals the scalar one
    E, nu = 3.0, 0.3
    S = np.zeros((6, 6))
    S[:3, :3] = -nu / E
    S[np.arange(3), np.arange(3)] = 1 / E
    S[np.arange(3, 6), np.arange(3, 6)] = 2 * (1 + nu) / E
    for th, ph in ((90, 0), (37, 141), (0, 0)):
        assert np.linalg.norm(directional_modulus(S, th, ph)) == pytest.approx(E, rel=1e-12)
    grid = directional_modulus_grid(S, [0, 37, 90], [0, 141])
    assert np.allclose(grid[1, 1], directional_modulus(S, 37, 141), rtol=1e-13)
    assert create_homogenization_figure(S, plot=False, save=False) is None
