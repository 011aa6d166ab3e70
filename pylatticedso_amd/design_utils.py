"""The two I/O helpers of ``pyLatticeDesign.utils`` that the reference's optimisation example imports (utils.py:132-425):
a pickle of the design lattice and the Grasshopper JSON.  Array-backed: the pickle holds the arrays of the design lattice
(not an object graph), the JSON has the reference's keys (strut end coordinates per cell, radii, box, relative density)."""
from __future__ import annotations

import json
import os
import pickle
from pathlib import Path

import numpy as np

_OUT = Path(__file__).resolve().parent.parent / "data" / "outputs" / "saved_lattice_file"


def save_lattice_object(lattice, file_name: str = "LatticeObject") -> str:
    """utils.py:132-361 (base ``Lattice`` state only): node coordinates, struts, radii, types, cell membership, parameters."""
    lat = lattice.lattice
    state = {"_pickle_format": "pylatticedso_amd.arrays.v1", "name_lattice": getattr(lattice, "name_lattice", None),
             "cell_size": (lattice.cell_size_x, lattice.cell_size_y, lattice.cell_size_z),
             "number_of_cells": (lattice.num_cells_x, lattice.num_cells_y, lattice.num_cells_z),
             "geom_types": list(lattice.geom_types), "radii": list(lattice.radii),
             "node_xyz": lat.node_xyz, "beam_conn": lat.beam_conn, "beam_radius": lat.beam_radius,
             "beam_type": lat.beam_type, "beam_cell0": lat.beam_cell0, "cell_pos": lat.cell_pos,
             "cell_coord": lat.cell_coord, "cell_size_per_cell": lat.cell_size, "cell_radii": lat.cell_radii}
    _OUT.mkdir(parents=True, exist_ok=True)
    path = _OUT / (file_name if str(file_name).endswith(".pkl") else f"{file_name}.pkl")
    with open(path, "wb") as fh:
        pickle.dump(state, fh)
    print(f"Lattice pickle saved successfully to {path}")
    return str(path)


def save_JSON_to_Grasshopper(lattice, nameLattice: str = "LatticeObject", multipleParts: int = 1):
    """utils.py:364-425: per part, the end coordinates of every cell's struts (struts shared by cells listed per cell, as
    the reference's loop over ``cell.beams_cell`` does), their radii, the lattice box and the mean relative density."""
    lat = lattice.lattice
    _OUT.mkdir(parents=True, exist_ok=True)
    n_cells = lat.n_cells
    per_part = max(1, n_cells // multipleParts)
    paths = []
    for part in range(multipleParts):
        name = f"{nameLattice}_part{part + 1}.json" if multipleParts > 1 else f"{nameLattice}.json"
        c0, c1 = part * per_part, min((part + 1) * per_part, n_cells)
        struts = lat.cell_beam_idx[lat.cell_beam_ptr[c0]:lat.cell_beam_ptr[c1]]
        ends = lat.node_xyz[lat.beam_conn[struts]].reshape(-1, 3)            # point1, point2, point1, ...
        obj = {"nodesX": ends[:, 0].tolist(), "nodesY": ends[:, 1].tolist(), "nodesZ": ends[:, 2].tolist(),
               "radii": np.asarray(lat.beam_radius[struts], dtype=float).tolist(),
               "maxX": lattice.x_max, "minX": lattice.x_min, "maxY": lattice.y_max, "minY": lattice.y_min,
               "maxZ": lattice.z_max, "minZ": lattice.z_min, "relativeDensity": lattice.get_relative_density()}
        path = os.path.join(_OUT, name)
        with open(path, "w") as fh:
            json.dump(obj, fh)
        print(f"Saved lattice part {part + 1} to {path}")
        paths.append(path)
    return paths
