"""Reference-derived fixture for the periodic homogenisation: the homogenised 6x6 matrix of the BCC / Hybrid1 /
Hybrid4 cell at every radius of the reference's committed dolfinx / PETSc Schur complements
(data/outputs/schur_complement/Schur_complement_{BCC,Hybrid1,Hybrid4}.npz, of which tests/golden/schur_*.npz hold every
second radius together with the boundary-node coordinates in the reference's node order, dumped by make_golden.py).

The periodic fluctuation problem of HomogenizedCell (homogenization_cell.py:200-252,309-331,405-436) only involves the cell
through its Schur complement on the boundary nodes (oracle.homogenize_from_schur): every interior dof is free and unloaded.
So these matrices are what the REFERENCE's own condensed operators say the homogenised matrix is - no dolfinx_mpc needed.
What stays unpinned: the dolfinx_mpc machinery itself (how it pairs slave and master dofs, its first-vertex-per-tag
reaction sum); the formulas are the reference's.

    python tests/golden/make_homogenization_fixture.py        ->  tests/golden/homogenized_from_schur.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import timoshenko_oracle as O  # noqa: E402

out = {}
for geom in ("BCC", "Hybrid1", "Hybrid4"):
    g = np.load(os.path.join(HERE, f"schur_{geom}.npz"))
    xyz = g["boundary_node_xyz"]
    Cs = []
    for S in g["schur_matrices"]:
        C, C_raw = O.homogenize_from_schur(S, xyz)
        assert np.linalg.norm(C_raw - C_raw.T) < 1e-8 * np.linalg.norm(C), geom
        Cs.append(C)
    out[f"{geom}_radius"] = g["radius_values"].ravel()
    out[f"{geom}_C"] = np.array(Cs)
    out[f"{geom}_boundary_node_xyz"] = xyz
    print(geom, "radii", out[f"{geom}_radius"], "C11", [float(c[0, 0]) for c in Cs])
np.savez_compressed(os.path.join(HERE, "homogenized_from_schur.npz"), **out)
