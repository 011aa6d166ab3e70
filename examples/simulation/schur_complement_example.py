"""Condense one hybrid BCC + Hybrid1 unit cell onto its boundary nodes on the GPU and report what came out
(counterpart of the reference's schur_complement_example.py, which prints the raw matrix)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "src"))

from pyLatticeSim.lattice_sim import LatticeSim                   # noqa: E402
from pyLatticeSim.utils_schur import get_schur_complement         # noqa: E402

cell = LatticeSim("simulation/hybrid_cell_simulation")
S = get_schur_complement(cell)
n_boundary = S.shape[0] // 6
eig = np.linalg.eigvalsh(0.5 * (S + S.T))
print(f"{cell.geom_types} cell, radii {cell.radii}: {cell.get_number_beams()} struts condensed onto {n_boundary} "
      f"boundary nodes -> S is {S.shape[0]} x {S.shape[1]}")
print(f"asymmetry {np.abs(S - S.T).max() / np.abs(S).max():.1e}; zero-energy modes (6 rigid-body motions per "
      f"connected part of the cell): {np.sum(np.abs(eig) < 1e-9 * eig.max())}; largest eigenvalue {eig.max():.4e}")
np.set_printoptions(precision=4, linewidth=160)
print("leading 6 x 6 block (node 0):\n", S[:6, :6])
