import sys, os, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from test_gpu_opti import _preset
from pylatticedso_amd.lattice_opti import LatticeOpti
n=12
L = LatticeOpti(_preset(optimization_parameters={"type": "unit_cell"}, geometry={"number_of_cells": {"x": n, "y": n, "z": n}}))
L._device = L.device_model(precond=int(sys.argv[1]) if len(sys.argv)>1 else 3, palette=int(sys.argv[2]) if len(sys.argv)>2 else 1)
c = L._cell_center
r = np.clip(0.05 + 0.03 * (np.sin(2 * np.pi * c[:, 0] / 8) * np.cos(2 * np.pi * c[:, 1] / 8)
                           + np.sin(2 * np.pi * c[:, 1] / 8) * np.cos(2 * np.pi * c[:, 2] / 8)
                           + np.sin(2 * np.pi * c[:, 2] / 8) * np.cos(2 * np.pi * c[:, 0] / 8)) / 1.5, 0.01, 0.1)
theta = np.asarray(L.normalize_optimization_parameters(list(r)))
L.fem_rtol = 1e-13
L.objective(list(theta))
g = np.asarray(L.gradient(list(theta)))
for i in (0, n**3//2+5, n**3-1):
    print("param", i, "r", r[i], "g", g[i])
    for h in (4e-3, 2e-3, 1e-3, 5e-4, 2e-4, 1e-4):
        tp, tm = theta.copy(), theta.copy(); tp[i]+=h; tm[i]-=h
        fp = L.objective(list(tp)); fm = L.objective(list(tm))
        print("   h", h, "fd", (fp-fm)/(2*h), "its", L._model.stats["iterations"], "rel", L._model.stats["rel_residual"])
