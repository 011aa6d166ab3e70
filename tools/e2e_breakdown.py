#!/usr/bin/env python3
"""Wall-clock breakdown of the drop-in call site for an n^3 lattice: LatticeSim(preset) and solve_FEM_FenicsX(lattice),
printed with the timing collector (host time per call, device HIP-event times as children)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylatticedso_amd.lattice_sim import LatticeSim              # noqa: E402
from pylatticedso_amd.timing import timing                       # noqa: E402
from pylatticedso_amd.utils_simulation import solve_FEM_FenicsX  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
geom = sys.argv[2] if len(sys.argv) > 2 else "Octet"
radius = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n},
                       "radii": [radius], "geom_types": [geom]},
          "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
          "boundary_conditions": {
              "Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"],
                                         "Value": [0, 0, 0, 0, 0, 0]}},
              "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
import torch  # noqa: E402,F401  (pages the ROCm runtime in before the clock starts, as in any user session)
torch.cuda.init()
for rep in range(2):
    timing.reset()
    t0 = time.perf_counter()
    L = LatticeSim(preset)
    t1 = time.perf_counter()
    xsol, model = solve_FEM_FenicsX(L, rtol=1e-8)
    t2 = time.perf_counter()
    print(f"run {rep}: {n}^3 {geom}: LatticeSim {t1 - t0:.2f} s, solve_FEM_FenicsX {t2 - t1:.2f} s "
          f"({model.stats['iterations']} PCG iterations, device solve {model.stats['ms_solve']:.1f} ms), "
          f"{L.lattice.n_beams} struts, len(xsol) {len(xsol)}")
    L._device.close()
timing.summary(name_width=70, min_total=0.005)
