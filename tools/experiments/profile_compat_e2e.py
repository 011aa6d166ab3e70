"""cProfile of the drop-in call site with reference_compat=True at 50^3 Octet (what end_to_end_reference_compat_s times)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pylatticedso_amd.lattice_sim import LatticeSim
from pylatticedso_amd.utils_simulation import solve_FEM_FenicsX
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n}, "radii": [0.03], "geom_types": ["Octet"]},
          "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
          "boundary_conditions": {"Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"], "Value": [0] * 6}},
                                  "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
L = LatticeSim(preset, reference_compat=True); solve_FEM_FenicsX(L); L._device.close()       # warm
for what in ("LatticeSim", "solve_FEM_FenicsX"):
    pr = cProfile.Profile(); t = time.time(); pr.enable()
    if what == "LatticeSim":
        L = LatticeSim(preset, reference_compat=True)
    else:
        solve_FEM_FenicsX(L)
    pr.disable(); print(what, round(time.time() - t, 3), "s")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print("\n".join(s.getvalue().splitlines()[6:26]))
