"""Where do the occasional ~75 ms of a repeated solve_DDM at 32^3 cells go?  Times the pieces of the call 20 times each."""
import gc, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_sim import LatticeSim  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": n, "y": n, "z": n},
                       "radii": [0.05], "geom_types": ["BCC"]},
          "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False,
                                    "DDM": {"enable_preconditioner": True, "preconditioner_type": "exact", "max_iterations": 20000,
                                            "schur_complement_computation": {"type": "RBF", "precision_greedy": 1e-6}}},
          "boundary_conditions": {
              "Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"], "Value": [0] * 6}},
              "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
L = LatticeSim(preset, enable_domain_decomposition_solver=True, data_roots=[os.path.join(ROOT, "tests", "golden")])
L.solve_DDM()
dev = L.ddm_model()
bn = L._boundary_nodes_by_index()
fixed = L.fixed_DOF[bn]
ubar = np.where(fixed, L.displacement_vector[bn], 0.0)
f = L.applied_force[bn]
def T(fn, reps=20):
    out = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); out.append(round(1e3 * (time.perf_counter() - t), 2))
    return out
res = {}
res["solve_only"] = T(lambda: dev.solve(rtol=1e-6, max_iter=20000, raise_on_noconv=False))
res["solve_only_device_ms"] = []
for _ in range(10):
    dev.solve(rtol=1e-6, max_iter=20000, raise_on_noconv=False); res["solve_only_device_ms"].append(round(dev.last_stats["ms_solve"], 2))
res["set_bc"] = T(lambda: dev.set_bc(fixed, ubar, f))
def sa():
    dev.set_bc(fixed, ubar, f); dev.assemble()
res["set_bc+assemble"] = T(sa)
res["spmv"] = T(lambda: dev.spmv(ubar))
res["solve_DDM"] = T(lambda: L.solve_DDM(), 12)
gc.disable()
res["solve_DDM_gc_off"] = T(lambda: L.solve_DDM(), 12)
print(json.dumps(res))
import cProfile, pstats, io
gc.enable()
pr = cProfile.Profile()
pr.enable()
for _ in range(12):
    L.solve_DDM()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
print(s.getvalue())
