"""LatticeOpti in DDM mode on a larger lattice (n^3 BCC cells, unit_cell parameterisation): cost of objective + gradient."""
import copy, os, sys, time, cProfile, pstats, io
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_opti import LatticeOpti
from pylatticedso_amd.lattice_sim import open_lattice_parameters
from pylatticedso_amd.timing import timing
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
preset = copy.deepcopy(open_lattice_parameters("optimization/optimization_DDM_surrogate"))
preset["geometry"].update(geom_types=["BCC"], radii=[0.05], number_of_cells={"x": n, "y": n, "z": n})
preset["simulation_parameters"]["DDM"].update(preconditioner_type="exact", max_iterations=20000)
L = LatticeOpti(preset, verbose=0, convergence_plotting=False, data_roots=[os.path.join(ROOT, "tests", "golden")])
rng = np.random.default_rng(0)
theta = 0.3 + 0.3 * rng.random(L.number_parameters)
for _ in range(2):
    L.objective(list(theta)); L.gradient(list(theta)); theta = np.clip(theta + 0.005 * rng.standard_normal(len(theta)), 0, 1)
timing.reset()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(5):
    L.objective(list(theta)); L.gradient(list(theta)); theta = np.clip(theta + 0.005 * rng.standard_normal(len(theta)), 0, 1)
dt = time.perf_counter() - t0
pr.disable()
print("cells", L.lattice.n_cells, "ms per objective + gradient: %.1f" % (1e3 * dt / 5), "CG iterations", L.iteration)
for t, k, c in sorted(((sum(v), k, len(v)) for k, v in timing.timings.items()), reverse=True)[:8]:
    print("    %8.1f ms  %5d x  %s" % (1e3 * t, c, k))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue())
