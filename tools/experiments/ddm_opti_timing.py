import copy, os, sys, time, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_opti import LatticeOpti
from pylatticedso_amd.lattice_sim import open_lattice_parameters
from pylatticedso_amd.timing import timing
for cells in ((5, 1, 1), (6, 1, 6), (12, 4, 4)):
    preset = copy.deepcopy(open_lattice_parameters("optimization/optimization_DDM_surrogate"))
    preset["geometry"]["geom_types"] = ["BCC"]
    preset["geometry"]["radii"] = [0.05]
    preset["simulation_parameters"]["DDM"]["preconditioner_type"] = "exact"   # (the preset as shipped names none, which the reference refuses too)
    preset["geometry"]["number_of_cells"] = dict(x=cells[0], y=cells[1], z=cells[2])
    timing.reset()
    t0 = time.perf_counter()
    L = LatticeOpti(preset, verbose=0, convergence_plotting=False, data_roots=[os.path.join(ROOT, "tests", "golden")])
    t1 = time.perf_counter()
    sol = L.optimize_lattice()
    t2 = time.perf_counter()
    dev_s = sum(sum(v) for k, v in timing.timings.items() if k.startswith("device:"))
    print(cells, "nit", sol.nit, "nfev", sol.nfev, "construct %.3f optimize %.3f s  per it %.1f ms  device per it %.2f ms" % (t1 - t0, t2 - t1, 1e3 * (t2 - t1) / max(sol.nit, 1), 1e3 * dev_s / max(sol.nit, 1)), "obj", L.denorm_objective, sol.success, flush=True)
    top = sorted(((sum(v), k, len(v)) for k, v in timing.timings.items()), reverse=True)[:8]
    for t, k, n in top:
        print("    %8.1f ms  %5d x  %s" % (1e3 * t, n, k))
if len(sys.argv) > 1:       # cProfile of the last case
    import cProfile, pstats, io
    preset["optimization_informations"]["max_iterations"] = 15
    L = LatticeOpti(preset, verbose=0, convergence_plotting=False, data_roots=[os.path.join(ROOT, "tests", "golden")])
    pr = cProfile.Profile(); pr.enable(); L.optimize_lattice(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue())
