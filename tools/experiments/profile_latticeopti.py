import cProfile, pstats, io, sys, time, copy
sys.path.insert(0, "/root/repo")
from pylatticedso_amd.lattice_opti import LatticeOpti
from pylatticedso_amd.lattice_sim import open_lattice_parameters
preset = copy.deepcopy(open_lattice_parameters("optimization/optimization_beam_flexion"))
preset["optimization_informations"]["optimization_parameters"] = {"type": "unit_cell", "hybrid": False}
L = LatticeOpti(preset, verbose=0, convergence_plotting=False)
L.optimize_lattice()
L = LatticeOpti(preset, verbose=0, convergence_plotting=False)
pr = cProfile.Profile(); pr.enable(); t = time.time()
sol = L.optimize_lattice()
pr.disable(); print("optimize_s", time.time() - t, "nit", sol.nit, "nfev", sol.nfev)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
