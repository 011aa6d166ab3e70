"""solve_DDM with the reference's assembled-Schur preconditioner on the device (pl_ddm_set_preconditioner, dense
Cholesky) against plain and Jacobi CG, on BCC cantilevers of growing size.  Usage (GPU box):
    python tools/experiments/ddm_preconditioner_scale.py [golden_dir]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd.lattice_sim import LatticeSim   # noqa: E402

golden = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
base = json.loads(str(np.load(os.path.join(golden, "ddm_bcc_4x2x2.npz"))["preset_json"]))
for ncell in [(6, 3, 3), (10, 5, 5), (16, 8, 8), (20, 10, 10)]:
    for label, pre in [("plain CG", None), ("nearest_reference", "nearest_reference"), ("exact", "exact")]:
        p = json.loads(json.dumps(base))
        p["geometry"]["number_of_cells"] = dict(zip("xyz", ncell))
        ddm = p["simulation_parameters"]["DDM"]
        ddm["max_iterations"] = 20000
        ddm["enable_preconditioner"] = pre is not None
        if pre:
            ddm["preconditioner_type"] = pre
        L = LatticeSim(p, enable_domain_decomposition_solver=True, data_roots=[golden])
        pos = L.lattice.cell_pos
        L.set_cell_radii(0.034 + 0.04 * pos[:, 0] / max(1, ncell[0] - 1) + 0.002 * (pos[:, 2] % 3))
        L.solve_DDM()                                   # first call: allocations, factor buffers
        t0 = time.perf_counter()
        xsol, info, _, b = L.solve_DDM()
        dt = time.perf_counter() - t0
        dev = L.ddm_model()
        print(f"{ncell}: {len(b):6d} free dofs  {label:18s} iterations {L.iteration:5d}  info {info}  "
              f"solve_DDM {1e3 * dt:8.1f} ms (device assemble {dev.last_stats['ms_assembly']:.1f} ms, "
              f"solve {dev.last_stats['ms_solve']:.1f} ms)", flush=True)
