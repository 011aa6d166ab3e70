"""solve_DDM beyond the dense limit of the assembled-Schur preconditioner: CG preconditioned by the diagonal (precond = 1)
and by the inverted 6 x 6 node blocks (precond = 3) of the assembled matrix, BCC cantilevers of growing size.
Usage (GPU box): python tools/experiments/ddm_block_jacobi.py [golden_dir]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd import _capi                     # noqa: E402
from pylatticedso_amd.lattice_sim import LatticeSim   # noqa: E402

golden = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")
base = json.loads(str(np.load(os.path.join(golden, "ddm_bcc_4x2x2.npz"))["preset_json"]))
for n in (16, 24, 32):
    p = json.loads(json.dumps(base))
    p["geometry"]["number_of_cells"] = dict(x=n, y=n, z=n)
    L = LatticeSim(p, enable_domain_decomposition_solver=True, data_roots=[golden])
    L.set_cell_radii(0.034 + 0.03 * L.lattice.cell_pos[:, 0] / (n - 1.0))
    L.ddm_model()
    cb = L.cell_boundary_nodes()
    n_nodes = L.max_index_boundary + 1
    bn = L._boundary_nodes_by_index()
    fixed, f = L.fixed_DOF[bn], L.applied_force[bn]
    for pre in (1, 3):
        with _capi.HipLattice.ddm(n_nodes, L.index_boundary[cb], L.schur_complements, L.cell_schur_index, precond=pre,
                                  check_every=0) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            dev.solve(rtol=1e-6, max_iter=20000, download=False)
            st = dev.solve(rtol=1e-6, max_iter=20000, download=False)
            print(f"{n}^3 cells {6 * n_nodes:7d} dofs  precond {pre}: {st['iterations']:5d} iterations  solve {st['ms_solve']:7.2f} ms  "
                  f"set-up {st['ms_assembly']:.3f} ms  ({1e3 * st['ms_solve'] / st['iterations']:.1f} us per iteration)", flush=True)
