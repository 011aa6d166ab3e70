#!/usr/bin/env python3
"""Small driver for rocprofv3 runs: builds the bench lattice, then launches the K*p kernel and full PCG iterations a
few times (HIP-event timed by the library).  Usage under the profiler:
    rocprofv3 --kernel-trace --stats -d OUT -o NAME --output-format csv -- python3 tools/profile_kernels.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d OUT -o NAME --output-format csv -- python3 tools/profile_kernels.py
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pylatticedso_amd import _capi, lattice_arrays as LA  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells", type=int, default=50)
ap.add_argument("--geom", default="Octet")
ap.add_argument("--radius", type=float, default=0.03)
ap.add_argument("--kernel", type=int, default=0)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--precond", type=int, default=3)
ap.add_argument("--palette", type=int, default=1)
args = ap.parse_args()
n = args.cells
lat = LA.generate((1, 1, 1), (n, n, n), [args.geom], [args.radius])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
f = np.zeros((lat.n_nodes, 6))
f[lat.node_xyz[:, 0] == float(n), 2] = -0.1
d = _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                     spmv_kernel=args.kernel, precond=args.precond, palette=args.palette)
d.set_bc(fixed, None, f)
d.assemble()
d.assemble()
d.assemble_bsr(False)
out = {"spmv_ms": d.time_kernel(0, args.reps), "pcg_iter_ms": d.time_kernel(3, args.reps),
       "record_ms": d.time_kernel(1, args.reps), "bsr_ms": d.time_kernel(2, 3), "bytes": d.algorithmic_bytes(),
       "struts": lat.n_beams, "nodes": lat.n_nodes}
print(json.dumps(out))
