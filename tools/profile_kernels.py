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
ap.add_argument("--graded", type=int, default=0,
                help="1: every cell gets its own radius (smooth non-periodic field): > 10^5 distinct strut records, "
                     "so K*p streams per-strut records whatever --palette says")
ap.add_argument("--precision", type=int, default=0)
ap.add_argument("--compact", type=int, default=0, help="-1: stream 64-byte records instead of the 40-byte compact ones")
args = ap.parse_args()
n = args.cells
override = None
if args.graded:
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    override = (args.radius * (0.8 + 0.4 * (0.5 + 0.5 * np.sin(0.113 * i + 0.271 * j + 0.419 * k)))).reshape(-1, 1)
lat = LA.generate((1, 1, 1), (n, n, n), [args.geom], [args.radius], cell_radii_override=override)
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
f = np.zeros((lat.n_nodes, 6))
f[lat.node_xyz[:, 0] == float(n), 2] = -0.1
d = _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                     spmv_kernel=args.kernel, precond=args.precond, palette=args.palette, precision=args.precision,
                     compact_records=args.compact)
d.set_bc(fixed, None, f)
d.assemble()
d.assemble()
d.assemble_bsr(False)
out = {"spmv_ms": d.time_kernel(0, args.reps), "pcg_iter_ms": d.time_kernel(3, args.reps),
       "record_ms": d.time_kernel(1, args.reps), "bsr_ms": d.time_kernel(2, 3), "bytes": d.algorithmic_bytes(),
       "struts": lat.n_beams, "nodes": lat.n_nodes, "distinct_radii": int(len(np.unique(lat.beam_radius))),
       "palette": args.palette, "graded": args.graded}
if args.precond >= 2:
    out.update(spmv_f32_ms=d.time_kernel(7, args.reps), pcg_iter_p1_ms=d.time_kernel(8, args.reps),
               pcg_iter_p2_ms=d.time_kernel(9, args.reps))
print(json.dumps(out))
