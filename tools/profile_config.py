#!/usr/bin/env python3
"""Driver for rocprofv3 passes over ONE BASELINE configuration's K*p as the solve applies it (pl_time_kernel 10: both
passes under node elimination, fp32-stored operands with precision 1):
    rocprofv3 --pmc SQ_INSTS_VALU ... --kernel-trace -d OUT -o c --output-format csv -- python3 tools/profile_config.py --config 2
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pylatticedso_amd import _capi, lattice_arrays as LA  # noqa: E402

CONFIGS = {1: ((50, 50, 50), ["Octet"], [0.03], 0), 2: ((100, 100, 100), ["BCC"], [0.05], 0),
           4: ((200, 200, 50), ["BCC", "Octet"], [0.04, 0.03], 6)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1, choices=[1, 2, 4])
ap.add_argument("--precision", type=int, default=0)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--which", type=int, default=10)
args = ap.parse_args()
cells, geom, radii, tm = CONFIGS[args.config]
lat = LA.generate((1, 1, 1), cells, geom, radii)
pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
f = np.zeros((lat.n_nodes, 6))
f[lat.node_xyz[:, 0] == float(cells[0]), 2] = -0.1
d = _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                     precond=3, palette=1, precision=args.precision, tile_modes=tm)
d.set_bc(fixed, None, f)
d.assemble()
print(json.dumps({"config": args.config, "precision": args.precision, "operator_ms": d.time_kernel(args.which, args.reps),
                  "struts": lat.n_beams, "version": _capi.load_library().pl_version().decode()}))
