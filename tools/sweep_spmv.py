#!/usr/bin/env python3
"""Time the K*p kernel variants of libpylattice_hip on one lattice (A/B in ONE process, interleaved rounds)."""
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
ap.add_argument("--variants", default="k2:l4:r1,k2:l2:r1,k2:l8:r1,k2:l1:r1,k2:l4:r0,k1:l4:r1")
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()

n = args.cells
lat = LA.generate((1, 1, 1), (n, n, n), [args.geom], [args.radius])
pen = LA.penalize(lat, LA.compute_lzone(lat))
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
f = np.zeros((lat.n_nodes, 6))
f[lat.node_xyz[:, 0] == float(n), 2] = -0.1
devs = {}
for v in args.variants.split(","):
    opt = dict(tok[0] and (tok[0], int(tok[1:])) for tok in v.split(":"))
    d = _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3,
                         spmv_kernel=opt.get("k", 0), lanes_per_node=opt.get("l", 0), reorder=opt.get("r", 1),
                         tile_nodes=opt.get("t", 0), palette=opt.get("p", 0))
    d.set_bc(fixed, None, f)
    d.assemble()
    devs[v] = d
ab = next(iter(devs.values())).algorithmic_bytes()
res = {v: {"spmv": [], "iter": []} for v in devs}
for _ in range(args.rounds):
    for v, d in devs.items():
        res[v]["spmv"].append(d.time_kernel(0, 30))
        res[v]["iter"].append(d.time_kernel(3, 30))
for v, r in res.items():
    ms = float(np.median(r["spmv"]))
    print(json.dumps({"variant": v, "spmv_us_med": ms * 1e3, "spmv_us_min": min(r["spmv"]) * 1e3,
                      "GBps": ab["spmv"] / ms / 1e6, "frac_8TBs": ab["spmv"] / ms / 1e6 / 8000,
                      "pcg_iter_us_med": float(np.median(r["iter"])) * 1e3}))
