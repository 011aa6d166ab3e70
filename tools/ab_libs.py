#!/usr/bin/env python3
"""A/B several builds of libpylattice_hip (one subprocess per .so, same lattice): K*p and PCG-iteration time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from pylatticedso_amd import _capi
_capi.load_library(sys.argv[1])
n = int(sys.argv[2])
d0 = np.load(sys.argv[3])
dev = _capi.HipLattice(d0["xyz"], d0["conn"], d0["rad"], d0["seg_len"], d0["seg_nsub"], 1013.0, 0.3)
fixed = np.zeros((len(d0["xyz"]), 6), np.uint8); fixed[d0["xyz"][:, 0] == 0.0] = 1
f = np.zeros((len(d0["xyz"]), 6)); f[d0["xyz"][:, 0] == float(n), 2] = -0.1
dev.set_bc(fixed, None, f); dev.assemble()
sp = [dev.time_kernel(0, 30) for _ in range(3)]; it = [dev.time_kernel(3, 30) for _ in range(3)]
print(json.dumps({"lib": sys.argv[1].split("/")[-1], "spmv_us": 1e3 * float(np.median(sp)), "iter_us": 1e3 * float(np.median(it))}))
''' % ROOT

if __name__ == "__main__":
    import numpy as np
    sys.path.insert(0, ROOT)
    from pylatticedso_amd import lattice_arrays as LA
    n = 50
    lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    np.savez("/tmp/ab_lat.npz", xyz=lat.node_xyz, conn=lat.beam_conn, rad=lat.beam_radius, seg_len=pen.seg_len,
             seg_nsub=pen.seg_nsub)
    for rnd in range(2):
        for lib in sys.argv[1:]:
            out = subprocess.run([sys.executable, "-c", CHILD, os.path.abspath(lib), str(n), "/tmp/ab_lat.npz"],
                                 capture_output=True, text=True)
            print(out.stdout.strip() or out.stderr[-400:], flush=True)
