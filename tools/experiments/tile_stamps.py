"""Where a workgroup of the tile K*p spends its life (experiment build with -DPL_TILE_STAMPS, PYLATTICE_HIP_LIB pointing at
it): median over the first 4096 workgroups of one launch of the time between the stamps - 0 kernel entry, 1 after the LDS
clear + barrier, 2 first visit's indices requested, 3 end of the strut loop, 4 after the barrier, 5 end of the epilogue.
    PYLATTICE_HIP_LIB=build_exp/libpl_stamps.so python tools/experiments/tile_stamps.py [cells] [palette]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pylatticedso_amd import _capi, lattice_arrays as LA   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
pal = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lat = LA.generate((1, 1, 1), (n, n, n), ["Octet"], [0.03])
pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
fixed = np.zeros((lat.n_nodes, 6), np.uint8)
fixed[lat.node_xyz[:, 0] == 0.0] = 1
f = np.zeros((lat.n_nodes, 6))
f[lat.node_xyz[:, 0] == float(n), 2] = -0.1
d = _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, 1013.0, 0.3, precond=3,
                     palette=pal)
d.set_bc(fixed, None, f)
d.assemble()
ms = d.time_kernel(0, 20)
lib = _capi.load_library()
out = (C.c_ulonglong * (8 * 4096))()
assert lib.pl_debug_tile_stamps(out) == 0
st = np.array(out, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
st = st[st[:, 5] > 0]
dt = np.diff(st[:, :6], axis=1) * 0.01          # 100 MHz -> us
names = ["entry -> LDS clear + barrier", "first-visit index loads issued", "strut loop", "barrier after the loop", "epilogue"]
print(f"{n}^3 Octet palette={pal}: K*p {ms * 1e3:.1f} us, {len(st)} workgroups stamped, lifetime median "
      f"{np.median(st[:, 5] - st[:, 0]) * 0.01:.2f} us")
for k, nm in enumerate(names):
    print(f"  {nm:34s} median {np.median(dt[:, k]):6.2f} us   mean {dt[:, k].mean():6.2f}")
span = (st[:, 5].max() - st[:, 0].min()) * 0.01
print(f"  first entry -> last exit of the stamped workgroups: {span:.1f} us")
