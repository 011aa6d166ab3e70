"""configs[3] design loop, phase by phase (wall per call; device ms from the handle's stats)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pylatticedso_amd import _capi, lattice_arrays as LA
n, iters = 24, 50
i3 = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 3) + 0.5
x, y, z = (2 * np.pi * i3[:, k] / 8 for k in range(3))
rc = np.clip(0.05 + 0.03 * (np.sin(x) * np.cos(y) + np.sin(y) * np.cos(z) + np.sin(z) * np.cos(x)) / 1.5, 0.01, 0.1)
lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05], cell_radii_override=rc.reshape(-1, 1))
pen = LA.penalize(lat, _capi.lzone(lat.node_xyz, lat.beam_conn, lat.beam_radius))
fixed, f, _ = bench.cantilever_bc(lat.node_xyz, float(n))
cell_of = lat.beam_cell0
T = {k: 0.0 for k in ("update_radii", "assemble", "solve", "objective", "sens", "step")}
dev_ms = {"assembly": 0.0, "solve": 0.0}
its = []
with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, bench.E, bench.NU,
                      precond=3, palette=1, warm_start=int(sys.argv[1]) if len(sys.argv) > 1 else 4) as dev:
    dev.set_bc(fixed, None, f)
    r = rc.copy()
    for k in range(iters + 2):
        t = [time.perf_counter()]
        dev.update_radii(r[cell_of]); t.append(time.perf_counter())
        dev.assemble(); t.append(time.perf_counter())
        u, st = dev.solve(rtol=1e-8, max_iter=100000); t.append(time.perf_counter())
        C = float((f * u).sum()); t.append(time.perf_counter())
        g = -np.bincount(cell_of, weights=dev.sens(None), minlength=len(r)); t.append(time.perf_counter())
        r = np.clip(r - 0.002 * g / max(np.abs(g).max(), 1e-300), 0.01, 0.1); t.append(time.perf_counter())
        if k >= 2:
            for name, a, b in zip(T, t[:-1], t[1:]):
                T[name] += b - a
            dev_ms["assembly"] += st["ms_assembly"]; dev_ms["solve"] += st["ms_solve"]; its.append(int(st["iterations"]))
print({k: round(1e3 * v / iters, 3) for k, v in T.items()}, "sum", round(1e3 * sum(T.values()) / iters, 3), "ms per design iteration")
print("device ms per design iteration:", {k: round(v / iters, 3) for k, v in dev_ms.items()}, "mean iterations", np.mean(its),
      "-> per reported iteration", round(1e3 * dev_ms["solve"] / sum(its), 2), "us")
