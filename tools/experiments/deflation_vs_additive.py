"""Host prototype (round 5): the dense level applied MULTIPLICATIVELY (deflation / balancing) instead of additively.
Device form today:  M^-1 = D^-1 + Z_t B_t^-1 Z_t^T + Z B^-1 Z^T  (all additive).  Variants with the same spaces:
  adef2   M^-1 = Q + P^T M1^-1 ... (A-DEF2: z = Q r + M1^-1 (r - A Q r) with Q = Z B^-1 Z^T: one extra A-product on a coarse vector)
  bnn     balancing: z = Q r + P M1^-1 P^T r  (two extra coarse corrections)
where M1^-1 = D^-1 + tile level.  Counts iterations on the bench cantilever (BCC on the Schur complement of the centres).
Usage: python tools/experiments/deflation_vs_additive.py GEOM n g_dense g_tile"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.argv = [sys.argv[0]] + (sys.argv[1:] if len(sys.argv) > 1 else ["BCC", "14", "3.5", "1.75"])
HERE = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(HERE, "bending_coarse_space.py")).read()
src = src[:src.index('print(f"{geom} {n}^3:')]          # reuse its set-up (lattice, A, b, d, fields, level, pcg)
exec(src)

print(f"{geom} {n}^3: {len(v)} unknowns; dense aggregates {g_dense:g}^3 cells, tiles {g_tile:g}^3 cells", flush=True)
tile12, nt, _ = level(g_tile, False, 12)
agg, na = aggregates(g_dense)
Z, nm = fields(xyz, agg, na, 12)
Z = Z[v]
keep = np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)
Z = Z[:, keep].tocsr()
AZ = (A @ Z).tocsr()
B = (Z.T @ AZ).toarray()
w_, V_ = np.linalg.eigh(0.5 * (B + B.T))
good = w_ > 1e-10 * w_.max()
Binv = (V_[:, good] / w_[good]) @ V_[:, good].T
Q = lambda r: Z @ (Binv @ (Z.T @ r))
M1 = lambda r: r / d + tile12(r)
print("nnz(AZ) rows touched:", int((np.asarray(abs(AZ).sum(axis=1)).ravel() > 1e-9 * abs(AZ).max()).sum()), "of", AZ.shape[0])
x0, it = pcg(A, b, lambda r: M1(r) + Q(r))
print(f"  additive (device form)            : {it:4d} iterations")


def adef2(r):
    q = Q(r)
    return q + M1(r - A @ q)


x1, it = pcg(A, b, adef2)
print(f"  A-DEF2  z = Qr + M1(r - A Q r)    : {it:4d} iterations   rel diff {np.linalg.norm(x1 - x0) / np.linalg.norm(x0):.1e}")


def bnn(r):
    q = Q(r)
    t = M1(r - A @ q)
    return q + t - Q(A @ t)


x2, it = pcg(A, b, bnn)
print(f"  BNN     z = Qr + P M1 P^T r       : {it:4d} iterations   rel diff {np.linalg.norm(x2 - x0) / np.linalg.norm(x0):.1e}")
# the same with the tile level applied multiplicatively on top (symmetric: pre- and post-)
