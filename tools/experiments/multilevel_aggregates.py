"""How much does a FINER block level under the tile level buy?  Host experiment (scipy, CPU oracle's K): additive
multi-level PCG  M^-1 = D^-1 + sum over levels of Z_l B_l^-1 Z_l^T  with rigid-body modes of brick aggregates; the
coarsest level is dense (all aggregates coupled, as the device's dense level), the others block diagonal (one 6 x 6 block
per aggregate, as the device's tile level).  BCC runs on the Schur complement of the cell centres (opts.condense).
Usage: python tools/experiments/multilevel_aggregates.py GEOM n  g_dense  g_block [g_block ...]   (edges in cells)"""
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import timoshenko_oracle as O, c_oracle          # noqa: E402
from pylatticedso_amd import lattice_arrays as LA             # noqa: E402

geom, n = sys.argv[1], int(sys.argv[2])
g_dense = float(sys.argv[3])
g_blocks = [float(a) for a in sys.argv[4:]]
E, NU = 1013.0, 0.3
radius = {"BCC": 0.05, "Octet": 0.03}[geom]


def rigid_modes(xyz, agg, n_agg):
    cnt = np.maximum(np.bincount(agg, minlength=n_agg), 1)
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / cnt[:, None]
    r = xyz - cen[agg]
    N = len(xyz)
    rows, cols, vals = [], [], []

    def put(node_dof, mode, v):
        rows.append(6 * np.arange(N) + node_dof)
        cols.append(6 * agg + mode)
        vals.append(v * np.ones(N))
    for k in range(3):
        put(k, k, 1.0)
        a, b = (k + 1) % 3, (k + 2) % 3
        put(b, 3 + k, r[:, a])
        put(a, 3 + k, -r[:, b])
        put(3 + k, 3 + k, 1.0)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, 6 * n_agg))


def pcg(A, b, Minv, rtol=1e-8, maxit=20000):
    x = np.zeros_like(b)
    r = b.copy()
    z = Minv(r)
    p = z.copy()
    rz = r @ z
    bn = np.linalg.norm(b)
    for k in range(maxit):
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p
        r -= a * Ap
        if np.linalg.norm(r) <= rtol * bn:
            return x, k + 1
        z = Minv(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, maxit


lat = LA.generate((1, 1, 1), (n, n, n), [geom], [radius])
pen = LA.penalize(lat, LA.compute_lzone(lat))
sc = c_oracle.condense_unique(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, sc).tocsr()
N = lat.n_nodes
xyz = lat.node_xyz
fixed = np.repeat(xyz[:, 0] < 1e-9, 6)
f = np.zeros(6 * N)
tip = np.flatnonzero(xyz[:, 0] > n - 1e-9)
f[6 * tip + 2] = -0.1 / len(tip)
if geom == "BCC":
    centre = np.abs(xyz - np.floor(xyz) - 0.5).max(axis=1) < 1e-9
    cdof = np.repeat(centre, 6)
    v = np.flatnonzero(~fixed & ~cdof)
    c = np.flatnonzero(cdof)
    Kvv, Kvc, Kcc = K[v][:, v].tocsr(), K[v][:, c].tocsr(), K[c][:, c]
    Kb = Kcc.tobsr(blocksize=(6, 6))
    inv = sp.bsr_matrix((np.linalg.inv(Kb.data), Kb.indices, Kb.indptr), shape=Kb.shape).tocsr()
    A = (Kvv - Kvc @ inv @ Kvc.T).tocsr()
    d = Kvv.diagonal()
else:
    v = np.flatnonzero(~fixed)
    A = K[v][:, v].tocsr()
    d = A.diagonal()
b = f[v]
reg = lambda M: M + 1e-12 * np.trace(M) / len(M) * np.eye(len(M))


def strain_modes(xyz, agg, n_agg):
    """six uniform strains per aggregate: u = eps (x - c), no rotation"""
    cnt = np.maximum(np.bincount(agg, minlength=n_agg), 1)
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / cnt[:, None]
    r = xyz - cen[agg]
    N = len(xyz)
    rows, cols, vals = [], [], []
    for q, (a, b2) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]):
        rows.append(6 * np.arange(N) + a); cols.append(6 * agg + q); vals.append(r[:, b2] if a == b2 else 0.5 * r[:, b2])
        if a != b2:
            rows.append(6 * np.arange(N) + b2); cols.append(6 * agg + q); vals.append(0.5 * r[:, a])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, 6 * n_agg))


def level(gc, dense, affine=False):
    na = int(np.ceil(n / gc - 1e-9))
    cell = np.minimum((xyz / gc).astype(int), na - 1)
    agg = (cell[:, 0] * na + cell[:, 1]) * na + cell[:, 2]
    Z = rigid_modes(xyz, agg, na ** 3)
    if affine:      # columns interleaved per aggregate: [6 rigid | 6 strain] -> block id = column // 6 % ... keep it simple:
        Zs = strain_modes(xyz, agg, na ** 3)
        Z = sp.hstack([Z, Zs]).tocsr()
    Z = Z[v]
    keep = np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)
    Z = Z[:, keep].tocsr()
    B = (Z.T @ A @ Z).toarray()
    if not dense:
        aid = (keep % (6 * na ** 3)) // 6
        B = B * (aid[:, None] == aid[None, :])
    cfac = sla.cho_factor(reg(B))
    return lambda r: Z @ sla.cho_solve(cfac, Z.T @ r), Z.shape[1]


print(f"{geom} {n}^3: {lat.n_beams} struts, {len(v)} unknowns", flush=True)
dense_apply, nd = level(g_dense, True, bool(int(os.environ.get("DENSE_AFFINE", "0"))))
x_ref, it = pcg(A, b, lambda r: r / d + dense_apply(r))
print(f"  Jacobi + dense level ({g_dense:g}^3 cells, {nd} dofs): {it}", flush=True)
affine = bool(int(os.environ.get("AFFINE", "0")))
applied = []
for gb in g_blocks:
    fn, nb = level(gb, False, affine)
    applied.append(fn)
    x, it = pcg(A, b, lambda r: r / d + dense_apply(r) + sum(fn(r) for fn in applied))
    assert np.linalg.norm(x - x_ref) < 1e-5 * np.linalg.norm(x_ref)
    print(f"  + block level {gb:g}^3 cells ({nb} dofs): {it}", flush=True)
