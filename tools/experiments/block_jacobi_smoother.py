"""Would 6 x 6 NODE BLOCKS instead of the diagonal as the fine level of the FEM path's preconditioner buy iterations?  Host
experiment (scipy, the CPU oracle's K): Jacobi / node-block Jacobi + block-diagonal 12-mode tile level + dense 12-mode level.
Usage: python tools/experiments/block_jacobi_smoother.py GEOM n g_dense g_tile"""
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import timoshenko_oracle as O, c_oracle          # noqa: E402
from pylatticedso_amd import lattice_arrays as LA             # noqa: E402

geom, n = sys.argv[1], int(sys.argv[2])
g_dense, g_tile = float(sys.argv[3]), float(sys.argv[4])
E, NU = 1013.0, 0.3
radius = {"BCC": 0.05, "Octet": 0.03}[geom]


def modes12(xyz, agg, n_agg):
    cnt = np.maximum(np.bincount(agg, minlength=n_agg), 1)
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / cnt[:, None]
    r = xyz - cen[agg]
    N = len(xyz)
    rows, cols, vals = [], [], []

    def put(node_dof, mode, v):
        rows.append(6 * np.arange(N) + node_dof); cols.append(12 * agg + mode); vals.append(v * np.ones(N))
    for k in range(3):
        put(k, k, 1.0)
        a, b = (k + 1) % 3, (k + 2) % 3
        put(b, 3 + k, r[:, a]); put(a, 3 + k, -r[:, b]); put(3 + k, 3 + k, 1.0)
    for q, (a, b2) in enumerate([(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]):
        put(a, 6 + q, r[:, b2] if a == b2 else 0.5 * r[:, b2])
        if a != b2:
            put(b2, 6 + q, 0.5 * r[:, a])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, 12 * n_agg))


def pcg(A, b, Minv, rtol=1e-8, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b)
    for k in range(maxit):
        Ap = A @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        if np.linalg.norm(r) <= rtol * bn:
            return x, k + 1
        z = Minv(r); rz_new = r @ z; p = z + (rz_new / rz) * p; rz = rz_new
    return x, maxit


lat = LA.generate((1, 1, 1), (n, n, n), [geom], [radius])
pen = LA.penalize(lat, LA.compute_lzone(lat))
sc = c_oracle.condense_unique(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, sc).tocsr()
N, xyz = lat.n_nodes, lat.node_xyz
fixed = np.repeat(xyz[:, 0] < 1e-9, 6)
f = np.zeros(6 * N)
tip = np.flatnonzero(xyz[:, 0] > n - 1e-9)
f[6 * tip + 2] = -0.1 / len(tip)
v = np.flatnonzero(~fixed)
A = K[v][:, v].tocsr()
d = A.diagonal()
b = f[v]
reg = lambda M: M + 1e-12 * np.trace(M) / len(M) * np.eye(len(M))

# node blocks of K (constrained nodes are whole nodes here: the clamped face)
Kb = K.tobsr(blocksize=(6, 6))
Kb.sort_indices()
blocks = np.zeros((N, 6, 6))
for i in range(N):
    cols = Kb.indices[Kb.indptr[i]:Kb.indptr[i + 1]]
    blocks[i] = Kb.data[Kb.indptr[i] + int(np.searchsorted(cols, i))]
binv = np.linalg.inv(blocks)
free_nodes = ~fixed.reshape(N, 6)[:, 0]


def block_jacobi(r):
    full = np.zeros(6 * N); full[v] = r
    z = np.einsum("nij,nj->ni", binv, full.reshape(N, 6))
    z[~free_nodes] = 0.0
    return z.ravel()[v]


def grid(gc):
    na = int(np.ceil(n / gc - 1e-9))
    cell = np.minimum((xyz / gc).astype(int), na - 1)
    return (cell[:, 0] * na + cell[:, 1]) * na + cell[:, 2], na ** 3


def level(gc):
    agg, na = grid(gc)
    Z = modes12(xyz, agg, na)[v]
    keep = np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)
    Z = Z[:, keep].tocsr()
    return Z, keep // 12


print(f"{geom} {n}^3: {lat.n_beams} struts, {len(v)} unknowns; dense level {g_dense:g}^3 cells, tile level {g_tile:g}^3 cells", flush=True)
Zd, _ = level(g_dense)
cd = sla.cho_factor(reg((Zd.T @ A @ Zd).toarray()))
dense = lambda r: Zd @ sla.cho_solve(cd, Zd.T @ r)
Zt, aid = level(g_tile)
Bt = (Zt.T @ A @ Zt).toarray() * (aid[:, None] == aid[None, :])
cb = sla.cho_factor(reg(Bt))
tile_block = lambda r: Zt @ sla.cho_solve(cb, Zt.T @ r)
for name, fine in (("diagonal", lambda r: r / d), ("6 x 6 node blocks", block_jacobi)):
    _, i0 = pcg(A, b, fine)
    _, i1 = pcg(A, b, lambda r: fine(r) + dense(r))
    _, i2 = pcg(A, b, lambda r: fine(r) + dense(r) + tile_block(r))
    print(f"  fine level = {name:18s}: alone {i0:5d}   + dense {i1:4d}   + dense + tile level [device form] {i2:4d}", flush=True)
