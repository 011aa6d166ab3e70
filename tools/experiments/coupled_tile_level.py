"""What would a COUPLED tile level buy?  Host experiment (scipy, the CPU oracle's K).  Today's device preconditioner is
additive: Jacobi + block-diagonal tile level (12 modes per ~3.3^3-cell brick, no coupling between bricks) + dense level
(12 modes per aggregate of bricks, all coupled).  Here the tile level's Galerkin operator A_t = Z_t^T A Z_t is kept WITH
its couplings and solved (a) exactly - the upper bound - or (b) by a fixed polynomial: k Chebyshev steps on A_t
preconditioned by its own block diagonal + the dense level (what a device version would run: one block-sparse product
with A_t per step).
Usage: python tools/experiments/coupled_tile_level.py GEOM n g_dense g_tile"""
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
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
Ad = (Zd.T @ A @ Zd).toarray()
cd = sla.cho_factor(reg(Ad))
dense = lambda r: Zd @ sla.cho_solve(cd, Zd.T @ r)
Zt, aid = level(g_tile)
At = (Zt.T @ A @ Zt).tocsr()
Bt = At.toarray() * (aid[:, None] == aid[None, :])
cb = sla.cho_factor(reg(Bt))
tile_block = lambda r: Zt @ sla.cho_solve(cb, Zt.T @ r)
_, it0 = pcg(A, b, lambda r: r / d + dense(r))
print(f"  Jacobi + dense ({Zd.shape[1]} dofs): {it0}")
x_ref, it1 = pcg(A, b, lambda r: r / d + dense(r) + tile_block(r))
print(f"  Jacobi + dense + block-diagonal tile level ({Zt.shape[1]} dofs)  [today]: {it1}")
lu = spla.splu(sp.csc_matrix(At + 1e-12 * sp.identity(At.shape[0])))
_, it2 = pcg(A, b, lambda r: r / d + Zt @ lu.solve(Zt.T @ r))
print(f"  Jacobi + COUPLED tile level, exact solve [upper bound]: {it2}   (nnz(A_t) = {At.nnz}, {At.nnz / At.shape[0]:.0f} per row)")

# tile-level operator solved by a fixed polynomial: Chebyshev on M_t^-1 A_t, M_t^-1 = block diagonal^-1 + P_d A_d^-1 P_d^T
# (dense level expressed in tile coordinates: Z_d = Z_t P exactly when the bricks nest; here by least squares)
P = spla.lsqr  # noqa (not used: the dense level is applied through the fine space instead)
Mt = lambda rt: sla.cho_solve(cb, rt) + spla.lsqr  # placeholder, replaced below


def make_Mt():
    # dense level restricted to the tile space: Z_d^T Z_t^+ ... simplest exact route: y = B^-1 r_t + R A_d^-1 R^T r_t with
    # R = (Z_t^T Z_t)^-1 Z_t^T Z_d  (coefficients of the aggregate modes in the brick modes)
    G = (Zt.T @ Zt).tocsc()
    R = spla.spsolve(G + 1e-14 * sp.identity(G.shape[0], format="csc"), (Zt.T @ Zd).tocsc())
    R = sp.csr_matrix(R)
    Adt = (R.T @ At @ R).toarray()
    cdt = sla.cho_factor(reg(Adt))
    return lambda rt: sla.cho_solve(cb, rt) + R @ sla.cho_solve(cdt, R.T @ rt)


Mt = make_Mt()
# spectrum of M_t^-1 A_t
nt = At.shape[0]
op = spla.LinearOperator((nt, nt), matvec=lambda x: Mt(At @ x))
lmax = float(np.real(spla.eigs(op, k=1, which="LM", return_eigenvectors=False, tol=1e-3)[0])) * 1.05
for frac in (8.0, 16.0):
    lmin = lmax / frac
    for ksteps in (1, 2, 3, 4):
        def cheb(rt, ksteps=ksteps, lmin=lmin):
            # k steps of Chebyshev iteration for A_t y = rt, y0 = 0 (a fixed symmetric polynomial in M_t^-1 A_t)
            theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
            sigma = theta / delta
            rho = 1.0 / sigma
            y = np.zeros_like(rt)
            res = rt.copy()
            dvec = Mt(res) / theta
            for j in range(ksteps):
                y = y + dvec
                if j + 1 == ksteps:
                    break
                res = res - At @ dvec
                rho_new = 1.0 / (2.0 * sigma - rho)
                dvec = rho_new * rho * dvec + (2.0 * rho_new / delta) * Mt(res)
                rho = rho_new
            return y
        x, it = pcg(A, b, lambda r: r / d + Zt @ cheb(Zt.T @ r))
        err = np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref)
        print(f"  Jacobi + coupled tile level, {ksteps} Chebyshev step(s) [lmax/{frac:g}] ({ksteps - 1} products with A_t): {it}   (err {err:.1e})", flush=True)
