"""Would a polynomial (Chebyshev) smoother in place of the Jacobi term pay on SEVERAL GPUs?  An ordinary PCG iteration has
one K*p and two global reductions; with  M^-1 = p_k(D^-1 A) D^-1 + Z A_c^-1 Z^T  an iteration has k K*p (neighbour exchange
only - overlappable) and still two reductions, so if the iteration count falls like ~1/k the reductions per solve fall
with it while the K*p count stays put.  Host experiment (scipy, CPU oracle's K), additive two-level PCG with rigid-body
modes on g^3-cell aggregates.   python tools/experiments/chebyshev_smoother.py GEOM n g [alpha]"""
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
sys.argv = [sys.argv[0]] + sys.argv[1:]
geom, n, g = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
alpha = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
E, NU = 1013.0, 0.3
radius = {"BCC": 0.05, "Octet": 0.03}[geom]
lat = LA.generate((1, 1, 1), (n, n, n), [geom], [radius])
pen = LA.penalize(lat, LA.compute_lzone(lat))
sc = c_oracle.condense_all(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
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


def rigid_modes(xyz, agg, n_agg):
    cnt = np.maximum(np.bincount(agg, minlength=n_agg), 1)
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / cnt[:, None]
    r = xyz - cen[agg]
    rows, cols, vals = [], [], []

    def put(node_dof, mode, val):
        rows.append(6 * np.arange(N) + node_dof); cols.append(6 * agg + mode); vals.append(val * np.ones(N))
    for k in range(3):
        put(k, k, 1.0)
        a, c = (k + 1) % 3, (k + 2) % 3
        put(c, 3 + k, r[:, a]); put(a, 3 + k, -r[:, c]); put(3 + k, 3 + k, 1.0)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, 6 * n_agg))


na = int(np.ceil(n / g - 1e-9))
cell = np.minimum((xyz / g).astype(int), na - 1)
agg = (cell[:, 0] * na + cell[:, 1]) * na + cell[:, 2]
Z = rigid_modes(xyz, agg, na ** 3)[v]
Z = Z[:, np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)].tocsr()
Ac = (Z.T @ A @ Z).toarray()
cf = sla.cho_factor(Ac + 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac)))
coarse = lambda r: Z @ sla.cho_solve(cf, Z.T @ r)
# lambda_max of D^-1 A by a few power iterations (what the device would do once per assembly)
x = np.random.default_rng(0).standard_normal(len(d))
for _ in range(30):
    x = (A @ x) / d
    lam = np.linalg.norm(x)
    x /= lam
lam_max = 1.05 * lam
nmat = [0]


def cheb(k, a, bnd):
    th, de = 0.5 * (bnd + a), 0.5 * (bnd - a)
    sig = th / de

    def apply(r):
        dz = (r / d) / th
        z = dz.copy()
        rho = 1.0 / sig
        for _ in range(k - 1):
            rho_n = 1.0 / (2.0 * sig - rho)
            res = r - A @ z
            nmat[0] += 1
            dz = rho_n * rho * dz + (2.0 * rho_n / de) * (res / d)
            z += dz
            rho = rho_n
        return z
    return apply


def pcg(Minv, rtol=1e-8, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b)
    for k in range(maxit):
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p; r -= a * Ap
        if np.linalg.norm(r) <= rtol * bn:
            return k + 1
        z = Minv(r); rz_new = r @ z; p = z + (rz_new / rz) * p; rz = rz_new
    return maxit


print(f"{geom} {n}^3, aggregates of {g:g}^3 cells ({Z.shape[1]} coarse dofs), lambda_max(D^-1 A) ~ {lam_max:.3f}, alpha {alpha:g}")
it0 = pcg(lambda r: r / d + coarse(r))
print(f"  Jacobi + coarse:            {it0:4d} iterations = {it0} K*p, {2 * it0} reductions")
for k in (2, 3, 4, 6):
    nmat[0] = 0
    sm = cheb(k, lam_max / alpha, lam_max)
    it = pcg(lambda r: sm(r) + coarse(r))
    print(f"  Chebyshev({k}) + coarse:      {it:4d} iterations = {it * k} K*p, {2 * it} reductions")

# multiplicative (symmetric two-grid cycle): z1 = S r; z2 = z1 + C (r - A z1); z = z2 + S^T (r - A z2)  (S symmetric here)
for k in (1, 2, 3, 4):
    sm = cheb(k, lam_max / alpha, lam_max)

    def vcycle(r, sm=sm):
        z = sm(r)
        z = z + coarse(r - A @ z)
        return z + sm(r - A @ z)
    it = pcg(vcycle)
    print(f"  V-cycle Cheb({k}) pre/post:    {it:4d} iterations = {it * (2 * k + 1)} K*p (+1 per iteration for p), {2 * it} reductions (+1 coarse)")
