"""Host prototype (round 5, verdict item 4): what would a coarse space that carries BENDING buy on bending-dominated lattices?
Additive multi-level PCG in the device's form - Jacobi + tile level (block diagonal, rigid + uniform strains per tile) + dense
level (rigid + uniform strains per aggregate) - on the cantilever of bench.py, BCC on the Schur complement of the cell centres
(opts.condense), with the dense level's space enriched:
  sa      smoothed aggregation of the dense level:  Z <- (I - omega D^-1 A) Z0,  omega = 4 / (3 rho(D^-1 A))
  bend6   + six pure-bending fields per aggregate: for every axis a and transverse b, u_a = x_a x_b, u_b = -x_a^2 / 2, theta = curl u / 2
  quad18  + all 18 quadratic displacement fields per aggregate (theta = curl u / 2)
Usage: python tools/experiments/bending_coarse_space.py GEOM n g_dense g_tile      (aggregate / tile edges in cells)"""
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import timoshenko_oracle as O, c_oracle          # noqa: E402
from pylatticedso_amd import lattice_arrays as LA             # noqa: E402

geom, n = sys.argv[1], int(sys.argv[2])
g_dense, g_tile = float(sys.argv[3]), float(sys.argv[4])
E, NU = 1013.0, 0.3
radius = {"BCC": 0.05, "Octet": 0.03, "Kelvin": 0.03, "Cubic": 0.05}[geom]


def fields(xyz, agg, n_agg, which):
    """Columns = modes of every aggregate, (6N, n_modes * n_agg); fields given as (u(r), theta(r)) of r = x - centre."""
    cnt = np.maximum(np.bincount(agg, minlength=n_agg), 1)
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / cnt[:, None]
    r = xyz - cen[agg]
    N = len(xyz)
    x, y, z = r.T
    one, zero = np.ones(N), np.zeros(N)
    modes = []
    ex = np.eye(3)
    for k in range(3):                                   # translations, rotations
        modes.append((ex[k][:, None] * one, np.zeros((3, N))))
    for k in range(3):
        w = ex[k]
        modes.append((np.cross(w, r).T, w[:, None] * one))
    if which >= 12:                                      # uniform strains (no rotation)
        for a, b in [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]:
            u = np.zeros((3, N))
            if a == b:
                u[a] = r[:, a]
            else:
                u[a], u[b] = 0.5 * r[:, b], 0.5 * r[:, a]
            modes.append((u, np.zeros((3, N))))
    quad = []
    if which == 18:                                      # six pure-bending fields
        for a in range(3):
            for b in range(3):
                if a == b:
                    continue
                u = np.zeros((3, N))
                u[a] = r[:, a] * r[:, b]
                u[b] = -0.5 * r[:, a] ** 2
                # theta = curl u / 2: only d u_a / d x_b = x_a and d u_b / d x_a = -x_a enter
                th = np.zeros((3, N))
                c = 3 - a - b                             # the third axis
                sgn = 1.0 if (b, a, c) in [(0, 1, 2), (1, 2, 0), (2, 0, 1)] else -1.0
                # (curl u)_c = eps_{c b a} d_b u_a + eps_{c a b} d_a u_b = eps_{cba} x_a - eps_{cab} x_a = 2 eps_{cba} x_a
                th[c] = sgn * r[:, a]
                quad.append((u, th))
    if which == 30:                                      # all quadratic displacement fields
        mons = [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]
        for comp in range(3):
            for (j, k) in mons:
                u = np.zeros((3, N))
                u[comp] = r[:, j] * r[:, k]
                grad = np.zeros((3, N))                  # d u_comp / d x_m
                grad[j] += r[:, k]
                grad[k] += r[:, j]
                th = np.zeros((3, N))
                # theta_i = 1/2 eps_{i m comp} d_m u_comp
                for i in range(3):
                    for m in range(3):
                        e = np.linalg.det(np.array([ex[i], ex[m], ex[comp]]))
                        if e != 0:
                            th[i] += 0.5 * e * grad[m]
                quad.append((u, th))
    modes += quad
    nm = len(modes)
    rows, cols, vals = [], [], []
    for q, (u, th) in enumerate(modes):
        for k in range(3):
            rows += [6 * np.arange(N) + k, 6 * np.arange(N) + 3 + k]
            cols += [nm * agg + q, nm * agg + q]
            vals += [u[k], th[k]]
    Z = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, nm * n_agg))
    return Z, nm


def pcg(A, b, Minv, rtol=1e-8, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = Minv(r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b)
    for k in range(maxit):
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p; r -= a * Ap
        if np.linalg.norm(r) <= rtol * bn:
            return x, k + 1
        z = Minv(r); rz_new = r @ z
        p = z + (rz_new / rz) * p; rz = rz_new
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
if geom == "BCC":
    centre = np.abs(xyz - np.floor(xyz) - 0.5).max(axis=1) < 1e-9
    cdof = np.repeat(centre, 6)
    v = np.flatnonzero(~fixed & ~cdof); c = np.flatnonzero(cdof)
    Kvv, Kvc, Kcc = K[v][:, v].tocsr(), K[v][:, c].tocsr(), K[c][:, c]
    Kb = Kcc.tobsr(blocksize=(6, 6))
    inv = sp.bsr_matrix((np.linalg.inv(Kb.data), Kb.indices, Kb.indptr), shape=Kb.shape).tocsr()
    A = (Kvv - Kvc @ inv @ Kvc.T).tocsr()
    d = Kvv.diagonal()
else:
    v = np.flatnonzero(~fixed)
    A = K[v][:, v].tocsr(); d = A.diagonal()
b = f[v]
reg = lambda M: M + 1e-12 * np.trace(M) / len(M) * np.eye(len(M))


def aggregates(gc):
    na = int(np.ceil(n / gc - 1e-9))
    cell = np.minimum((xyz / gc).astype(int), na - 1)
    return (cell[:, 0] * na + cell[:, 1]) * na + cell[:, 2], na ** 3


def level(gc, dense, which, smooth=False):
    agg, na = aggregates(gc)
    Z, nm = fields(xyz, agg, na, which)
    Z = Z[v]
    if smooth:
        rho = spl.eigsh(sp.diags(1 / np.sqrt(d)) @ A @ sp.diags(1 / np.sqrt(d)), k=1, which="LA", return_eigenvectors=False)[0]
        Z = (Z - (4.0 / (3.0 * rho)) * sp.diags(1 / d) @ (A @ Z)).tocsr()
    keep = np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)
    Z = Z[:, keep].tocsr()
    B = (Z.T @ A @ Z).toarray()
    if not dense:
        aid = keep // nm
        B = B * (aid[:, None] == aid[None, :])
    w, V = np.linalg.eigh(0.5 * (B + B.T))               # (enriched spaces can be nearly dependent: pseudo-inverse)
    good = w > 1e-10 * w.max()
    Binv = (V[:, good] / w[good]) @ V[:, good].T
    return (lambda r: Z @ (Binv @ (Z.T @ r))), Z.shape[1], int((~good).sum())


print(f"{geom} {n}^3: {lat.n_beams} struts, {len(v)} unknowns; dense aggregates {g_dense:g}^3 cells, tiles {g_tile:g}^3 cells", flush=True)
tile12, nt, _ = level(g_tile, False, 12)
x_ref = None
for name, which, smooth in [("dense 12 modes (device form)", 12, False), ("dense 12 modes, smoothed aggregation", 12, True),
                            ("dense 12 + 6 bending modes", 18, False), ("dense 12 + 18 quadratic modes", 30, False)]:
    fn, nd, dropped = level(g_dense, True, which, smooth)
    x, it = pcg(A, b, lambda r: r / d + tile12(r) + fn(r))
    if x_ref is None:
        x_ref = x
    assert np.linalg.norm(x - x_ref) < 1e-5 * np.linalg.norm(x_ref)
    print(f"  Jacobi + tile level (12 modes, {nt} dofs) + {name:38s} ({nd:5d} dofs, {dropped} dropped): {it:4d} iterations", flush=True)
for name, which in [("tile level 12 + 6 bending modes", 18)]:
    tfn, nt2, dropped = level(g_tile, False, which)
    fn, nd, _ = level(g_dense, True, 12)
    x, it = pcg(A, b, lambda r: r / d + tfn(r) + fn(r))
    print(f"  Jacobi + {name} ({nt2} dofs, {dropped} dropped) + dense 12 modes: {it:4d} iterations", flush=True)
    fn, nd, _ = level(g_dense, True, 18)
    x, it = pcg(A, b, lambda r: r / d + tfn(r) + fn(r))
    print(f"  Jacobi + {name} + dense 12 + 6 bending: {it:4d} iterations", flush=True)
