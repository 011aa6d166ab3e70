"""Does a coarse space of rigid motions PLUS uniform strains per aggregate see the soft modes of bending-dominated
lattices?  Host experiment (scipy) on the device-assembled K: additive two-level PCG  M^-1 = D^-1 + Z (Z^T K Z)^-1 Z^T
with Z = 6 rigid-body modes per brick aggregate (what libpylattice_hip ships) against Z = 12 modes
(u = a + G (x - c), theta = axial(skew G)), for BCC and Octet cantilevers.  DESIGN.md section 11, item 3.
Usage (GPU box): python tools/experiments/affine_coarse_space.py [cells per edge = 16] [aggregate edge in cells = 4]"""
import os
import sys

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pylatticedso_amd import _capi, lattice_arrays as LA   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = int(sys.argv[2]) if len(sys.argv) > 2 else 4
E, NU = 1013.0, 0.3


def modes(xyz, agg, n_agg, affine):
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / np.bincount(agg, minlength=n_agg)[:, None]
    r = xyz - cen[agg]
    N = len(xyz)
    m = 12 if affine else 6
    rows, cols, vals = [], [], []

    def put(node_dof, mode, v):
        rows.append(6 * np.arange(N) + node_dof)
        cols.append(m * agg + mode)
        vals.append(v * np.ones(N))
    for k in range(3):
        put(k, k, 1.0)                                   # translations
    if not affine:
        # rotation omega_k: u = e_k x r, theta = e_k
        for k in range(3):
            a, b = (k + 1) % 3, (k + 2) % 3
            put(b, 3 + k, r[:, a])
            put(a, 3 + k, -r[:, b])
            put(3 + k, 3 + k, 1.0)
    else:
        # G_ab: u_a = r_b; theta = 1/2 axial(G - G^T): theta_c = 1/2 eps_cab G_ba ... (G_ab contributes -1/2 eps_cab)
        eps = np.zeros((3, 3, 3))
        eps[0, 1, 2] = eps[1, 2, 0] = eps[2, 0, 1] = 1.0
        eps[0, 2, 1] = eps[2, 1, 0] = eps[1, 0, 2] = -1.0
        for a in range(3):
            for b in range(3):
                mode = 3 + 3 * a + b
                put(a, mode, r[:, b])
                for c in range(3):
                    # theta = 1/2 curl u: theta_c = 1/2 eps_c b a d u_a / d x_b = 1/2 eps_cba G_ab
                    if eps[c, b, a] != 0.0:
                        put(3 + c, mode, 0.5 * eps[c, b, a])
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, m * n_agg))


def pcg(K, b, Minv, rtol=1e-8, maxit=20000):
    x = np.zeros_like(b)
    r = b.copy()
    z = Minv(r)
    p = z.copy()
    rz = r @ z
    bn = np.linalg.norm(b)
    for k in range(maxit):
        Ap = K @ p
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


geoms = {"BCC": 0.05, "Octet": 0.03}
for geom in (sys.argv[3].split(",") if len(sys.argv) > 3 else list(geoms)):
    radius = geoms[geom]
    lat = LA.generate((1, 1, 1), (n, n, n), [geom], [radius])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU, reorder=0) as dev:
        dev.set_bc(np.zeros((lat.n_nodes, 6), bool))
        dev.assemble()
        dev.assemble_bsr(False)
        rowptr, col, vals = dev.get_bsr()
    N = lat.n_nodes
    K = sp.bsr_matrix((vals, col, rowptr), shape=(6 * N, 6 * N)).tocsr()
    xyz = lat.node_xyz
    fixed = np.repeat(xyz[:, 0] < 1e-9, 6)
    free = np.flatnonzero(~fixed)
    f = np.zeros(6 * N)
    tip = np.flatnonzero(xyz[:, 0] > n - 1e-9)
    f[6 * tip + 2] = -0.1 / len(tip)
    Kff = K[free][:, free].tocsr()
    d = Kff.diagonal()
    na = int(np.ceil(n / g))
    cell = np.minimum((xyz / g).astype(int), na - 1)
    agg = (cell[:, 0] * na + cell[:, 1]) * na + cell[:, 2]
    out = [f"{geom} {n}^3: {lat.n_beams} struts, {len(free)} free dofs, aggregates {na}^3 of {g}^3 cells |"]
    x_ref, it = pcg(Kff, f[free], lambda r: r / d)
    out.append(f"Jacobi {it}")
    for affine in (False, True):
        Z = modes(xyz, agg, na ** 3, affine)[free]
        keep = np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)
        Z = Z[:, keep].tocsr()
        Ac = (Z.T @ Kff @ Z).toarray()
        Ac += 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac))
        cf = sla.cho_factor(Ac)
        solve = lambda y, cf=cf: sla.cho_solve(cf, y)
        x, it = pcg(Kff, f[free], lambda r: r / d + Z @ solve(Z.T @ r))
        assert np.linalg.norm(x - x_ref) < 1e-5 * np.linalg.norm(x_ref)
        out.append(f"{'rigid+strain (12)' if affine else 'rigid (6)'} x {Z.shape[1]} coarse dofs: {it}")
    # fine-level alternatives with the rigid coarse space: 6x6 node-block Jacobi, and (BCC is bipartite: corner nodes
    # only touch centre nodes) the same on the Schur complement of the centre nodes
    Z = modes(xyz, agg, na ** 3, False)[free]
    Z = Z[:, np.flatnonzero(np.asarray(abs(Z).sum(axis=0)).ravel() > 0)].tocsr()
    Ac = (Z.T @ Kff @ Z).toarray()
    cf = sla.cho_factor(Ac + 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac)))
    solve = lambda y: sla.cho_solve(cf, y)
    Kb = K.tobsr(blocksize=(6, 6))
    Dblk = np.zeros((N, 6, 6))
    for i in range(N):
        js = Kb.indices[Kb.indptr[i]:Kb.indptr[i + 1]]
        Dblk[i] = Kb.data[Kb.indptr[i]:Kb.indptr[i + 1]][js == i].sum(axis=0)
    fm = (~fixed).reshape(N, 6)
    for i in range(N):                                   # constrained dofs: identity rows
        off = ~fm[i]
        Dblk[i][off, :] = 0
        Dblk[i][:, off] = 0
        Dblk[i][off, off] = 1
    Dinv = np.linalg.inv(Dblk)

    def block_jacobi(r):
        full = np.zeros(6 * N)
        full[free] = r
        return np.einsum("nij,nj->ni", Dinv, full.reshape(N, 6)).ravel()[free]
    x, it = pcg(Kff, f[free], lambda r: block_jacobi(r) + Z @ solve(Z.T @ r))
    out.append(f"| node-block Jacobi + rigid: {it}")
    print("  ".join(out), flush=True)
