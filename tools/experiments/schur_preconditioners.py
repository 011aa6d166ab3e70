"""BCC with the cell centres eliminated (opts.condense): which fine-level preconditioner suits the Schur complement S of
the corner nodes?  Host experiment (scipy, CPU oracle's assembled K): additive two-level PCG on S with the rigid-body
coarse space and (a) diag(K_vv) - what libpylattice_hip uses today, the principal part of the ordinary preconditioner -
(b) diag(S), (c) the 6 x 6 node blocks of S, each with the coarse operator Z^T K Z (today) or Z_v^T S Z_v.
Usage: python tools/experiments/schur_preconditioners.py [cells per edge = 16] [aggregate edge in cells = 4]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = int(sys.argv[2]) if len(sys.argv) > 2 else 4
E, NU = 1013.0, 0.3


def rigid_modes(xyz, agg, n_agg):
    cen = np.stack([np.bincount(agg, xyz[:, k], n_agg) for k in range(3)], 1) / np.maximum(np.bincount(agg, minlength=n_agg), 1)[:, None]
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


lat = LA.generate((1, 1, 1), (n, n, n), ["BCC"], [0.05])
pen = LA.penalize(lat, LA.compute_lzone(lat))
sc = c_oracle.condense_unique(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
K = O.assemble_condensed(lat.node_xyz, lat.beam_conn, sc).tocsr()
N = lat.n_nodes
xyz = lat.node_xyz
fixed = np.repeat(xyz[:, 0] < 1e-9, 6)
f = np.zeros(6 * N)
tip = np.flatnonzero(xyz[:, 0] > n - 1e-9)
f[6 * tip + 2] = -0.1 / len(tip)
centre = np.abs(xyz - np.floor(xyz) - 0.5).max(axis=1) < 1e-9
cdof = np.repeat(centre, 6)
v = np.flatnonzero(~fixed & ~cdof)
c = np.flatnonzero(cdof)
Kvv, Kvc, Kcc = K[v][:, v].tocsr(), K[v][:, c].tocsr(), K[c][:, c].tocsc()
Kcc_inv = spla.splu(Kcc)
S = (Kvv - Kvc @ sp.csr_matrix(Kcc_inv.solve(Kvc.T.toarray()))).tocsr() if n <= 8 else None
if S is None:   # block-diagonal K_cc: invert the 6 x 6 blocks
    Kb = Kcc.tobsr(blocksize=(6, 6))
    assert (np.diff(Kb.indptr) == 1).all()
    inv = sp.bsr_matrix((np.linalg.inv(Kb.data), Kb.indices, Kb.indptr), shape=Kb.shape).tocsr()
    S = (Kvv - Kvc @ inv @ Kvc.T).tocsr()
b = f[v]            # no load on centres, no prescribed displacement
free = np.flatnonzero(~fixed)
na = int(np.ceil(n / g))
cell = np.minimum((xyz / g).astype(int), na - 1)
agg = (cell[:, 0] * na + cell[:, 1]) * na + cell[:, 2]
Zfull = rigid_modes(xyz, agg, na ** 3)
Zf = Zfull[free]
Kff = K[free][:, free]
keep = np.flatnonzero(np.asarray(abs(Zf).sum(axis=0)).ravel() > 0)
Ac_K = (Zf[:, keep].T @ Kff @ Zf[:, keep]).toarray()
Zv = Zfull[v][:, keep].tocsr()
Ac_S = (Zv.T @ S @ Zv).toarray()
reg = lambda A: A + 1e-12 * np.trace(A) / len(A) * np.eye(len(A))
cf = {"Z^T K Z": sla.cho_factor(reg(Ac_K)), "Zv^T S Zv": sla.cho_factor(reg(Ac_S))}

dK = Kvv.diagonal()
dS = S.diagonal()
Sb = S.tobsr(blocksize=(6, 6))
nv = len(v) // 6
assert len(v) % 6 == 0
blocks = np.zeros((nv, 6, 6))
for i in range(nv):
    js = Sb.indices[Sb.indptr[i]:Sb.indptr[i + 1]]
    blocks[i] = Sb.data[Sb.indptr[i]:Sb.indptr[i + 1]][js == i].sum(axis=0)
binv = np.linalg.inv(blocks)
Kvb = Kvv.tobsr(blocksize=(6, 6))
kblocks = np.zeros((nv, 6, 6))
for i in range(nv):
    js = Kvb.indices[Kvb.indptr[i]:Kvb.indptr[i + 1]]
    kblocks[i] = Kvb.data[Kvb.indptr[i]:Kvb.indptr[i + 1]][js == i].sum(axis=0)
kbinv = np.linalg.inv(kblocks)
fine = {"diag(K_vv)": lambda r: r / dK, "diag(S)": lambda r: r / dS,
        "blocks(K_vv)": lambda r: np.einsum("nij,nj->ni", kbinv, r.reshape(nv, 6)).ravel(),
        "blocks(S)": lambda r: np.einsum("nij,nj->ni", binv, r.reshape(nv, 6)).ravel()}
x_ref = None
print(f"BCC {n}^3, r = 0.05: {lat.n_beams} struts, {len(v)} Schur dofs, {len(keep)} coarse dofs ({g}^3-cell aggregates)", flush=True)
for cname, cfac in cf.items():
    for fname, fn in fine.items():
        x, it = pcg(S, b, lambda r, fn=fn, cfac=cfac: fn(r) + Zv @ sla.cho_solve(cfac, Zv.T @ r))
        if x_ref is None:
            x_ref = x
        assert np.linalg.norm(x - x_ref) < 1e-5 * np.linalg.norm(x_ref)
        print(f"  coarse {cname:10s} fine {fname:12s}: {it} iterations", flush=True)
# for reference: the un-condensed system
dff = Kff.diagonal()
cfK = cf["Z^T K Z"]
Zk = Zf[:, keep].tocsr()
x, it = pcg(Kff, f[free], lambda r: r / dff + Zk @ sla.cho_solve(cfK, Zk.T @ r))
print(f"  un-condensed K, Jacobi + rigid: {it} iterations")

# ---- three levels as on the device: dense rigid level on large aggregates + a tile level (small aggregates, 6 rigid or
# 12 rigid + uniform-strain modes, one small SPD block per tile) + fine level
if len(sys.argv) > 3:
    gt = int(sys.argv[3])                      # tile edge in cells
    nt = int(np.ceil(n / gt))
    tcell = np.minimum((xyz / gt).astype(int), nt - 1)
    tile = (tcell[:, 0] * nt + tcell[:, 1]) * nt + tcell[:, 2]

    def tile_modes(affine):
        cen = np.stack([np.bincount(tile, xyz[:, k], nt ** 3) for k in range(3)], 1) / np.maximum(np.bincount(tile, minlength=nt ** 3), 1)[:, None]
        r = xyz - cen[tile]
        m = 12 if affine else 6
        rows, cols, vals = [], [], []

        def put(node_dof, mode, val):
            rows.append(6 * np.arange(N) + node_dof)
            cols.append(m * tile + mode)
            vals.append(val * np.ones(N))
        for k in range(3):
            put(k, k, 1.0)
            a, b2 = (k + 1) % 3, (k + 2) % 3
            put(b2, 3 + k, r[:, a])
            put(a, 3 + k, -r[:, b2])
            put(3 + k, 3 + k, 1.0)
        if affine:   # six uniform strains: u = eps (x - c), no rotation
            pairs = [(0, 0), (1, 1), (2, 2), (0, 1), (1, 2), (0, 2)]
            for q, (a, b2) in enumerate(pairs):
                put(a, 6 + q, r[:, b2] if a == b2 else 0.5 * r[:, b2])
                if a != b2:
                    put(b2, 6 + q, 0.5 * r[:, a])
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(6 * N, m * nt ** 3))

    print(f"three levels: dense level {g}^3-cell aggregates, tile level {gt}^3 cells", flush=True)
    for affine in (False, True):
        Zt = tile_modes(affine)[v]
        kt = np.flatnonzero(np.asarray(abs(Zt).sum(axis=0)).ravel() > 0)
        Zt = Zt[:, kt].tocsr()
        m = 12 if affine else 6
        for op_name, op in (("S", S),):
            Bt = (Zt.T @ op @ Zt).toarray()
            # block diagonal part only (one block per tile)
            tid = kt // m
            mask = tid[:, None] == tid[None, :]
            cft = sla.cho_factor(reg(Bt * mask))
            for fname in ("diag(K_vv)", "blocks(K_vv)", "blocks(S)"):
                fn = fine[fname]
                cfac = cf["Z^T K Z"]
                x, it = pcg(S, b, lambda r: fn(r) + Zt @ sla.cho_solve(cft, Zt.T @ r) + Zv @ sla.cho_solve(cfac, Zv.T @ r))
                assert np.linalg.norm(x - x_ref) < 1e-5 * np.linalg.norm(x_ref)
                print(f"  tile modes {m:2d}  fine {fname:12s}: {it} iterations", flush=True)
