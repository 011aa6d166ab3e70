"""CPU oracle (TEST INFRASTRUCTURE ONLY) — numpy/scipy restatement of pyLatticeDSO's beam FEM.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``pylatticedso_amd``) never does.

Parity status: PINNED by the reference's committed dolfinx outputs
``data/outputs/schur_complement/Schur_complement_{BCC,Hybrid1,Hybrid4}.npz`` (subset under
``tests/golden/schur_*.npz``) — see ``tests/test_oracle_golden.py``.  The reference FEM path itself
(dolfinx 0.9.0 / PETSc / gmsh, pyproject.toml:22-30) is not installable here, so what is restated is
the weak form the reference hands to dolfinx plus gmsh's 1-D subdivision rule.

What each function follows (paths relative to the reference repo):

* ``section_constants``      src/pyLatticeSim/material_definition.py:44-45,122-156  (kappa=0.9, G=E/2(1+nu))
* ``local_frame``            src/pyLatticeSim/beam_model.py:197-216
* ``sub_element_stiffness``  src/pyLatticeSim/simulation_base.py:141-156 (strains), :190-197 (shear terms use
                             quadrature_degree 1 = mid-point rule), :220-225 (bilinear form)
* ``gmsh_subdivisions``      src/pyLatticeSim/lattice_generation.py:50-64 (h = 0.05*cell_size_x) + gmsh 1-D mesher
* ``assemble_submeshed``     src/pyLatticeSim/lattice_generation.py:134-175, simulation_base.py:465-476
* ``schur_complement``       src/pyLatticeSim/schur_complement.py:75-147
* ``solve_dirichlet``        src/pyLatticeSim/simulation_base.py:465-514 (assemble with bcs, lifting, point loads, LU)
* ``reference_cg``           src/pyLatticeSim/conjugate_gradient_solver.py:15-122
* ``submesh_vertices`` / ``homogenize_submeshed``  src/pyLatticeSim/homogenization_cell.py:112-147,200-252,
                             309-331,367-376,405-436 (no committed homogenisation output in the reference and
                             dolfinx_mpc is absent; pinned through ``homogenize_from_schur`` instead: the periodic
                             homogenised matrix follows from the cell's Schur complement on its boundary nodes alone,
                             and the reference commits those - tests/golden/homogenized_from_schur.npz)
* ``homogenize_from_schur``  the same procedure on the condensed cell: what the reference's own dolfinx / PETSc Schur
                             complements (data/outputs/schur_complement/*.npz) say the homogenised matrix is
* ``condensed_beam`` / ``beam_matrix``  closed-form static condensation of the above (derivation in DESIGN.md §3)
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

KAPPA = 0.9          # material_definition.py:45
PENALIZATION = 1.5   # beam.py:71 / lattice_sim.py:112
MESH_FRACTION = 0.05  # lattice_generation.py:50 (mesh_element_lenght)


# --------------------------------------------------------------------------------------------
# element
# --------------------------------------------------------------------------------------------
def section_constants(radius, E, nu, kappa=KAPPA):
    """C = [ES, kGS, kGS, GJ, EI, EI] for a circular section (material_definition.py:113,142-156)."""
    G = E / (2.0 * (1.0 + nu))
    S = math.pi * radius ** 2
    I = math.pi * radius ** 4 / 4.0
    J = 2.0 * I
    return np.array([E * S, G * kappa * S, G * kappa * S, G * J, E * I, E * I])


def local_frame(t):
    """(t, a1, a2) exactly as beam_model.py:197-216 builds them from the element tangent."""
    t = np.asarray(t, dtype=float)
    t = t / np.linalg.norm(t)
    ex, ey, ez = np.eye(3)
    e1 = ey if abs(t[1]) < abs(t[0]) else ex
    te1 = float(t @ e1)
    e2 = ez if abs(t[2]) < abs(te1) else e1
    a1 = np.cross(t, e2)
    a1 /= np.linalg.norm(a1)
    a2 = np.cross(t, a1)
    a2 /= np.linalg.norm(a2)
    return t, a1, a2


def sub_element_stiffness(xa, xb, C):
    """12x12 stiffness of ONE P1xP1 line element A->B; DOF order [wA(3), thA(3), wB(3), thB(3)].

    strains (simulation_base.py:141-156), with w' = (wB-wA)/l, th' = (thB-thA)/l and th_m = (thA+thB)/2
    for the two shear rows (integrated with the mid-point rule, simulation_base.py:193-195):
        e0 = w'.t   e1 = w'.a1 - th_m.a2   e2 = w'.a2 + th_m.a1   e3 = th'.t   e4 = th'.a1   e5 = th'.a2
    K_e = l * sum_i C_i b_i b_i^T
    """
    xa = np.asarray(xa, float)
    xb = np.asarray(xb, float)
    d = xb - xa
    l = float(np.linalg.norm(d))
    t, a1, a2 = local_frame(d)
    Bm = np.zeros((6, 12))
    # w' rows
    Bm[0, 0:3], Bm[0, 6:9] = -t / l, t / l
    Bm[1, 0:3], Bm[1, 6:9] = -a1 / l, a1 / l
    Bm[2, 0:3], Bm[2, 6:9] = -a2 / l, a2 / l
    # mid-point rotations in the shear rows
    Bm[1, 3:6], Bm[1, 9:12] = -0.5 * a2, -0.5 * a2
    Bm[2, 3:6], Bm[2, 9:12] = 0.5 * a1, 0.5 * a1
    # th' rows
    Bm[3, 3:6], Bm[3, 9:12] = -t / l, t / l
    Bm[4, 3:6], Bm[4, 9:12] = -a1 / l, a1 / l
    Bm[5, 3:6], Bm[5, 9:12] = -a2 / l, a2 / l
    return l * (Bm.T * C) @ Bm


def gmsh_subdivisions(length, h):
    """Number of equal P1 elements gmsh puts on a straight line of ``length`` with uniform size ``h``.

    gmsh's 1-D mesher integrates 1/h along the curve (a = length/h for a uniform field) and uses
    N_points = int(a + 1.99), i.e. n_elements = int(a + 0.99) (>= 1).  ``ceil(a)`` agrees except for
    frac(a) in (0, 0.01].  Pinned by tests/golden/schur_*.npz (round/floor miss by 1e-5..1e-4).
    """
    a = length / h
    return max(1, int(a + 0.99))


# --------------------------------------------------------------------------------------------
# reference-faithful sub-meshed assembly
# --------------------------------------------------------------------------------------------
def assemble_submeshed(node_xyz, seg_conn, seg_radius, E, nu, h, kappa=KAPPA):
    """Global K (CSR, 6 DOF/vertex) on the gmsh-like sub-mesh of every segment.

    ``seg_conn``/``seg_radius`` are the lattice's (already penalised) segments with their ACTUAL radius
    (penalised segments carry 1.5 r, beam.py:405-411).  Vertices 0..N-1 are the lattice nodes, the
    interior sub-nodes of each segment follow.  Returns (K, n_vertices).
    """
    node_xyz = np.asarray(node_xyz, float)
    N = len(node_xyz)
    rows, cols, vals = [], [], []
    nv = N
    for (ia, ib), r in zip(np.asarray(seg_conn), np.asarray(seg_radius)):
        xa, xb = node_xyz[ia], node_xyz[ib]
        L = float(np.linalg.norm(xb - xa))
        n = gmsh_subdivisions(L, h)
        C = section_constants(r, E, nu, kappa)
        ids = [ia] + list(range(nv, nv + n - 1)) + [ib]
        nv += n - 1
        Ke = sub_element_stiffness(xa, xa + (xb - xa) / n, C)  # identical for every sub-element of the line
        for e in range(n):
            dofs = np.r_[6 * ids[e] + np.arange(6), 6 * ids[e + 1] + np.arange(6)]
            rows.append(np.repeat(dofs, 12))
            cols.append(np.tile(dofs, 12))
            vals.append(Ke.ravel())
    K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(6 * nv, 6 * nv)).tocsr()
    return K, nv


def assemble_submeshed_fast(node_xyz, seg_conn, seg_radius, E, nu, h, kappa=KAPPA):
    """Same matrix and vertex numbering as assemble_submeshed, built with array operations (one 12x12 per SEGMENT,
    repeated over its sub-elements) - the form bench.py's reference-faithful CPU leg times, so that the Python loop
    above does not count against the reference's compiled dolfinx assembly."""
    node_xyz = np.asarray(node_xyz, float)
    conn = np.asarray(seg_conn, dtype=np.int64)
    rad = np.asarray(seg_radius, float)
    N, S = len(node_xyz), len(conn)
    xa, xb = node_xyz[conn[:, 0]], node_xyz[conn[:, 1]]
    d = xb - xa
    L = np.sqrt((d * d).sum(axis=1))
    n = np.maximum(1, np.floor(L / h + 0.99).astype(np.int64))
    l = L / n
    # frames (beam_model.py:197-216), vectorised local_frame
    t = d / L[:, None]
    ex, ey, ez = np.eye(3)
    e1 = np.where((np.abs(t[:, 1]) < np.abs(t[:, 0]))[:, None], ey, ex)
    te1 = (t * e1).sum(axis=1)
    e2 = np.where((np.abs(t[:, 2]) < np.abs(te1))[:, None], ez, e1)
    a1 = np.cross(t, e2)
    a1 /= np.sqrt((a1 * a1).sum(axis=1))[:, None]
    a2 = np.cross(t, a1)
    a2 /= np.sqrt((a2 * a2).sum(axis=1))[:, None]
    G = E / (2.0 * (1.0 + nu))
    Sx = math.pi * rad ** 2
    I = math.pi * rad ** 4 / 4.0
    C = np.stack([E * Sx, G * kappa * Sx, G * kappa * Sx, G * 2.0 * I, E * I, E * I], axis=1)      # (S, 6)
    Bm = np.zeros((S, 6, 12))
    il = (1.0 / l)[:, None]
    for row, v in ((0, t), (1, a1), (2, a2)):
        Bm[:, row, 0:3], Bm[:, row, 6:9] = -v * il, v * il
        Bm[:, row + 3, 3:6], Bm[:, row + 3, 9:12] = -v * il, v * il
    Bm[:, 1, 3:6] = Bm[:, 1, 9:12] = -0.5 * a2
    Bm[:, 2, 3:6] = Bm[:, 2, 9:12] = 0.5 * a1
    Ke = np.einsum("s,sik,si,sil->skl", l, Bm, C, Bm)                                              # (S, 12, 12)
    # vertex ids of every sub-element: lattice nodes first, interior sub-nodes segment by segment
    first_new = N + np.concatenate([[0], np.cumsum(n - 1)[:-1]])
    nel = int(n.sum())
    seg_of = np.repeat(np.arange(S), n)
    e_in = np.arange(nel) - np.repeat(np.cumsum(n) - n, n)
    va = np.where(e_in == 0, conn[seg_of, 0], first_new[seg_of] + e_in - 1)
    vb = np.where(e_in == n[seg_of] - 1, conn[seg_of, 1], first_new[seg_of] + e_in)
    dofs = np.concatenate([6 * va[:, None] + np.arange(6), 6 * vb[:, None] + np.arange(6)], axis=1)  # (nel, 12)
    rows = np.repeat(dofs, 12, axis=1).ravel()
    cols = np.tile(dofs, (1, 12)).ravel()
    nv = N + int((n - 1).sum())
    K = sp.coo_matrix((Ke[seg_of].ravel(), (rows, cols)), shape=(6 * nv, 6 * nv)).tocsr()
    return K, nv


def schur_complement(K, boundary_dofs):
    """S = K_BB - K_BI K_II^-1 K_IB (schur_complement.py:75-147), dense."""
    n = K.shape[0]
    bd = np.asarray(boundary_dofs)
    mask = np.ones(n, bool)
    mask[bd] = False
    it = np.flatnonzero(mask)
    K = K.tocsc()
    Kii = K[it][:, it].tocsc()
    Kib = K[it][:, bd]
    Kbb = K[bd][:, bd].toarray()
    lu = spla.splu(Kii)
    U = lu.solve(Kib.toarray())
    return Kbb - Kib.T @ U


def solve_dirichlet(K, fixed_mask, ubar, f):
    """Direct solve with dolfinx semantics (simulation_base.py:465-514): constrained rows/cols -> identity,
    RHS lifted by -K[:,c] ubar_c, b_c = ubar_c, then point loads added (also on constrained dofs).
    Returns the full displacement vector."""
    fixed_mask = np.asarray(fixed_mask, bool).ravel()
    ubar = np.asarray(ubar, float).ravel()
    f = np.asarray(f, float).ravel()
    free = np.flatnonzero(~fixed_mask)
    fix = np.flatnonzero(fixed_mask)
    K = K.tocsr()
    b = -(K[free][:, fix] @ ubar[fix]) + f[free]
    u = np.zeros(K.shape[0])
    u[fix] = ubar[fix] + f[fix]          # b.setValues(ADD) after set_bc (simulation_base.py:494-498)
    u[free] = spla.splu(K[free][:, free].tocsc()).solve(b)
    return u


# --------------------------------------------------------------------------------------------
# closed-form condensation: one 2-node element per lattice beam
# --------------------------------------------------------------------------------------------
def segment_flexibility(length, n, radius, E, nu, kappa=KAPPA):
    """Tip flexibility of a clamped chain of ``n`` identical sub-elements (total ``length``).

    returns (f_axial, f_torsion, f11, f12, f22) with the bending-plane block [[f11, f12], [f12, f22]]
    acting on (transverse force, bending moment) -> (deflection, rotation):
        f11 = L/(kGS) + L^3/(3EI) (1 - 1/(4 n^2)),  f12 = L^2/(2EI),  f22 = L/(EI)
    """
    ES, GS, _, GJ, EI, _ = section_constants(radius, E, nu, kappa)
    L = length
    return (L / ES, L / GJ,
            L / GS + L ** 3 / (3.0 * EI) * (1.0 - 1.0 / (4.0 * n * n)),
            L ** 2 / (2.0 * EI), L / EI)


def condensed_beam(radius, seg_len, seg_n, E, nu, kappa=KAPPA, pen=PENALIZATION):
    """5 stiffness scalars (ka, kt, a, b, c) of the lattice beam A->B made of up to three colinear
    segments [pen(L1) | mid | pen(L2)] in series (zero-length segments are skipped).

    (a, b, c) is the inverse of the summed, tip-transported bending flexibility at end B:
        F = sum_i T_i^T F_i T_i,  T_i = [[1, 0], [d_i, 1]],  d_i = distance from segment end to B
        [[a, -b], [-b, c]] = F^-1
    """
    radii = (pen * radius, radius, pen * radius)
    L = float(sum(seg_len))
    fa = ft = f11 = f12 = f22 = 0.0
    s = 0.0
    for l, n, r in zip(seg_len, seg_n, radii):
        if l <= 0.0:
            continue
        ga, gt, g11, g12, g22 = segment_flexibility(l, n, r, E, nu, kappa)
        d = L - (s + l)
        fa += ga
        ft += gt
        f11 += g11 + 2.0 * d * g12 + d * d * g22
        f12 += g12 + d * g22
        f22 += g22
        s += l
    det = f11 * f22 - f12 * f12
    return 1.0 / fa, 1.0 / ft, f22 / det, f12 / det, f11 / det


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def beam_matrix(scalars, d):
    """12x12 stiffness of the condensed beam with end-to-end vector d = xB - xA.

    End-B forces from the relative deformation (du = uB - uA + d x thA ... see DESIGN.md §3):
        F_B = ka (du.t) t + a du_perp + b t x dth
        M_B = kt (dth.t) t + c dth_perp - b t x du
    and F_A = -F_B, M_A = -M_B - d x F_B (equilibrium).
    """
    ka, kt, a, b, c = scalars
    d = np.asarray(d, float)
    L = np.linalg.norm(d)
    t = d / L
    P = np.eye(3) - np.outer(t, t)
    T = np.outer(t, t)
    Tx = _skew(t)
    Kbb = np.block([[ka * T + a * P, b * Tx], [-b * Tx, kt * T + c * P]])
    # e = q_B - R q_A ; R = [[I, -[d]x], [0, I]]   (uB_rigid = uA + thA x d = uA - d x thA)
    R = np.block([[np.eye(3), -_skew(d)], [np.zeros((3, 3)), np.eye(3)]])
    K = np.zeros((12, 12))
    K[6:, 6:] = Kbb
    K[:6, 6:] = -R.T @ Kbb
    K[6:, :6] = -Kbb @ R
    K[:6, :6] = R.T @ Kbb @ R
    return K


def assemble_condensed(node_xyz, conn, scalars):
    """Global K (CSR) with one condensed 2-node element per beam; scalars[B,5]."""
    rows, cols, vals = [], [], []
    node_xyz = np.asarray(node_xyz, float)
    for (ia, ib), sc in zip(np.asarray(conn), np.asarray(scalars)):
        Ke = beam_matrix(sc, node_xyz[ib] - node_xyz[ia])
        dofs = np.r_[6 * ia + np.arange(6), 6 * ib + np.arange(6)]
        rows.append(np.repeat(dofs, 12))
        cols.append(np.tile(dofs, 12))
        vals.append(Ke.ravel())
    n = 6 * len(node_xyz)
    return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()


def beam_apply(scalars, d, xa, xb):
    """Matrix-free product of one condensed beam: returns (fA[6], fB[6]) for end vectors xa, xb (each [u,th])."""
    ka, kt, a, b, c = scalars
    d = np.asarray(d, float)
    L = np.linalg.norm(d)
    t = d / L
    du = xb[:3] - xa[:3] + np.cross(d, xa[3:])
    dth = xb[3:] - xa[3:]
    dut = du @ t
    dtt = dth @ t
    FB = (ka - a) * dut * t + a * du + b * np.cross(t, dth)
    MB = (kt - c) * dtt * t + c * dth - b * np.cross(t, du)
    FA = -FB
    MA = -MB - np.cross(d, FB)
    return np.r_[FA, MA], np.r_[FB, MB]


# --------------------------------------------------------------------------------------------
# periodic homogenisation of one cell on the sub-meshed model
# --------------------------------------------------------------------------------------------
CORNER_TAGS = [[1000, 1001, 1002, 1003, 1004, 1005, 1006, 1007]]                       # lattice.py:612-614
EDGE_TAGS = [[102, 104, 106, 107], [100, 108, 105, 111], [101, 109, 103, 110]]
FACE_TAGS = [[10, 15], [11, 14], [12, 13]]


def submesh_vertices(node_xyz, seg_conn, h):
    """Coordinates of every vertex of ``assemble_submeshed``'s mesh, in its numbering."""
    node_xyz = np.asarray(node_xyz, float)
    out = [node_xyz]
    for ia, ib in np.asarray(seg_conn):
        xa, xb = node_xyz[ia], node_xyz[ib]
        n = gmsh_subdivisions(float(np.linalg.norm(xb - xa)), h)
        if n > 1:
            out.append(xa + (xb - xa) * (np.arange(1, n) / n)[:, None])
    return np.vstack(out)


def homogenize_submeshed(K, vert_xyz, node_tag):
    """6x6 ``homogenizeMatrix`` of one cell, following HomogenizedCell step by step on the sub-meshed K.

    ``node_tag[i]`` = Point.tag of lattice node i (vertices beyond ``len(node_tag)`` are sub-mesh vertices and carry
    none).  Periodic groups by boundary tag with the first tag of each group as master (:210-252); translations of the
    vertex at the mean of the mesh coordinates fixed when there is one (:335-376) - otherwise the first lattice node,
    where the reference would be left with a singular matrix; load case k: -K w_k (:200-206); macro stress =
    sum over tagged nodes of (K u_tot)[:3] (x) r (:309-331); rows [00, 11, 22, 10, 20, 21]; symmetrised (:435).
    Returns (C, C_unsymmetrised, [u_tot of the six cases, (n_vertices, 6)])."""
    vert_xyz = np.asarray(vert_xyz, float)
    nv = len(vert_xyz)
    tag = np.asarray(node_tag)
    master = np.arange(nv)
    for group in CORNER_TAGS + EDGE_TAGS + FACE_TAGS:
        m = np.flatnonzero(tag == group[0])
        for t in group[1:]:
            s_ = np.flatnonzero(tag == t)
            if len(m) and len(s_):
                assert len(m) == len(s_) == 1, "one node per boundary tag (what the reference's pairing supports)"
                master[s_[0]] = m[0]
    reps = np.unique(master)
    slot = np.full(nv, -1)
    slot[reps] = np.arange(len(reps))
    cols = (6 * slot[master][:, None] + np.arange(6)).ravel()
    P = sp.csr_matrix((np.ones(6 * nv), (np.arange(6 * nv), cols)), shape=(6 * nv, 6 * len(reps)))
    centre = vert_xyz.mean(axis=0)
    hit = np.flatnonzero(np.all(np.abs(vert_xyz - centre) <= 1e-6, axis=1))
    anchor = int(hit[0]) if len(hit) else 0
    fixed = 6 * slot[master[anchor]] + np.arange(3)
    free = np.setdiff1d(np.arange(P.shape[1]), fixed)
    Kr = (P.T @ K @ P).tocsc()
    lu = spla.splu(Kr[free][:, free].tocsc())
    boundary = np.flatnonzero(tag > 0)
    x, y, z = vert_xyz.T
    zero = np.zeros(nv)
    fields = [(x, zero, zero), (zero, y, zero), (zero, zero, z), (y, x, zero), (z, zero, x), (zero, z, y)]
    cols_C, u_tots = [], []
    for wx, wy, wz in fields:
        w = np.zeros((nv, 6))
        w[:, 0], w[:, 1], w[:, 2] = wx, wy, wz
        b = -(P.T @ (K @ w.ravel()))
        ur = np.zeros(P.shape[1])
        ur[free] = lu.solve(b[free])
        u_tot = w + (P @ ur).reshape(nv, 6)
        R = (K @ u_tot.ravel()).reshape(nv, 6)
        macro = R[boundary, :3].T @ vert_xyz[boundary]
        cols_C.append([macro[0, 0], macro[1, 1], macro[2, 2], macro[1, 0], macro[2, 0], macro[2, 1]])
        u_tots.append(u_tot)
    C_raw = np.array(cols_C).T
    return 0.5 * (C_raw + C_raw.T), C_raw, u_tots


# --------------------------------------------------------------------------------------------
# the reference's hand-written CG
# --------------------------------------------------------------------------------------------
def reference_cg(A, b, M=None, maxiter=100, tol=1e-5, mintol=1e-5, restart_every=1000, alpha_max=0.1,
                 callback=None):
    """Restatement of conjugate_gradient_solver.py:15-122 (x0 = 0, alpha clamp, restart, three stop tests)."""
    matvec = (lambda v: A @ v)
    prec = (lambda v: M @ v) if M is not None else (lambda v: v)
    x = np.zeros(b.shape[0])
    r = b - matvec(x)
    # NB: without a preconditioner the reference sets ``z = r`` (an alias, not a copy) and then updates r in
    # place, so a restart (p = z.copy()) picks up the CURRENT residual; with M it picks up the previous z.
    z = prec(r) if M is not None else r
    p = z.copy()
    rz_old = r @ z
    norm_b = np.linalg.norm(b)
    info = 1
    for k in range(maxiter):
        Ap = matvec(p)
        alpha = min(rz_old / (p @ Ap), alpha_max)
        x += alpha * p
        r -= alpha * Ap
        if callback is not None:
            callback(x)
        if k % restart_every == 0 and k > 0:
            p = z.copy()
        if np.linalg.norm(r) <= tol * norm_b:
            info = 0
            break
        if np.linalg.norm(p) < mintol * (np.linalg.norm(x) + 1e-12):
            info = 0
            break
        if alpha < 1e-6:
            info = 2
        z = prec(r) if M is not None else r
        rz_new = r @ z
        p = z + (rz_new / rz_old) * p
        rz_old = rz_new
    return x, info


def homogenize_from_schur(S, xyz_b, lo=(0.0, 0.0, 0.0), size=(1.0, 1.0, 1.0), tol=1e-9):
    """6x6 homogenised matrix of one periodic cell from its Schur complement S (6 n_b x 6 n_b, node-major
    [ux, uy, uz, thx, thy, thz]) on the boundary nodes ``xyz_b`` - the procedure of ``homogenize_submeshed``
    (homogenization_cell.py:200-252, 309-331, 405-436) after the exact elimination of every interior dof: the interior is
    free and unloaded in the fluctuation problem (the load -K w is the reaction to a displacement field imposed on ALL
    nodes, so total interior displacements minimise the energy for the given boundary values), hence
        energy = 1/2 u_B^T S u_B,  u_B = w_B + P v,  P^T S (w_B + P v) = 0,  sigma = sum_b (S u_B)_b[:3] (x) x_b.
    Periodic groups: boundary nodes with equal coordinates modulo the cell size; the translations of the first group are
    fixed (the stresses do not depend on which).  Returns (C symmetrised, C_raw)."""
    S = np.asarray(S, float)
    xyz_b = np.asarray(xyz_b, float)
    nb = len(xyz_b)
    rel = (xyz_b - np.asarray(lo)) / np.asarray(size)
    wrapped = np.where(np.abs(rel - 1.0) <= tol, 0.0, rel)
    key = np.round(wrapped / (10 * tol)).astype(np.int64)
    groups, master = {}, np.arange(nb)
    for i in range(nb):
        master[i] = groups.setdefault(tuple(key[i]), i)
    reps = np.unique(master)
    slot = np.full(nb, -1)
    slot[reps] = np.arange(len(reps))
    P = np.zeros((6 * nb, 6 * len(reps)))
    for i in range(nb):
        P[6 * i:6 * i + 6, 6 * slot[master[i]]:6 * slot[master[i]] + 6] = np.eye(6)
    free = np.setdiff1d(np.arange(P.shape[1]), np.arange(3))
    Sr = P.T @ S @ P
    x, y, z = xyz_b.T
    zero = np.zeros(nb)
    cols = []
    for wx, wy, wz in [(x, zero, zero), (zero, y, zero), (zero, zero, z), (y, x, zero), (z, zero, x), (zero, z, y)]:
        w = np.zeros((nb, 6))
        w[:, 0], w[:, 1], w[:, 2] = wx, wy, wz
        b = -(P.T @ (S @ w.ravel()))
        v = np.zeros(P.shape[1])
        v[free] = np.linalg.solve(Sr[np.ix_(free, free)], b[free])
        R = (S @ (w.ravel() + P @ v)).reshape(nb, 6)
        macro = R[:, :3].T @ xyz_b
        cols.append([macro[0, 0], macro[1, 1], macro[2, 2], macro[1, 0], macro[2, 0], macro[2, 1]])
    C_raw = np.array(cols).T
    return 0.5 * (C_raw + C_raw.T), C_raw
