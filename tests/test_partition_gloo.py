"""N>1 path on CPU: 2 processes (gloo) each build their slab, apply the LOCAL condensed operator with the CPU
oracle, sum interface forces with an all-reduce on the packed interface vector (the same packing the RCCL path of
libpylattice_hip uses), and run the weighted-dot PCG.  Compared with the un-partitioned lattice."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import c_oracle
from pylatticedso_amd import lattice_arrays as LA
from pylatticedso_amd import partition as PT

E, NU = 1013.0, 0.3
CASES = [(["Octet"], [0.03], (3, 4, 2), 1), (["BCC"], [0.05], (4, 2, 2), 0), (["BCC", "Octet"], [0.04, 0.03], (2, 5, 2), 1)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_problem(geom, radii, ncell):
    lat = LA.generate((1, 1, 1), ncell, geom, radii)
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    sc = c_oracle.condense_unique(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
    return lat, sc


def _worker(rank, world, port, case, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    geom, radii, ncell, axis = CASES[case]
    slab = PT.build_slab((1, 1, 1), ncell, geom, radii, rank, world, axis=axis)
    keys = [None] * world
    dist.all_gather_object(keys, slab.iface_key)
    ok, gid, nsg = PT.global_interface_ids(keys, rank)
    loc = slab.iface_local[ok]
    sc = c_oracle.condense_unique(slab.beam_radius, slab.seg_len, slab.seg_nsub, E, NU)
    n = len(slab.node_xyz)

    def sum_shared(y):
        pack = torch.zeros(nsg, 6, dtype=torch.float64)
        pack[gid] = torch.from_numpy(y[loc])
        dist.all_reduce(pack)
        y[loc] = pack[gid].numpy()
        return y

    w = sum_shared(np.ones((n, 6)))
    w = 1.0 / w

    # the neighbour exchange of libpylattice_hip (pl_dist_set_peers): per peer the common nodes in the order of their
    # global interface ids, partial rows sent to the rank on the other side, received rows added - must give exactly
    # what the all-reduce over all planes gives
    peer = slab.iface_peer[ok]

    def sum_shared_p2p(y):
        y = y.copy()
        reqs, recv = [], {}
        for pr in np.unique(peer):
            sel = np.flatnonzero(peer == pr)
            rows = loc[sel][np.argsort(gid[sel], kind="stable")]
            recv[int(pr)] = (rows, torch.zeros(len(rows), 6, dtype=torch.float64))
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(y[rows])), dst=int(pr)))
            reqs.append(dist.irecv(recv[int(pr)][1], src=int(pr)))
        for q in reqs:
            q.wait()
        for rows, buf in recv.values():
            y[rows] += buf.numpy()
        return y

    probe = np.random.default_rng(rank).standard_normal((n, 6))
    assert np.array_equal(sum_shared_p2p(probe), sum_shared(probe.copy()))
    assert set(np.unique(peer)) <= {rank - 1, rank + 1} and len(peer) == len(loc)

    def wdot(a, b):
        t = torch.tensor([float((w * a * b).sum())], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    def A(x):
        return sum_shared(c_oracle.spmv(slab.node_xyz, slab.beam_conn, sc, x))

    # 1) operator: x = smooth global function of the coordinates -> identical on shared nodes
    xyz = slab.node_xyz
    x = np.stack([np.sin(xyz @ [1.0, 2.0, 3.0] + k) for k in range(6)], axis=1)
    y = A(x)
    # 2) cantilever PCG (clamp Xmin, load -0.1 in z on Xmax), Jacobi, weighted dots
    fixed = np.zeros((n, 6), bool)
    fixed[xyz[:, 0] == 0.0] = True
    tgt = xyz[:, 0] == float(ncell[0])
    ntg = torch.tensor([float((w[:, 2] * tgt).sum())], dtype=torch.float64)
    dist.all_reduce(ntg)
    f = np.zeros((n, 6))
    f[tgt, 2] = -0.1 / float(ntg[0])
    # plain CG is enough for this small check.  As in libpylattice_hip's RCCL path, p.(A p) is formed from the LOCAL
    # partial products without multiplicity weights and rides in the same all-reduce as the interface rows; only
    # the dots of assembled vectors (r.r) use the 1/multiplicity weights.
    m = (~fixed).astype(float)
    r = m * f
    p = r.copy()
    u = np.zeros((n, 6))
    rr = wdot(r, r)
    bb = rr
    for it in range(5000):
        Ap = m * c_oracle.spmv(slab.node_xyz, slab.beam_conn, sc, p)         # local partial product
        pack = torch.zeros(nsg * 6 + 1, dtype=torch.float64)
        pack[:nsg * 6].view(nsg, 6)[gid] = torch.from_numpy(Ap[loc])
        pack[-1] = float((p * Ap).sum())
        dist.all_reduce(pack)                                                  # ONE collective: rows + scalar
        Ap[loc] = pack[:nsg * 6].view(nsg, 6)[gid].numpy()
        alpha = rr / float(pack[-1])
        u += alpha * p
        r -= alpha * Ap
        rr_new = wdot(r, r)
        if rr_new <= 1e-24 * bb:
            break
        p = r + (rr_new / rr) * p
        rr = rr_new
    two = _two_level_forms(slab, sc, xyz, w, m, f, ncell, axis, sum_shared, sum_shared_p2p)
    out.put((rank, slab.node_xyz, y, u, len(slab.beam_conn), nsg, two))
    dist.barrier()
    dist.destroy_process_group()


def _two_level_forms(slab, sc, xyz, w, m, f, ncell, axis, sum_shared, sum_shared_p2p):
    """The two-level PCG (Jacobi + rigid-body modes of slices of the lattice, dense coarse operator replicated on every
    rank) in the ordinary form and in the single-reduction form of libpylattice_hip (pl_cg1.h: Chronopoulos-Gear
    recurrences, Z^T r by recurrence, ONE all-reduce per iteration besides the interface exchange).  Returns both
    solutions, iteration counts and the number of all-reduces per iteration of the new form."""
    n = len(xyz)
    counts = {"allreduce": 0}

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
        dist.all_reduce(t)
        counts["allreduce"] += 1
        return t.numpy()

    def K_local(x):
        return m * c_oracle.spmv(slab.node_xyz, slab.beam_conn, sc, m * x)

    # aggregates: slices along x of the WHOLE lattice (the same on every rank), modes about fixed reference points
    nagg = max(2, int(ncell[0]))
    agg = np.minimum((xyz[:, 0] / float(ncell[0]) * nagg).astype(int), nagg - 1)
    cen = np.stack([(np.arange(nagg) + 0.5) * ncell[0] / nagg, np.full(nagg, ncell[1] / 2.0),
                    np.full(nagg, ncell[2] / 2.0)], axis=1)
    Z = np.zeros((n, 6, 6 * nagg))
    d = xyz - cen[agg]
    for i in range(n):
        c0 = 6 * agg[i]
        Z[i, :3, c0:c0 + 3] = np.eye(3)
        Z[i, 3:, c0 + 3:c0 + 6] = np.eye(3)
        rx, ry, rz = d[i]
        Z[i, :3, c0 + 3:c0 + 6] = [[0, rz, -ry], [-rz, 0, rx], [ry, -rx, 0]]     # omega x (x - c)
    Z *= m[:, :, None]
    Zf = Z.reshape(n * 6, -1)
    KZ = np.stack([sum_shared(K_local(Zf[:, j].reshape(n, 6))).ravel() for j in range(Zf.shape[1])], axis=1)
    Ac = allreduce(Zf.T @ (w.ravel()[:, None] * KZ))
    Ac = 0.5 * (Ac + Ac.T) + 1e-12 * np.trace(Ac) / len(Ac) * np.eye(len(Ac))   # slices without free dofs: empty rows
    diag = sum_shared(_local_diag(slab, sc, n))
    dinv = np.where(m > 0, 1.0 / diag, 0.0)
    restrict = lambda v: Zf.T @ (w * v).ravel()            # local share of Z^T v for an ASSEMBLED v
    bb = float(allreduce(np.array([(w * f * m * f).sum()]))[0])

    # ordinary form: K p exchange, p.Ap, then [Z^T r | r.D^-1 r | r.r]
    x = np.zeros((n, 6)); r = m * f; p = np.zeros((n, 6)); rz_old = 0.0
    for it0 in range(3000):
        red = allreduce(np.concatenate([restrict(r), [(w * dinv * r * r).sum(), (w * r * r).sum()]]))
        if red[-1] <= 1e-24 * bb:
            break
        yc = np.linalg.solve(Ac, red[:-2])
        z = dinv * r + (Zf @ yc).reshape(n, 6)
        rz = red[-2] + red[:-2] @ yc
        p = z + (rz / rz_old if it0 else 0.0) * p
        Ap = K_local(p)
        pap = float(allreduce(np.array([(p * Ap).sum()]))[0])
        Ap = sum_shared_p2p(Ap)
        alpha = rz / pap
        x += alpha * p
        r -= alpha * Ap
        rz_old = rz
    x_ord = x

    # single-reduction form
    x = np.zeros((n, 6)); r = m * f; p = np.zeros((n, 6)); s = np.zeros((n, 6))
    rc = allreduce(restrict(r))                              # once, before the loop
    sc_ = np.zeros_like(rc)
    rc0 = np.abs(rc).max()
    g_old = a_old = 0.0
    per_iter = []
    for it1 in range(3000):
        before = counts["allreduce"]
        yc = np.linalg.solve(Ac, rc)
        u = dinv * r + (Zf @ yc).reshape(n, 6)
        wl = K_local(u)
        delta_loc = (u * wl).sum()                           # LOCAL partial product, no weights
        wv = sum_shared_p2p(wl)                              # neighbour exchange of the interface rows
        red = allreduce(np.concatenate([restrict(wv), [delta_loc, (w * dinv * r * r).sum(), (w * r * r).sum()]]))
        per_iter.append(counts["allreduce"] - before)
        wc, delta, gamma, rr = red[:-3], red[-3], red[-2] + rc @ yc, red[-1]
        if rr <= 1e-24 * bb:
            break
        beta = gamma / g_old if it1 else 0.0
        alpha = gamma / (delta - (beta * gamma / a_old if it1 else 0.0))
        p = u + beta * p
        s = wv + beta * s
        x += alpha * p
        r -= alpha * s
        sc_ = wc + beta * sc_
        rc = rc - alpha * sc_
        g_old, a_old = gamma, alpha
    drift = np.abs(allreduce(restrict(r)) - rc).max() / rc0     # relative to the initial coarse residual
    return x_ord, it0, x, it1, max(per_iter), drift


def _local_diag(slab, sc, n):
    """diag of the LOCAL operator, by probing with unit vectors per dof kind (6 products - struts couple different
    nodes, so e_k on every node at once picks up off-diagonal node blocks: use the 6 x 6 node blocks' own diagonal
    through the oracle's assembled matrix instead)."""
    from oracle import timoshenko_oracle as O
    K = O.assemble_condensed(slab.node_xyz, slab.beam_conn, sc)
    return np.asarray(K.diagonal()).reshape(n, 6)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_two_rank_slab_operator_and_pcg(case):
    geom, radii, ncell, axis = CASES[case]
    lat, sc = _global_problem(geom, radii, ncell)
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[4] for r in res) == lat.n_beams          # every strut owned exactly once
    assert res[0][5] > 0
    x = np.stack([np.sin(lat.node_xyz @ [1.0, 2.0, 3.0] + k) for k in range(6)], axis=1)
    yref = c_oracle.spmv(lat.node_xyz, lat.beam_conn, sc, x)
    fixed = np.zeros((lat.n_nodes, 6), bool)
    fixed[lat.node_xyz[:, 0] == 0.0] = True
    tgt = lat.node_xyz[:, 0] == float(ncell[0])
    f = np.zeros((lat.n_nodes, 6))
    f[tgt, 2] = -0.1 / tgt.sum()
    uref, it, _ = c_oracle.pcg(lat.node_xyz, lat.beam_conn, sc, fixed, np.zeros_like(f), f, rtol=1e-13)
    key = {tuple(np.round(p, 9)): i for i, p in enumerate(lat.node_xyz)}
    for rank, xyz, y, u, _, _, two in res:
        ids = np.array([key[tuple(np.round(p, 9))] for p in xyz])
        assert np.linalg.norm(y - yref[ids]) / np.linalg.norm(yref) < 1e-12
        assert np.linalg.norm(u - uref[ids]) / np.linalg.norm(uref) < 1e-7
        # two-level PCG, ordinary and single-reduction form (libpylattice_hip opts.cg_form = 1)
        x_ord, it_ord, x_one, it_one, reductions, drift = two
        assert np.linalg.norm(x_ord - uref[ids]) / np.linalg.norm(uref) < 1e-7
        assert np.linalg.norm(x_one - uref[ids]) / np.linalg.norm(uref) < 1e-7
        assert reductions == 1                       # ONE all-reduce per iteration (+ the neighbour exchange)
        assert it_one <= it_ord * 1.03 + 2
        assert drift < 1e-10                         # Z^T r carried by recurrence stays the restriction of r
