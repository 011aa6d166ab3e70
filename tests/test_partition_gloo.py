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
    out.put((rank, slab.node_xyz, y, u, len(slab.beam_conn), nsg))
    dist.barrier()
    dist.destroy_process_group()


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
    for rank, xyz, y, u, _, _ in res:
        ids = np.array([key[tuple(np.round(p, 9))] for p in xyz])
        assert np.linalg.norm(y - yref[ids]) / np.linalg.norm(yref) < 1e-12
        assert np.linalg.norm(u - uref[ids]) / np.linalg.norm(uref) < 1e-7
