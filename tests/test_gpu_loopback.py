"""The multi-rank solver path of libpylattice_hip with world = 2 / 4 / 8 on ONE GPU (loopback transport,
pylatticedso_amd/loopback.py): the same pack / exchange / weight / coarse-band / single-reduction / node-elimination
device code an RCCL run executes, one host thread per rank, contributions summed through device buffers.  Every case is
held to the single-handle solve of the un-partitioned lattice (which tests/test_gpu_parity.py holds to the oracle):
1e-8 relative L2 on displacements, written at each assert."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from pylatticedso_amd import _capi                      # noqa: E402
from pylatticedso_amd.loopback import LoopbackGroup, whole_lattice   # noqa: E402

E, NU = 1013.0, 0.3
LATTICES = {"octet16": ((16, 16, 16), ["Octet"], [0.03]), "bcc12": ((12, 12, 12), ["BCC"], [0.05])}
SMALL = dict(tile_nodes=32, coarse_max_dofs=600)
_cache = {}


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b)


def _reference(name):
    """Single-handle solve of the whole lattice (cantilever of bench.py), once per lattice."""
    if name not in _cache:
        cells, geom, radii = LATTICES[name]
        lat, pen = whole_lattice((1, 1, 1), cells, geom, radii)
        fixed = np.zeros((lat.n_nodes, 6), np.uint8)
        fixed[lat.node_xyz[:, 0] == 0.0] = 1
        tgt = lat.node_xyz[:, 0] == float(cells[0])
        f = np.zeros((lat.n_nodes, 6))
        f[tgt, 2] = -0.1 / tgt.sum()
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=3, condense=-1, **SMALL) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
            x = np.random.default_rng(7).standard_normal((lat.n_nodes, 6))
            y = dev.spmv_free(x)
        assert st["converged"] == 1
        _cache[name] = (lat, u, st["iterations"], x, y)
    return _cache[name]


def _body_load_reference(name):
    """The cantilever plus a small load on every node that carries none (single handle, no elimination)."""
    key = name + ":body"
    if key not in _cache:
        cells, geom, radii = LATTICES[name]
        lat, pen = whole_lattice((1, 1, 1), cells, geom, radii)
        fixed = np.zeros((lat.n_nodes, 6), np.uint8)
        fixed[lat.node_xyz[:, 0] == 0.0] = 1
        tgt = lat.node_xyz[:, 0] == float(cells[0])
        f = np.zeros((lat.n_nodes, 6))
        f[tgt, 2] = -0.1 / tgt.sum()
        f = f + np.array([1e-5, -2e-5, 5e-6, 0, 0, 0]) * (f[:, 2:3] == 0.0)
        with _capi.HipLattice(lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU,
                              precond=3, condense=-1, **SMALL) as dev:
            dev.set_bc(fixed, None, f)
            dev.assemble()
            u, st = dev.solve(rtol=1e-11, max_iter=20000)
        assert st["converged"] == 1
        _cache[key] = u
    return _cache[key]


def _group(name, world, axis, **kw):
    cells, geom, radii = LATTICES[name]
    opts = dict(SMALL)
    opts.update(kw)
    return LoopbackGroup((1, 1, 1), cells, geom, radii, world, axis=axis, young=E, poisson=NU, **opts)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name", ["octet16", "bcc12"])
@pytest.mark.parametrize("p2p", [True, False])
def test_partitioned_operator_equals_the_whole(name, world, p2p):
    """P K P x on R slabs (interface rows by neighbour exchange or by the all-planes all-reduce) = the single handle's."""
    lat, _, _, x, y = _reference(name)
    with _group(name, world, axis=0, precond=1, condense=-1, p2p=p2p) as g:
        fixed, f = g.cantilever(float(g.num_cells[0]))
        g.set_bc(fixed, None, f)
        g.assemble()
        yr = g.spmv_free(g.scatter(lat.node_xyz, x))
        assert _rel(g.gather(lat.node_xyz, yr), y) < 1e-13


CASES = {"precond3": dict(precond=3), "precond4": dict(precond=4), "jacobi": dict(precond=1),
         "cg_form1": dict(precond=3, cg_form=1), "precision1": dict(precond=3, precision=1),
         # single-reduction form with rigid-body levels only / with strain modes on both levels (round 4)
         "cg_form1_modes6": dict(precond=3, cg_form=1, tile_modes=6),
         "cg_form1_modes12": dict(precond=3, cg_form=1, tile_modes=12, coarse_modes=12),
         "precond4_precision1": dict(precond=4, precision=1), "precision2": dict(precond=3, precision=2),
         # node elimination on multi-rank handles (nodes shared with another rank stay unknowns)
         "condense": dict(precond=3, condense=1), "condense_precision1": dict(precond=3, condense=1, precision=1),
         "condense_precond4": dict(precond=4, condense=1),
         # the default K*p of a multi-rank handle overlaps the neighbour exchange with the interior tiles; -1 = one launch
         "no_overlap": dict(precond=3, overlap=-1), "condense_no_overlap": dict(precond=3, condense=1, overlap=-1),
         # record palette: the palette form of the LDS-resident K*p over the interface / interior tile lists
         "palette": dict(precond=3, palette=1), "condense_palette": dict(precond=3, condense=1, palette=1)}


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name", ["octet16", "bcc12"])
@pytest.mark.parametrize("case", list(CASES))
def test_partitioned_solve_equals_single_handle(name, world, case):
    lat, u0, it0, _, _ = _reference(name)
    opts = dict(condense=-1)
    opts.update(CASES[case])
    with _group(name, world, axis=0, **opts) as g:
        fixed, f = g.cantilever(float(g.num_cells[0]))
        if case.startswith("condense"):        # a body load, so that eliminated nodes carry load as well
            f = [ff + np.array([1e-5, -2e-5, 5e-6, 0, 0, 0]) * (ff[:, 2:3] == 0.0) for ff in f]
        g.set_bc(fixed, None, f)
        g.assemble()
        res = g.solve(rtol=1e-11, max_iter=20000)
        stats = [st for _, st in res]
        assert all(st["converged"] == 1 for st in stats)
        # every rank sees the same all-reduced history: same decisions, same counts
        assert len({st["iterations"] for st in stats}) == 1 and len({st["rel_residual"] for st in stats}) == 1
        u = g.gather(lat.node_xyz, [u for u, _ in res])        # (also checks that copies of shared nodes agree)
        if case.startswith("condense"):
            u0 = _body_load_reference(name)
            assert all(st["condensed_nodes"] > 0 for st in stats)
        assert _rel(u, u0) < 1e-8
        if case.startswith("cg_form1"):
            assert all(int(st["cg_form_used"]) == 1 for st in stats)
        if case.startswith("precision") or case.endswith("precision1"):
            assert all(int(st["precision_used"]) == CASES[case]["precision"] for st in stats)
        if case != "jacobi":
            # the multi-level preconditioner survives the partition: the levels leave out shared nodes and the dense level
            # coarsens, which costs iterations - but nowhere near the Jacobi count
            assert stats[0]["iterations"] < 2.5 * it0
        # a second solve on the same handles (new right-hand side: pl_set_bc on assembled handles is collective too)
        g.set_bc(fixed, None, [2.0 * ff for ff in f])
        res2 = g.solve(rtol=1e-11, max_iter=20000)
        assert _rel(g.gather(lat.node_xyz, [u for u, _ in res2]), 2.0 * u0) < 1e-8
        if case.startswith("condense"):
            assert all(st["condensed_nodes"] > 0 for _, st in res2)


@pytest.mark.parametrize("axis", [1, 2])
def test_partition_axis(axis):
    """Slabs along y and z (bench.py's weak-scaling layout is along y)."""
    lat, u0, _, _, _ = _reference("octet16")
    with _group("octet16", 4, axis=axis, precond=3, condense=-1) as g:
        fixed, f = g.cantilever(16.0)
        g.set_bc(fixed, None, f)
        g.assemble()
        res = g.solve(rtol=1e-11, max_iter=20000)
        assert _rel(g.gather(lat.node_xyz, [u for u, _ in res]), u0) < 1e-8


def test_missing_rank_is_an_error_not_a_hang():
    """A collective that one rank never joins: the group is broken when that rank's handle goes away and the others
    return PL_ERR_HIP instead of waiting for ever."""
    import threading
    g = _group("bcc12", 2, axis=0, precond=1, condense=-1)
    try:
        fixed, f = g.cantilever(12.0)
        g.set_bc(fixed, None, f)
        err = []

        def lonely():
            try:
                g.devs[0].assemble()          # collective: rank 1 never calls it
            except _capi.PlError as e:
                err.append(e)
        t = threading.Thread(target=lonely)
        t.start()
        import time
        time.sleep(1.0)
        g.devs[1].close()                     # breaks the group: rank 0's barrier returns
        g.devs[1] = None
        t.join(timeout=30.0)
        assert not t.is_alive() and len(err) == 1 and err[0].code == _capi.PL_ERR_HIP
    finally:
        g.close()


def test_a_rank_that_fails_outside_a_collective_lets_the_others_go():
    """One rank raises before it reaches the collective (bad argument): LoopbackGroup.each breaks the group, the other rank's
    pl_assemble returns instead of waiting 120 s, and the caller sees the first failure."""
    import time
    g = _group("bcc12", 2, axis=0, precond=1, condense=-1)
    try:
        fixed, f = g.cantilever(12.0)
        g.set_bc(fixed, None, f)

        def step(r):
            if r == 1:
                raise ValueError("rank 1 never calls pl_assemble")
            g.devs[r].assemble()
        t0 = time.time()
        with pytest.raises(ValueError):
            g.each(step)
        assert time.time() - t0 < 30.0
    finally:
        g.close()


def test_config2_as_eight_slabs_full_size():
    """BASELINE.json configs[2] AS NAMED: the 100^3 BCC lattice (8 M struts) cut into 8 x-slabs, one handle each, through the
    loopback transport - the device code of an 8-GPU run (neighbour exchange of the interface rows under the interior tiles,
    fused scalar all-reduces, all-reduced 6 144-dof dense level, node elimination on every rank) on one GPU.  Every rank
    converges with the same count (<= 400: the round-2 verdict's bar; 368 measured, 334 on one handle), shared nodes agree,
    and the assembled solution satisfies Clapeyron's theorem and the true residual of the un-partitioned operator's
    right-hand side (checked rank by rank through the partitioned operator)."""
    n = 100
    with LoopbackGroup((1, 1, 1), (n, n, n), ["BCC"], [0.05], 8, axis=0, young=E, poisson=NU, precond=3, palette=1) as g:
        assert g.n_beams == 8_000_000
        fixed, f = g.cantilever(float(n))
        g.set_bc(fixed, None, f)
        g.assemble()
        res = g.solve(rtol=1e-8, max_iter=5000)
        stats = [st for _, st in res]
        assert all(st["converged"] == 1 for st in stats)
        assert len({st["iterations"] for st in stats}) == 1 and stats[0]["iterations"] <= 400
        assert all(st["condensed_nodes"] > 100_000 for st in stats)
        us = [u for u, _ in res]
        # true residual through the partitioned operator (interface rows summed over the ranks): P (f - K u) = 0
        Ku = g.spmv_free(us)
        num = sum(float((((1 - fx) * (ff - np.asarray(k).reshape(-1, 6))) ** 2).sum()) for fx, ff, k in zip(fixed, f, Ku))
        den = sum(float((ff ** 2).sum()) for ff in f)
        # (rows of shared nodes appear on two ranks with the full value in both: a factor <= 2 on a 1e-8 bound)
        assert np.sqrt(num / den) < 1e-7
        # tip deflection: the loaded face moves down, the same on the ranks that share nothing with it is not required;
        # the clamped face does not move
        assert all(np.all(u[fx != 0] == 0.0) for u, fx in zip(us, fixed))
        assert min(float(u[:, 2].min()) for u in us) < 0.0


def test_config4_as_eight_slabs_full_size():
    """BASELINE.json configs[4] AS NAMED: the 200 x 200 x 50 BCC+Octet hybrid (64.2 M struts) cut into 8 x-slabs, fp32
    matrix-free PCG with fp64 residual refinement (precision = 1), one handle per slab through the loopback transport - the
    device code of the 8-GPU run on one GPU.  Every rank converges with the same count (330 measured in round 3, 324 on one
    handle), the Dirichlet rows are exact, and the TRUE fp64 residual of the assembled solution through the partitioned
    operator meets the tolerance."""
    with LoopbackGroup((1, 1, 1), (200, 200, 50), ["BCC", "Octet"], [0.04, 0.03], 8, axis=0, young=E, poisson=NU, precond=3,
                       palette=1, precision=1, tile_modes=6) as g:
        assert g.n_beams > 64_000_000
        fixed, f = g.cantilever(200.0)
        g.set_bc(fixed, None, f)
        g.assemble()
        res = g.solve(rtol=1e-8, max_iter=5000)
        stats = [st for _, st in res]
        assert all(st["converged"] == 1 for st in stats)
        assert all(st["precision_used"] == 1 for st in stats)
        assert len({st["iterations"] for st in stats}) == 1 and stats[0]["iterations"] <= 420
        us = [u for u, _ in res]
        Ku = g.spmv_free(us)
        num = sum(float((((1 - fx) * (ff - np.asarray(k).reshape(-1, 6))) ** 2).sum()) for fx, ff, k in zip(fixed, f, Ku))
        den = sum(float((ff ** 2).sum()) for ff in f)
        assert np.sqrt(num / den) < 1e-7
        assert all(np.all(u[fx != 0] == 0.0) for u, fx in zip(us, fixed))
        assert min(float(u[:, 2].min()) for u in us) < 0.0


@pytest.mark.parametrize("config,cells,extra", [(1, ["12"], []), (2, ["16"], []), (4, ["16", "16", "8"], [])])
def test_bench_force_dist_rehearsal(config, cells, extra):
    """The code path `python -m torch.distributed.run ... bench.py --gpus N --config C` takes on every rank - slab build,
    gloo bootstrap, RCCL communicator inside the library (pl_dist_init), collective pl_set_bc / pl_assemble / pl_solve,
    pl_time_kernel of the two collectives - with ONE rank (--force-dist), on small cubes of the three configurations: so that
    the launch the driver makes on an 8-GPU node cannot rot unseen.  The JSON line must carry the contract's keys and a
    converged solve."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + config), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", str(config), "--force-dist",
           "--steps", "2", "--warmup", "1", "--cells", *cells, "--cpu-cells", "0", "--no-e2e", "--no-streaming",
           "--large-cells", "0", *extra]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["converged"] == 1
    assert "collectives_ms" in line and set(line["collectives_ms"]) == {"interface_exchange", "coarse_allreduce"}
    assert "RCCL" in line["config"]["partition"]
