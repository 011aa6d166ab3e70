#!/usr/bin/env python3
"""bench.py — beams/s of "assembly + PCG solve" on synthetic periodic lattices (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 50x50x50 Octet, r = 0.03, cell size 1, VeroClear,
joint penalisation on, cantilever (all 6 dofs clamped on Xmin, total force -0.1 in Z on Xmax), fp64.
For N > 1 GPUs the lattice is 50 x (50 N) x 50 cut into N y-slabs of 50 layers (weak scaling: 3.03 M struts per
GPU; growing along y keeps the cantilever's aspect ratio, hence its conditioning, fixed); interface forces and
PCG dot products are all-reduced with RCCL inside libpylattice_hip.

One "step" = per-strut stiffness build (condensed records + Jacobi diagonal + coarse operator of the two-level
preconditioner and its dense factorisation) + explicit BSR(6x6) global-K assembly + matrix-free PCG solve to
||r|| <= rtol ||b|| — all on data already resident in HBM.
value = struts of the whole job * steps / time.  The roofline object prices the dominant kernel (K*p) with the
algorithmic bytes of SURVEY.md 8(d) and a HIP-event timing taken on the library's own stream; cpu_baseline is the
plain-C oracle (oracle/beam_pcg.c) on all host cores, plus the reference-faithful sub-meshed + sparse-LU leg, on
bounded samples of the same workload; end_to_end_s is the wall clock of the drop-in call site (LatticeSim +
solve_FEM_FenicsX) for the same lattice.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

E, NU = 1013.0, 0.3
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cantilever_bc(xyz, x_max, n_targets_global=None):
    """fixed / ubar / f arrays of the cantilever: clamp Xmin, total Fz = -0.1 spread over the Xmax nodes."""
    n = len(xyz)
    fixed = np.zeros((n, 6), np.uint8)
    fixed[xyz[:, 0] == 0.0] = 1
    tgt = xyz[:, 0] == x_max
    f = np.zeros((n, 6))
    cnt = n_targets_global if n_targets_global is not None else int(tgt.sum())
    f[tgt, 2] = -0.1 / cnt
    return fixed, f, tgt


def dev_kernel_name(kernel, reorder):
    k = kernel if kernel else (3 if reorder else 2)
    return {1: "k_spmv_atomic", 2: "k_spmv_gather", 3: "k_spmv_tile"}[k]


def cpu_baseline(cells, radius, rtol, splu_cells):
    """CPU legs on the host cores of this box, on bounded samples of the same workload (same lattice type / BCs, fewer
    cells).  (ii) of SURVEY 8(d): the plain-C oracle - condensed struts, matrix-free Jacobi-PCG - on ALL cores (OpenMP,
    thread count reported) and on one; (i): the reference-faithful discretisation - every penalised segment sub-meshed
    like gmsh does, scipy CSR assembly + SuperLU (stand-in for PETSc preonly/lu) - on a size it can finish."""
    from oracle import c_oracle, timoshenko_oracle as O
    from pylatticedso_amd import lattice_arrays as LA
    lat = LA.generate((1, 1, 1), (cells,) * 3, ["Octet"], [radius])
    pen = LA.penalize(lat, LA.compute_lzone(lat))
    fixed, f, _ = cantilever_bc(lat.node_xyz, float(cells))
    legs = {}
    u = None
    for name, mt in (("all_cores", True), ("one_core", False)):
        t0 = time.perf_counter()
        sc = c_oracle.condense_unique(lat.beam_radius, pen.seg_len, pen.seg_nsub, E, NU)
        u, it, rel = c_oracle.pcg(lat.node_xyz, lat.beam_conn, sc, fixed, np.zeros_like(f), f, rtol=rtol, maxit=100000,
                                  all_cores=mt)
        legs[name] = (lat.n_beams / (time.perf_counter() - t0), time.perf_counter() - t0, abs(it))
    threads = c_oracle.num_threads()
    out = {"value": legs["all_cores"][0], "unit": "beams/s", "cores": threads, "kind": "port",
           "sample": f"{cells}^3 Octet r={radius} cantilever, {lat.n_beams} struts, {legs['all_cores'][2]} Jacobi-PCG "
                     f"iterations to rtol {rtol:g} in {legs['all_cores'][1]:.1f} s (oracle/beam_pcg.c oracle_pcg_mt, gcc -O2 "
                     f"-fopenmp, {threads} threads of {os.cpu_count()} host CPUs; the GPU runs a multi-level PCG with "
                     f"4x fewer iterations - a different preconditioner, see config.preconditioner)",
           "one_core": {"value": legs["one_core"][0], "unit": "beams/s", "cores": 1,
                        "sample": f"same sample, oracle_pcg (scatter form), {legs['one_core'][1]:.1f} s"}}
    if splu_cells > 0:
        sl = LA.generate((1, 1, 1), (splu_cells,) * 3, ["Octet"], [radius])
        sp_ = LA.penalize(sl, LA.compute_lzone(sl))
        xyz, conn, rad = penalised_segments(sl, sp_)
        n_load = int((sl.node_xyz[:, 0] == float(splu_cells)).sum())
        t0 = time.perf_counter()
        K, nv = O.assemble_submeshed_fast(xyz, conn, rad, E, NU, 0.05)
        t1 = time.perf_counter()
        fx = np.zeros((nv, 6), bool)
        ff = np.zeros((nv, 6))
        fx[:sl.n_nodes][sl.node_xyz[:, 0] == 0.0] = True
        ff[:sl.n_nodes][sl.node_xyz[:, 0] == float(splu_cells), 2] = -0.1 / n_load
        O.solve_dirichlet(K, fx, np.zeros((nv, 6)), ff)
        t2 = time.perf_counter()
        out["reference_faithful"] = {
            "value": sl.n_beams / (t2 - t0), "unit": "beams/s", "kind": "port",
            "cores": "scipy SuperLU (1 thread)",
            "sample": f"{splu_cells}^3 Octet r={radius} cantilever, {sl.n_beams} struts -> {len(conn)} penalised segments "
                      f"-> {6 * nv} dofs on the gmsh-like sub-mesh: CSR assembly {t1 - t0:.1f} s + splu factor/solve "
                      f"{t2 - t1:.1f} s (oracle/timoshenko_oracle.py; stand-in for dolfinx + PETSc LU, which are not "
                      f"installable here)"}
    return out, u


def penalised_segments(lat, pen):
    """Explicit node / segment list of the penalised lattice (what the reference hands to gmsh): lattice nodes first,
    then the penalisation points; segments [pen@point1 | middle | pen@point2] with their actual radii."""
    B, nid = lat.n_beams, lat.n_nodes
    has1, has2 = pen.seg_len[:, 0] > 0, pen.seg_len[:, 2] > 0
    q1, q2 = np.full(B, -1, np.int64), np.full(B, -1, np.int64)
    q1[has1] = nid + np.arange(has1.sum())
    nid += int(has1.sum())
    q2[has2] = nid + np.arange(has2.sum())
    a, b = lat.beam_conn[:, 0].astype(np.int64), lat.beam_conn[:, 1].astype(np.int64)
    s, e = np.where(has1, q1, a), np.where(has2, q2, b)
    xyz = np.concatenate([lat.node_xyz, pen.pen_xyz[has1, 0], pen.pen_xyz[has2, 1]])
    conn = np.concatenate([np.c_[a[has1], q1[has1]], np.c_[s, e], np.c_[q2[has2], b[has2]]])
    rad = np.concatenate([1.5 * lat.beam_radius[has1], lat.beam_radius, 1.5 * lat.beam_radius[has2]])
    return xyz, conn, rad


def end_to_end(cells, geom, radius, rtol):
    """What a user of the drop-in call site waits for: LatticeSim(preset) (host lattice build, penalisation, BCs) +
    solve_FEM_FenicsX (pl_create, upload, assembly, solve, reactions, write-back), wall clock, once."""
    from pylatticedso_amd.lattice_sim import LatticeSim
    from pylatticedso_amd.utils_simulation import solve_FEM_FenicsX
    preset = {"geometry": {"cell_size": {"x": 1, "y": 1, "z": 1}, "number_of_cells": {"x": cells, "y": cells, "z": cells},
                           "radii": [radius], "geom_types": [geom]},
              "simulation_parameters": {"enable": True, "material": "VeroClear", "periodicity": False},
              "boundary_conditions": {
                  "Displacement": {"Fixed": {"Surface": ["Xmin"], "DOF": ["X", "Y", "Z", "RX", "RY", "RZ"],
                                             "Value": [0, 0, 0, 0, 0, 0]}},
                  "Force": {"Load": {"Surface": ["Xmax"], "DOF": ["Z"], "Value": [-0.1]}}}}
    t0 = time.perf_counter()
    L = LatticeSim(preset)
    t1 = time.perf_counter()
    xsol, model = solve_FEM_FenicsX(L, rtol=rtol)
    t2 = time.perf_counter()
    L._device.close()
    return {"total_s": t2 - t0, "lattice_sim_s": t1 - t0, "solve_fem_s": t2 - t1,
            "what": "LatticeSim(preset) + solve_FEM_FenicsX(lattice) through the drop-in call site, first call "
                    "(includes pl_create, topology upload, reactions, write-back)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=50, help="cells per edge (per GPU along y)")
    ap.add_argument("--geom", default="Octet")
    ap.add_argument("--radius", type=float, default=0.03)
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--max-iter", type=int, default=100000)
    ap.add_argument("--kernel", type=int, default=0, help="spmv_kernel option of the library (0 = auto)")
    ap.add_argument("--reorder", type=int, default=1)
    ap.add_argument("--lpn", type=int, default=0, help="lanes per node of the gather kernel (0 = library default)")
    ap.add_argument("--tile-nodes", type=int, default=0, help="target nodes per K*p tile (0 = library default)")
    ap.add_argument("--precond", type=int, default=0,
                    help="1 = Jacobi, 2 = Jacobi + rigid-body coarse space (dense), 3 = 2 + tile level, 4 = 3 + "
                         "rank-local dense level; 0 = 3")
    ap.add_argument("--coarse-max-dofs", type=int, default=0,
                    help="upper bound on the dofs of the dense coarse level (0 = library default, 3072)")
    ap.add_argument("--palette", type=int, default=1, help="1 = K*p reads palette ids when the records repeat")
    ap.add_argument("--condense", type=int, default=0,
                    help="exact elimination of an independent node set inside the PCG: 0 = automatic (bipartite node "
                         "graphs such as BCC), 1 = whenever possible, -1 = never (see pylattice_hip.h)")
    ap.add_argument("--tile-modes", type=int, default=0,
                    help="modes per block of the preconditioner's tile level: 0 / 12 = rigid + uniform strains, 6 = rigid")
    ap.add_argument("--coarse-modes", type=int, default=0,
                    help="modes per aggregate of the dense level: 0 / 6 = rigid, 12 = rigid + uniform strains")
    ap.add_argument("--cg-form", type=int, default=0,
                    help="1 = single-reduction PCG (one all-reduce per iteration on several GPUs, three more stored "
                         "vectors); 0 = ordinary form")
    ap.add_argument("--precision", type=int, default=0,
                    help="0 = fp64 (headline), 1 = fp32 inner PCG + fp64 refinement, 2 = fp32 p and K*p only")
    ap.add_argument("--cpu-cells", type=int, default=36,
                    help="edge of the CPU-baseline sample (0 = skip); 36 = 1.1 M struts, ~10-15 s on one core")
    ap.add_argument("--splu-cells", type=int, default=5,
                    help="edge of the reference-faithful (sub-meshed + sparse LU) CPU sample (0 = skip); 5 = ~15 s")
    ap.add_argument("--no-streaming", action="store_true", help="skip the palette-off / graded-lattice K*p measurement")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end timing through solve_FEM_FenicsX")
    ap.add_argument("--no-bsr", action="store_true", help="leave the explicit BSR assembly out of the step")
    ap.add_argument("--interface-allreduce", action="store_true",
                    help="multi-GPU: sum the interface rows with one all-reduce over all planes instead of the "
                         "neighbour exchange")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (slab build, RCCL communicator) even with one rank (rehearsal)")
    args = ap.parse_args()

    # Libraries (RCCL prints a version banner) write to the C-level stdout: keep fd 1 for the ONE JSON line only.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_dist
    if args.precond == 0:
        # 4 (rank-local dense level) was measured on the single-GPU rehearsal of the multi-rank path: +11 us per
        # iteration and +2 ms per assembly for 12-14 % fewer iterations when the global level is coarse - a wash, so 3
        args.precond = 3
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # host-side bootstrap only (unique id, counts, barrier, max of the timings): gloo.  The data path's RCCL
        # communicator lives inside libpylattice_hip (pl_dist_init) - ONE communicator and one librccl per process
        # (torch's bundled librccl.so and /opt/rocm/lib/librccl.so share the soname librccl.so.1, so the loader maps
        # whichever came first exactly once).
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from pylatticedso_amd import _capi, lattice_arrays as LA, partition as PT
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()

    n = args.cells
    ncell = (n, n * world, n)
    t0 = time.perf_counter()
    if not multi:
        lat = LA.generate((1, 1, 1), ncell, [args.geom], [args.radius])
        pen = LA.penalize(lat, LA.compute_lzone(lat))
        xyz, conn, rad, seg_len, seg_nsub = lat.node_xyz, lat.beam_conn, lat.beam_radius, pen.seg_len, pen.seg_nsub
        n_tgt = None
    else:
        slab = PT.build_slab((1, 1, 1), ncell, [args.geom], [args.radius], rank, world, axis=1)
        xyz, conn, rad, seg_len, seg_nsub = slab.node_xyz, slab.beam_conn, slab.beam_radius, slab.seg_len, slab.seg_nsub
    log(f"[rank {rank}] host lattice build {time.perf_counter() - t0:.1f} s: {len(conn)} struts, {len(xyz)} nodes")

    grid = None
    if multi:
        # every rank must cut the same brick / aggregate grid: hand over the box and node count of the whole lattice
        nn = torch.tensor([float(len(xyz))], dtype=torch.float64)
        dist.all_reduce(nn)
        grid = ((0.0, 0.0, 0.0), tuple(float(v) for v in ncell), int(nn.item()))
    dev = _capi.HipLattice(xyz, conn, rad, seg_len, seg_nsub, E, NU, device=local_rank, spmv_kernel=args.kernel,
                           reorder=args.reorder, lanes_per_node=args.lpn, precond=args.precond, grid=grid, palette=args.palette,
                           tile_nodes=args.tile_nodes, coarse_max_dofs=args.coarse_max_dofs, precision=args.precision,
                           condense=args.condense, cg_form=args.cg_form, tile_modes=args.tile_modes,
                           coarse_modes=args.coarse_modes)
    n_beams_total = len(conn)
    if multi:
        keys = [None] * world
        dist.all_gather_object(keys, slab.iface_key)
        ok, gid, nsg = PT.global_interface_ids(keys, rank)
        uid = [_capi.HipLattice.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        # interface rows: neighbour exchange (grouped ncclSend/ncclRecv with the two adjacent slabs) unless the
        # all-planes all-reduce is asked for; a single-rank rehearsal has no neighbours
        peers = None if (args.interface_allreduce or world == 1) else slab.iface_peer[ok]
        dev.dist_init(rank, world, uid[0], slab.iface_local[ok], gid, nsg, shared_peer=peers)
        # global number of loaded nodes / struts (shared nodes counted once: they belong to the lower slab)
        lower_plane = np.zeros(len(xyz), bool)
        if rank > 0:
            lower_plane[slab.iface_local[ok][slab.iface_key[ok][:, 0] == slab.layers[0]]] = True
        cnt = torch.tensor([float(((xyz[:, 0] == float(n)) & ~lower_plane).sum()), float(len(conn))],
                           dtype=torch.float64)
        dist.all_reduce(cnt)
        n_tgt, n_beams_total = int(cnt[0].item()), int(cnt[1].item())
    fixed, f, _ = cantilever_bc(xyz, float(n), n_tgt)
    dev.set_bc(fixed, None, f)

    def step():
        dev.assemble()
        if not args.no_bsr:
            dev.assemble_bsr(False)
        return dev.solve(rtol=args.rtol, max_iter=args.max_iter, download=False)

    def sync():
        torch.cuda.synchronize()     # device-wide: covers the library's own streams
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        st = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = step()
    sync()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel: K*p.  HIP events on the library's stream (torch events only see torch's stream).
    ms_spmv = dev.time_kernel(0, 50)
    ms_iter = dev.time_kernel(3, 50)
    ms_rec = dev.time_kernel(1, 20)
    ms_bsr = dev.time_kernel(2, 10) if not args.no_bsr else None
    ms_f32 = None
    if args.precond >= 2 and args.kernel in (0, 3) and args.reorder:
        ms_f32 = {"spmv_f32_storage": dev.time_kernel(7, 50), "pcg_iteration_precision1": dev.time_kernel(8, 50),
                  "pcg_iteration_precision2": dev.time_kernel(9, 50)}
    # the two collectives of an iteration, alone (every rank makes the same calls; rank 0's clock is reported)
    ms_coll = {"interface_allreduce": dev.time_kernel(5, 50), "coarse_allreduce": dev.time_kernel(6, 50)} if multi else None
    ab = dev.algorithmic_bytes()
    achieved = ab["spmv"] / (ms_spmv * 1e-3) / 1e9
    # HBM bytes per K*p launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs;
    # gfx950: FETCH_SIZE counts half of 16-B/lane streaming reads - see profiles/README.md).  Valid for the default
    # single-GPU workload only.
    traffic = traffic_src = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_spmv_latest.json")
    if world == 1 and n == 50 and args.geom == "Octet" and os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        if pmc.get("spmv_kernel") == dev_kernel_name(args.kernel, args.reorder) and \
                pmc.get("record_palette", 0) == args.palette:
            traffic = (2.0 * pmc["fetch_kb"] + pmc["write_kb"]) * 1024.0
            traffic_src = {"file": "profiles/pmc_spmv_latest.json", "measured_on": pmc.get("build", "unknown build"),
                           "note": "PMC counters cannot be read inside this run (rocprofv3 --pmc is a separate "
                                   "pass); this is the committed pass for this workload / kernel / palette setting"}

    # The same kernel on lattices whose records do NOT repeat (graded / optimised radii: what every pl_update_radii loop
    # runs): K*p then streams one 40-byte record per strut instead of 2-byte palette ids.  Measured on this lattice with
    # the palette switched off, and on a graded copy (own radius per cell, > 10^5 distinct records) incl. a whole step.
    streaming = None
    if world == 1 and not args.no_streaming and args.kernel in (0, 3) and args.reorder:
        streaming = {}
        with _capi.HipLattice(xyz, conn, rad, seg_len, seg_nsub, E, NU, device=local_rank, spmv_kernel=args.kernel,
                              reorder=args.reorder, precond=args.precond, palette=0, tile_nodes=args.tile_nodes,
                              coarse_max_dofs=args.coarse_max_dofs) as d2:
            d2.set_bc(fixed, None, f)
            d2.assemble()
            ms = d2.time_kernel(0, 50)
            streaming["palette_off"] = {"ms": ms, "achieved": ab["spmv"] / (ms * 1e-3) / 1e9,
                                        "frac": ab["spmv"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        i3 = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), axis=-1).reshape(-1, 3)
        rad_cell = (args.radius * (0.8 + 0.4 * (0.5 + 0.5 * np.sin(0.113 * i3[:, 0] + 0.271 * i3[:, 1] + 0.419 * i3[:, 2]))))
        glat = LA.generate((1, 1, 1), ncell, [args.geom], [args.radius], cell_radii_override=rad_cell.reshape(-1, 1))
        gpen = LA.penalize(glat, _capi.lzone(glat.node_xyz, glat.beam_conn, glat.beam_radius))
        with _capi.HipLattice(glat.node_xyz, glat.beam_conn, glat.beam_radius, gpen.seg_len, gpen.seg_nsub, E, NU,
                              device=local_rank, spmv_kernel=args.kernel, reorder=args.reorder, precond=args.precond,
                              palette=args.palette, tile_nodes=args.tile_nodes,
                              coarse_max_dofs=args.coarse_max_dofs) as d3:
            d3.set_bc(fixed, None, f)
            d3.assemble()
            gst = d3.solve(rtol=args.rtol, max_iter=args.max_iter, download=False)
            torch.cuda.synchronize()
            t0g = time.perf_counter()
            for _ in range(3):
                d3.assemble()
                gst = d3.solve(rtol=args.rtol, max_iter=args.max_iter, download=False)
            torch.cuda.synchronize()
            dtg = (time.perf_counter() - t0g) / 3
            ms = d3.time_kernel(0, 50)
            streaming["graded"] = {"ms": ms, "achieved": ab["spmv"] / (ms * 1e-3) / 1e9,
                                   "frac": ab["spmv"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "distinct_radii": int(len(np.unique(glat.beam_radius))),
                                   "beams_per_s": glat.n_beams / dtg, "ms_per_step": dtg * 1e3,
                                   "pcg_iterations": gst["iterations"],
                                   "step": "records + palette attempt + Jacobi diag + coarse levels + PCG (no BSR)"}
        streaming.update({"bound": "hbm", "kernel": "K*p: k_spmv_tile<.., kRecCompact>", "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "algorithmic_bytes": ab["spmv"],
                          "traffic": (2.0 * 126177.28125 + 24322.515625) * 1024.0,
                          "traffic_source": "profiles/r02_cj_pmc_streaming_p0.json / _gr.json (separate rocprofv3 --pmc FETCH_SIZE and "
                                            "WRITE_SIZE passes of tools/prof_stream.sh: 2 x 129.2 MB fetched + 24.9 MB "
                                            "written per launch, 1.06 x the algorithmic bytes)"})

    out = {
        "metric": "beams/s assembly+PCG-solve", "value": n_beams_total * args.steps / dt, "unit": "beams/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {0: "f64", 1: "f32 storage + f64 refinement (f64 arithmetic)", 2: "f32 p/Kp storage, f64 x/r"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"{ncell[0]}x{ncell[1]}x{ncell[2]} {args.geom} r={args.radius} cantilever "
                               f"(BASELINE.json configs[1] per GPU)",
                   "struts": n_beams_total, "struts_per_gpu": len(conn), "nodes_per_gpu": len(xyz),
                   "partition": "single GPU" if world == 1 else
                   f"{world} y-slabs, RCCL " + ("interface all-reduce" if args.interface_allreduce else
                                                "neighbour exchange of interface rows (ncclSend/ncclRecv)") +
                   " + fused scalar all-reduces",
                   "rtol": args.rtol, "pcg_iterations": st["iterations"], "converged": st["converged"],
                   "rel_residual": st["rel_residual"], "precision": args.precision,
                   "inner_solves": st.get("restarts", 0.0), "condensed_nodes": int(st.get("condensed_nodes", 0)),
                   "cg_form": int(st.get("cg_form_used", 0)),
                   "preconditioner": {1: "Jacobi", 2: "two-level (Jacobi + rigid-body coarse space)",
                                      3: "multi-level (Jacobi + tile blocks + dense rigid-body coarse space)",
                                      4: "multi-level (Jacobi + tile blocks + rank-local dense level + all-reduced "
                                         "dense rigid-body coarse space)"}[args.precond],
                   "step": "records + Jacobi diag" + (" + coarse operator/factorisation" if args.precond >= 2 else "")
                           + ("" if args.no_bsr else " + BSR(6x6) K") + " + matrix-free PCG",
                   "spmv_kernel": args.kernel, "reorder": args.reorder, "record_palette": args.palette},
        "roofline": {"bound": "hbm", "kernel": "K*p: " + dev_kernel_name(args.kernel, args.reorder),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": ab["spmv"], "ms": ms_spmv,
                     "note": "tiles are the bricks of the preconditioner's tile level, nested into the aggregates of its "
                             "dense level (DESIGN.md section 7): 15^3 tiles of 152 nodes at 50^3 Octet cost K*p 3 us "
                             "against the 13^3 tiles of round 1 (0.93 -> 0.86; 0.88 since its index prefetch moved behind the gathers) and, with the strain modes of both block "
                             "levels, save 35 of 155 PCG iterations"},
        "kernels_ms": {"spmv": ms_spmv, "pcg_iteration": ms_iter, "record_build": ms_rec, "bsr_fill": ms_bsr,
                       "pcg_iter_GBps": ab["pcg_iter"] / (ms_iter * 1e-3) / 1e9,
                       "solve_ms_last": st["ms_solve"], "assembly_ms_last": st["ms_assembly"], "fp32_modes": ms_f32},
    }
    if streaming is not None:
        out["roofline_streaming"] = streaming
    if ms_coll is not None:
        out["collectives_ms"] = ms_coll
        # one RCCL per process: torch's bundled librccl.so and /opt/rocm's share the soname, the loader maps the first
        with open("/proc/self/maps") as fh:
            out["rccl_libraries_mapped"] = sorted({ln.split()[-1] for ln in fh if "librccl" in ln})
        out["process_group_backend"] = "gloo (host bootstrap only); data path: RCCL communicator inside libpylattice_hip"
    dev.close()
    if rank == 0 and world == 1 and args.cpu_cells > 0:
        cb, _ = cpu_baseline(args.cpu_cells, args.radius, args.rtol, args.splu_cells)
        out["cpu_baseline"] = cb
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_e2e and not args.force_dist:
        out["end_to_end_s"] = end_to_end(n, args.geom, args.radius, args.rtol)
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
