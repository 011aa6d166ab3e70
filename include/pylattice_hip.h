/*
 * pylattice_hip.h — C ABI of libpylattice_hip.so (MI355X / gfx950).
 *
 * The reference (Tcadart/pyLatticeDSO, 100 % Python) has no FFI of its own; the seam this library sits behind is
 * the dolfinx/PETSc work done inside
 *     solve_FEM_FenicsX            src/pyLatticeSim/utils_simulation.py:21-56
 *     SimulationBase.solve_problem src/pyLatticeSim/simulation_base.py:465-514   (assemble K, lifting, LU solve)
 *     get_schur_complement         src/pyLatticeSim/utils_schur.py:22-53
 *     LatticeOpti.calculate_gradient src/pyLatticeOpti/lattice_opti.py:735-907   (u^T dK/dr u, adjoint form)
 * Each entry point below cites the reference code it replaces.  All pointers are caller-owned HOST buffers
 * (C-contiguous) unless the name ends in _dev; device state is owned by the handle.  Every function returns
 * 0 on success or a negative pl_status; pl_last_error() gives the message.  No C++ exceptions cross the ABI.
 * Handles are not thread-safe; distinct handles are independent.
 *
 * Unknown ordering everywhere: node-major, 6 values per node [ux, uy, uz, thx, thy, thz]
 * (mixed P1xP1 space of simulation_base.py:201-213).
 */
#ifndef PYLATTICE_HIP_H
#define PYLATTICE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pl_context *pl_handle;

typedef enum {
  PL_OK = 0,
  PL_ERR_ARG = -1,       /* bad argument (null pointer, negative size, index out of range) */
  PL_ERR_HIP = -2,       /* HIP runtime error (message has the hipError string) */
  PL_ERR_STATE = -3,     /* call order violated (e.g. solve before set_bc) */
  PL_ERR_NOCONV = -4,    /* PCG hit max iterations (solution and stats are still written) */
  PL_ERR_NAN = -5,       /* NaN/Inf detected in the residual */
  PL_ERR_NODEVICE = -6   /* no HIP device visible */
} pl_status;

/* The lattice the FEM mesh is built from.  One entry per DESIGN strut (before joint penalisation); the
 * penalised end segments (LatticeSim.set_penalized_beams, lattice_sim.py:245-308) and the gmsh sub-mesh
 * (latticeGeneration.mesh_lattice_cells, lattice_generation.py:64-101) are described per strut by seg_len/seg_nsub
 * and condensed in closed form on the device. */
typedef struct {
  int64_t n_nodes;
  int64_t n_beams;
  const double *node_xyz;     /* [3*n_nodes] */
  const int32_t *beam_conn;   /* [2*n_beams] point1, point2 */
  const double *beam_radius;  /* [n_beams]   un-penalised radius r (end segments use pen_coef*r) */
  const double *seg_len;      /* [3*n_beams] geometric length of [pen@point1, middle, pen@point2], 0 = absent */
  const int32_t *seg_nsub;    /* [3*n_beams] number of equal P1 sub-elements per segment (>=1 where len>0) */
} pl_mesh_t;

/* ABI handshake.  pl_opts_t and pl_stats_t grow from release to release; a caller compiled (or a ctypes binding written)
 * against another header must fail with PL_ERR_ARG instead of having memory overrun.  Both structs therefore START with
 * the size the CALLER believes they have: pl_default_opts(&o, sizeof o) refuses any other size than the library's and
 * stamps struct_size / abi_version, pl_create / pl_create_ddm check the stamp, and pl_solve checks stats->struct_size,
 * which the caller sets (= sizeof(pl_stats_t)) before the call.  pl_opts_size() / pl_stats_size() / pl_abi_version()
 * let a binding assert its layout when it loads the library. */
#define PL_ABI_VERSION 6u

typedef struct {
  uint32_t struct_size;  /* sizeof(pl_opts_t) as the caller sees it; written by pl_default_opts, checked by pl_create */
  uint32_t abi_version;  /* PL_ABI_VERSION, written by pl_default_opts */
  double young;          /* E   (materials/<name>.json Young_modulus) */
  double poisson;        /* nu */
  double kappa;          /* shear correction, 0.9 in material_definition.py:45 */
  double pen_coef;       /* radius multiplier of penalised segments, 1.5 (beam.py:71) */
  int32_t device;        /* HIP device ordinal */
  int32_t spmv_kernel;   /* 0 = auto (3 if reorder else 2), 1 = per-strut + f64 global atomics, 2 = per-node gather
                            (sliced ELL), 3 = per-strut with LDS tile accumulators (in its LDS-resident form - a tile's rows
                            of x and the record / direction palette in LDS, one 32-bit word per strut visit - whenever the
                            lattice has <= 256 distinct records or strut vectors; environment PL_TILE_LDS=0 keeps the form
                            that gathers from global memory, for A/B runs) */
  int32_t precond;       /* 1 = Jacobi (diagonal); 2 = two-level: Jacobi + rigid-body-mode coarse space on brick
                            aggregates, dense solve; 3 = 2 plus a tile level (rigid-body modes of every K*p tile,
                            6 x 6 block solves); 4 = 3 plus a rank-LOCAL dense level (aggregates of this handle only,
                            nodes shared with other ranks left out, no communication) under the global one: for
                            multi-GPU runs, where the all-reduced global level has to coarsen with the rank count.
                            2, 3 and 4 need reorder = 1;
                            5 = the dense Cholesky factor of P K P + (I - P) itself (6 n_nodes <= PL_DDM_DENSE_MAX; built by
                            pl_assemble from the BSR blocks, fp64 inverse factor): the PCG converges in one or two steps - what
                            PETSc's preonly / LU is to the reference, for lattices of a few hundred nodes (its own presets) where
                            hundreds of PCG iterations cost more than a factorisation; pl_stats_t.precond_used = 5, or 1 (Jacobi)
                            when the matrix is not positive definite (a mechanism).  Single-GPU handles */
  int32_t reorder;       /* 0 = keep caller's node numbering on the device, 1 = spatial tile reordering */
  int32_t check_every;   /* PCG: iterations between host-side convergence checks; 0 = adaptive (32 while far from
                            the threshold, then what the observed decay rate predicts is still needed) */
  int32_t lanes_per_node;/* gather kernels: wave lanes sharing one node, 1/2/4/8/16 (0 -> 4) */
  int32_t tile_nodes;    /* target nodes per brick/tile of the spatial reordering, <= 512 (0 -> 256) */
  int32_t coarse_max_dofs; /* precond = 2/3: upper bound on 6 * (number of aggregates) (0 -> 2100 below 10^6 nodes
                              on one GPU, 3072 otherwise) */
  int32_t palette;       /* 1: K*p (LDS-tile kernel) reads 2-byte palette ids instead of 64-byte records when the lattice
                            has <= ~30 000 distinct records (compared on 40 mantissa bits, i.e. to 1e-12) */
  int32_t local_max_dofs;  /* precond = 4: upper bound on 6 * (aggregates of the rank-local level) (0 -> 3072) */
  int32_t precision;     /* storage precision of the PCG vectors (multi-level PCG on the LDS-tile kernel, i.e. precond >=
                            2 with reorder = 1; any other configuration runs fp64).  0 = fp64.  1 = fp32 inner PCG with
                            fp64 refinement: x, r, p, K*p stored in fp32, restart from the TRUE fp64 residual
                            P(f - K u) every ~4 decades.  2 = only p and K*p stored in fp32, x and the residual
                            recurrence in fp64, true residual verified at the end.  All products and sums are evaluated
                            in fp64 in every mode; rtol always refers to the true fp64 residual in modes 1 and 2 */
  int32_t restart_every; /* > 0: every restart_every-th iteration rebuilds the search direction as the reference's CG does
                            (conjugate_gradient_solver.py:96-97; solve_DDM passes 500000).  Jacobi / DDM paths */
  double alpha_max;      /* > 0: clamp the CG step like conjugate_gradient_solver.py:79 (DDM solves use 100) */
  /* Multi-GPU only: bounding box and node count of the WHOLE lattice, so that every rank cuts the same brick /
   * aggregate grid (all zero -> derived from this handle's own nodes). */
  double grid_lo[3];
  double grid_hi[3];
  int64_t grid_nodes;
  double mintol;         /* > 0: also stop (converged, pl_stats_t.stop_reason = 1) when ||p|| < mintol (||x|| + 1e-12), the
                            reference's "direction norm" test (conjugate_gradient_solver.py:102-105; solve_DDM passes
                            1e-12), and report info = 2 when a step length fell below 1e-6 (:107-109).  These tests need
                            the host to see every iteration: use check_every = 1.  Jacobi / DDM paths */
  int32_t compact_records; /* LDS-tile K*p without a palette (graded / optimised lattices): 0 or 1 = stream 40-byte records
                              (the 5 stiffness scalars; the strut vector is recomputed from the node coordinates),
                              -1 = stream the 64-byte records */
  int32_t condense;      /* multi-level PCG (precond >= 2; fp64 and precision = 1; one or several GPUs - nodes shared with another
                            rank are never eliminated, and either every rank eliminates or none does): exact elimination of an independent set of nodes (no
                            two share a strut, none carries a Dirichlet dof; chosen at pl_create, interior nodes first)
                            inside the solver - CG runs on the Schur complement of the other nodes, the vector kernels skip
                            the eliminated rows, every iteration pays a second K*p, x still receives every node.
                            0 = automatic: when >= 45 % of the nodes can go (bipartite node graphs such as BCC: 793 -> 503
                            iterations and 275 -> 228 ms at 100^3), 1 = whenever candidates exist, -1 = never */
  int32_t chol_persistent; /* 1: factor the dense coarse operator in ONE persistent launch with a grid barrier per block
                              column instead of one launch per block column.  Measured slower (2.94 vs 2.86 ms per
                              assembly at 50^3 Octet): the release / acquire fences of a barrier across the 8 XCDs cost
                              what the kernel boundary costs - kept as an experiment switch */
  int32_t cg_form;       /* multi-level PCG (precond 2 / 3, fp64, no node elimination): 1 = single-reduction form (Chronopoulos-
                          * Gear recurrences, the dense level's residual carried by recurrence): ONE all-reduce per iteration on
                          * several GPUs instead of three exchanges, for three more stored vectors.  0 = ordinary form. */
  int32_t tile_modes;    /* tile level of the multi-level PCG: 0 / 12 = rigid-body + uniform-strain modes per tile (12 x 12 blocks;
                          * both CG forms, one or several GPUs), 6 = rigid-body modes only */
  int32_t coarse_modes;  /* dense level of the multi-level PCG: 0 = automatic (12 from 250 k nodes of the whole lattice), 12 =
                          * rigid-body + uniform-strain modes per aggregate (needs tile_modes = 12; fewer, larger aggregates),
                          * 6 = rigid-body modes */
  int32_t overlap;       /* multi-GPU handles with the neighbour exchange (pl_dist_set_peers), LDS-tile K*p: 0 / 1 = the tiles
                          * that own interface rows run first and their rows travel (pack, send / recv, add) on a second
                          * stream while the interior tiles run, -1 = one launch, then the exchange */
  int32_t coarse_storage; /* storage of the dense level's inverse factor W (the two triangular GEMVs of every iteration read it):
                           * 32 = fp32, 16 = bfloat16 (half the bytes; W16^T W16 is still symmetric positive definite and
                           * fixed - measured iteration counts unchanged), 0 = automatic: bfloat16 from 1 024 dofs on (round 4; from
                           * 3 072 until then: 100^3 BCC 2 x 22 us of a 354-us iteration; at 1 536 dofs the GEMVs are
                           * latency-bound and the step gains 1 - 2 %, iteration counts of five lattice types unchanged) */
  int32_t warm_start;    /* design loops (LatticeOpti.objective / gradient call the solver over and over on a slowly changing
                          * lattice, lattice_opti.py:569-570): 1 = pl_solve starts from the previous CONVERGED solution of this
                          * handle instead of from zero (one extra K*x; the stopping test is unchanged: ||r|| <= rtol ||b|| of the
                          * current right-hand side).  Multi-level PCG in fp64, ordinary form, single-GPU handles; ignored
                          * elsewhere.  0 = every solve starts from zero (what bench.py times on configs[1] / [2] / [4]: a loop of
                          * IDENTICAL solves must not start from its own answer).  2 (round 5) = start from 2 x_prev - x_prev2,
                          * the linear extrapolation of the handle's last two solutions - a design loop moves along a smooth path
                          * (configs[3]: 249 -> 208 iterations per solve on average); 3 = quadratic extrapolation of the last three
                          * (193; amplifies a jagged path threefold - opt-in); 4 = the GALERKIN start: x0 = the combination of the
                          * handle's last six solutions (environment PL_WARM_VECTORS, 2 ... 8) that is nearest to the solution of the
                          * CURRENT system in its energy norm, (V^T K V) c = V^T b - one operator application and a handful of dot
                          * products per stored vector, one 6 x 6 solve on the host.  A projection: it contains the candidates of 1, 2
                          * and 3 and can do no worse than any of them (configs[3]: 160 iterations per solve).  With the fp32 inner
                          * solver (precision = 1) 2, 3 and 4 act as 1. */
  int32_t short_iteration; /* small lattices (few K*p tiles: the dense level's explicit inverse can be read once per tile and
                          * iteration): 1 = the SHORT form of the multi-level PCG iteration (pl_small.h) - the dense level's
                          * solve and the prolongation fused into one launch that writes z = M^-1 r and r.z, the search
                          * direction p = z + beta p formed inside the next K*p launch: 3 dependent launches per iteration
                          * instead of 5 (4 instead of 6 under node elimination); 0 = automatic (on where it applies: fp64,
                          * ordinary CG form, single-GPU handle, n_tiles x modes x dense dofs small), -1 = never;
                          * 2 = EXPERIMENTAL: the whole loop as ONE persistent launch (pl_persist.h: single-reduction CG without
                          * node elimination, one workgroup per tile, hand-offs through write-through stores and flags; at most
                          * 256 tiles of <= 512 nodes, dense level <= 2 048 dofs) - measured slower than 1, kept for
                          * the record; pl_stats_t.short_iteration_used = 2.
                          * pl_stats_t.short_iteration_used says what ran */
} pl_opts_t;

typedef struct {
  uint32_t struct_size;    /* IN: sizeof(pl_stats_t) as the caller sees it (pl_solve returns PL_ERR_ARG on any other value) */
  int32_t iterations;
  int32_t converged;       /* 1 if ||r|| <= rtol*||b|| */
  int32_t reserved_i;
  double rel_residual;     /* ||r||/||b|| of the recurrence at exit */
  double b_norm;
  double ms_assembly;      /* last pl_assemble (+ pl_assemble_bsr) on the device, HIP events */
  double ms_solve;         /* last pl_solve, HIP events around the PCG loop */
  double ms_spmv_avg;      /* average K*x kernel time inside the last pl_solve (HIP events, sampled) */
  double precond_used;     /* preconditioner the solve actually ran with (opts->precond numbering): differs from the
                              request when a dense level was not positive definite and the solve fell back to Jacobi */
  double restarts;         /* precision = 1 / 2: inner solves taken (each ends with a true-residual evaluation) */
  double precision_used;   /* precision mode the solve ran in (0 when the request did not apply) */
  double info;             /* the reference CG's return code (conjugate_gradient_solver.py:75,99-109): 0 converged,
                              1 not converged, 2 not converged and a step length fell below 1e-6 */
  double stop_reason;      /* which test ended a converged solve: 0 ||r|| <= rtol ||b||, 1 direction norm (mintol) */
  double condensed_nodes;  /* nodes eliminated exactly inside the solve (opts.condense) */
  double cg_form_used;     /* 1: the solve ran in the single-reduction form (opts.cg_form) */
  double kp_form;          /* which K*p the solve applied: 1 = LDS-resident tile kernel with the record palette (k_spmv_tile_lds),
                              2 = LDS-resident tile kernel streaming 40-byte records (k_spmv_tile_lds_t), 3 = tile kernel gathering
                              from global memory (k_spmv_tile), 4 = row form (k_spmv_rows), 5 = per-node gather, 6 = global
                              atomics, 7 = DDM cell product */
  double comm_world;       /* multi-GPU handles: ranks of the communicator as the COMMUNICATOR reports them (ncclCommCount, or
                              the loopback group's size) - not what the launcher's environment says; 0 on a single-GPU handle */
  double comm_rank;        /* this handle's rank in it (ncclCommUserRank) */
  double short_iteration_used;  /* 1: the solve ran the short form of the iteration (opts.short_iteration) */
  double reserved[2];
} pl_stats_t;

/* Fills *o with the defaults.  struct_size = sizeof(pl_opts_t) of the caller's header; PL_ERR_ARG (and *o untouched)
 * when it differs from the library's. */
int pl_default_opts(pl_opts_t *o, uint32_t struct_size);
uint32_t pl_opts_size(void);
uint32_t pl_stats_size(void);
uint32_t pl_abi_version(void);
const char *pl_last_error(void);
const char *pl_version(void);

/* Joint-penalisation length L_zone at both ends of every strut of a NON-periodic lattice, lzone[2*b + end] (end 0 =
 * beam_conn[2b]): over the other struts meeting at that node, the largest r_other / tan(angle / 2); 1e-7 where the
 * angle exceeds 170 degrees, 0 where no other strut meets the node.  Replaces the per-node double loop of
 * Lattice.define_angles_between_beams (lattice.py:871-904) with Beam.get_angle_between_beams (beam.py:204-277) and
 * function_penalization_Lzone (utils.py:432-453).  Stand-alone (no handle): the result decides seg_len / seg_nsub of
 * pl_mesh_t.  Periodic single cells (Schur datasets) keep the host restatement. */
int pl_lzone(int device, int64_t n_nodes, int64_t n_beams, const double *node_xyz, const int32_t *beam_conn,
             const double *beam_radius, double *lzone);

/* Upload topology + geometry; builds the node->strut incidence.  Replaces BeamModel.__init__ /
 * latticeGeneration (beam_model.py:57-105, lattice_generation.py:64-175). */
int pl_create(const pl_mesh_t *mesh, const pl_opts_t *opts, pl_handle *out);
void pl_destroy(pl_handle h);

/* Domain-decomposition operator (LatticeSim.solve_DDM, lattice_sim.py:1111-1252): unknowns are the n_nodes
 * cell-boundary nodes; every cell c couples its nb boundary nodes cell_nodes[c*nb..] (order of
 * Cell.define_node_order_to_simulate, cell.py:611-680) through the dense Schur complement S[cell_S[c]] ((6nb)^2,
 * row-major).  The handle then serves pl_set_bc / pl_assemble / pl_spmv / pl_spmv_free / pl_solve / pl_reactions with
 * K := sum_c B_c^T S_c B_c; pl_solve runs plain CG (opts->precond = 0, as the reference's default), Jacobi-CG
 * (opts->precond = 1), CG preconditioned by the factorised assembled matrix (opts->precond = 2, below), by the inverted
 * 6 x 6 node blocks of the assembled matrix (opts->precond = 3: any size; built by pl_assemble for the current Dirichlet
 * mask from the same matrices as precond = 2) or by the node blocks plus a dense level on aggregates of nodes
 * (opts->precond = 4, pl_ddm_set_geometry below), with opts->alpha_max. */
int pl_create_ddm(int64_t n_nodes, int64_t n_cells, int32_t nb, const int32_t *cell_nodes, int32_t n_S, const double *S,
                  const int32_t *cell_S, const pl_opts_t *opts, pl_handle *out);

/* The reference's preconditioner of the DDM solve (LatticeSim.define_preconditioner / build_preconditioner,
 * lattice_sim.py:1333-1415; Cell.build_coupling_operator / build_local_preconditioner, cell.py:754-827):
 * G = sum_c B_c^T Shat_c B_c on the free dofs, factorised once (SuperLU there; dense Cholesky + explicit inverse
 * factor on the device here, so 6 n_nodes <= PL_DDM_DENSE_MAX), z = G^-1 r per iteration.  Selected by
 * opts->precond = 2 at pl_create_ddm; pl_assemble builds and factorises G for the current Dirichlet mask.
 * Shat_c defaults to the operator's own matrices (preconditioner_type "exact": CG converges in one step); this call
 * installs another palette - one mean matrix ("mean"), or the dataset matrices with the nearest-radius index per
 * cell ("nearest_reference").  pl_set_bc on an assembled precond = 2 handle invalidates the factorisation (it
 * depends on the Dirichlet mask): call pl_assemble again before pl_solve.  S = NULL goes back to the default.  When G is not positive definite (an indefinite
 * surrogate matrix) pl_assemble falls back to Jacobi (the reference: LU -> ILU) and pl_stats_t.precond_used says 1. */
#define PL_DDM_DENSE_MAX 16384
int pl_ddm_set_preconditioner(pl_handle h, int32_t n_S, const double *S /*[n_S][6nb][6nb]*/,
                              const int32_t *cell_S /*[n_cells]*/);

/* New cell matrices on an existing DDM handle (round 5): a design loop changes every S_c between two solves
 * (LatticeOpti.set_optimization_parameters -> reset_cell_with_new_radii, lattice_opti.py:467-560, lattice_sim.py:1421-1497) while
 * the cells keep their nodes.  S / cell_S as in pl_create_ddm (same nb; n_S may differ).  The handle's preconditioner data is
 * dropped: pl_assemble before the next pl_solve.  A palette installed by pl_ddm_set_preconditioner stays. */
int pl_ddm_update_matrices(pl_handle h, int32_t n_S, const double *S /*[n_S][6nb][6nb]*/, const int32_t *cell_S /*[n_cells]*/);

/* Beyond PL_DDM_DENSE_MAX dofs (round 5): opts->precond = 4 on a DDM handle = the node blocks of precond = 3 plus a dense
 * level, M^-1 = B^-1 + Z A_c^-1 Z^T with A_c = Z^T P G P Z - twelve modes (six rigid-body motions, six uniform strains) per
 * aggregate of boundary nodes, aggregates = boxes of a regular grid over the nodes' bounding box, as many as
 * opts->coarse_max_dofs / 12 allows (0: 1 536 dofs).  It stands where the reference factorises the assembled matrix
 * (build_preconditioner, lattice_sim.py:1351-1415): not a direct solve, but an iteration count that no longer grows with the
 * lattice (32^3 BCC cells: 2.2 x fewer iterations than the node blocks).  The modes need the node positions, which
 * pl_create_ddm does not take: this call hands them over (before pl_assemble; calling it again re-cuts the aggregates).
 * pl_assemble builds the node blocks, A_c from the cell matrices (the palette of pl_ddm_set_preconditioner if one is set)
 * and its Cholesky / inverse factor; when A_c is not positive definite the node blocks alone are used and
 * pl_stats_t.precond_used says 3. */
int pl_ddm_set_geometry(pl_handle h, const double *node_xyz /*[3 n_nodes]*/);

/* Dirichlet / load data per dof.  fixed[6N] (0/1), ubar[6N] prescribed values (read where fixed), f[6N] nodal
 * loads.  Replaces apply_displacement_all_nodes_with_lattice_data / apply_force_on_all_nodes_with_lattice_data
 * (full_scale_lattice_simulation.py:39-73,124-153).  Any of ubar/f may be NULL (= zeros). */
int pl_set_bc(pl_handle h, const uint8_t *fixed, const double *ubar, const double *f);

/* Periodic constraints for one-cell homogenisation (HomogenizedCell.periodic_boundary_condition,
 * homogenization_cell.py:210-252: dolfinx_mpc ties all six dofs of opposite corner / edge / face nodes): master[i] = the node
 * whose dofs node i shares (master[master[i]] == master[i]; i itself for an unconstrained node).  pl_solve then solves
 * P^T K P v = P^T f, u = P v, as CG on Q K Q with Q the orthogonal projector "average over each group" - f must be periodic
 * (the caller averages it over the groups) and the Dirichlet flags equal on all members of a group.  Jacobi PCG of a
 * single-GPU handle (opts.precond = 1).  master = NULL: no constraints. */
int pl_set_periodic(pl_handle h, const int32_t *master /*[n_nodes] or NULL*/);

/* New radii, same topology/segment geometry (optimisation loop; Cell.change_beam_radius cell.py:896-917). */
int pl_update_radii(pl_handle h, const double *beam_radius);
/* Per-strut multiplicity: strut b stands for beam_mult[b] identical struts in parallel between its two nodes (its record,
 * sensitivity and energy scale with it; the back-substitution of pl_node_mod gives every copy its share of the section
 * force).  This is what the reference's own model is on lattices whose struts lie in cell faces or on cell edges (Octet,
 * Cubic, Kelvin ...): LatticeSim.set_penalized_beams splits a strut shared by k cells once PER CELL and keeps every copy
 * (lattice_sim.py:250-303; Beam hashes by identity, beam.py:78-82; the mesher de-duplicates by object,
 * lattice_generation.py:152-160), and Lattice.check_hybrid_collision does the same to struts cut by another geometry's
 * node (lattice.py:1111-1215).  NULL = 1 everywhere (the default).  The next pl_assemble picks it up. */
int pl_set_multiplicity(pl_handle h, const double *beam_mult /*[n_beams] > 0, or NULL*/);
/* New penalised-segment geometry (when the caller re-runs the angle search, lattice_sim.py:1421-1497). */
int pl_update_segments(pl_handle h, const double *seg_len, const int32_t *seg_nsub);

/* Per-strut stiffness build ("assembly" of the matrix-free operator): condensed element records + the
 * preconditioner.  Replaces Material.compute_mechanical_properties + the FFCx element kernel
 * (material_definition.py:142-156, simulation_base.py:220-225).
 * Once pl_assemble_bsr has been called on the handle, pl_assemble also refreshes that explicit matrix (same with_bc):
 * the fill runs on a second stream next to the factorisation of the coarse operator.  Collective on a multi-GPU
 * handle (pl_dist_init): every rank must call it. */
int pl_assemble(pl_handle h);

/* Explicit global K as BSR(6x6) on the device (dolfinx assemble_matrix, simulation_base.py:473-476).
 * with_bc != 0 applies dolfinx's Dirichlet treatment (constrained rows/cols zeroed, unit diagonal). */
int pl_assemble_bsr(pl_handle h, int with_bc, int64_t *n_block_rows, int64_t *n_blocks);
int pl_get_bsr(pl_handle h, int64_t *rowptr /*[N+1]*/, int32_t *colidx /*[nblk]*/, double *vals /*[36*nblk]*/);

/* y = K x with the full (unconstrained) operator; test hook and building block of reactions. */
int pl_spmv(pl_handle h, const double *x, double *y);
/* y = P K P x with P the projector on free dofs (the PCG operator). */
int pl_spmv_free(pl_handle h, const double *x, double *y);
/* y = BSR * x using the explicitly assembled matrix (cross-check of the two paths). */
int pl_spmv_bsr(pl_handle h, const double *x, double *y);

/* Solve K u = f with the Dirichlet data of pl_set_bc by Jacobi-PCG on the matrix-free operator.
 * Replaces the PETSc KSP(preonly)+PC(LU) solve of simulation_base.py:501-511.  u[6N] gets the full field
 * (prescribed values on constrained dofs; u == NULL keeps it on the device only).  rtol is on ||r||/||b||. */
int pl_solve(pl_handle h, double rtol, int32_t max_iter, double *u, pl_stats_t *stats);

/* R = K u on every dof (caller keeps the constrained ones).  Replaces
 * calculate_reaction_force_and_moment_at_position (simulation_base.py:582-645). */
int pl_reactions(pl_handle h, const double *u, double *R);

/* Per-strut sensitivity s_b = lam_e^T (dK_e/dr_b) u_e at fixed segment geometry (lam == NULL -> lam = u; u == NULL -> the
 * solution of the last pl_solve of this handle, which is still on the device: a design loop need not upload it again).
 * Replaces the dS/dr contraction of LatticeOpti.calculate_gradient (lattice_opti.py:746-902) and the dormant
 * Material.compute_gradient (material_definition.py:163-231). */
int pl_sens(pl_handle h, const double *u, const double *lam, double *dCdr);

/* Displacements of the penalisation points ("node_mod" points: the junctions between the penalised end segments and the
 * middle segment of every strut, LatticeSim.set_penalized_beams lattice_sim.py:245-308, which the reference meshes as
 * FE vertices and reads back in set_result_diplacement_on_lattice_object, full_scale_lattice_simulation.py:77-107).
 * They are interior points of the condensed strut: recovered in closed form from the end displacements u[6N].
 * out[12*b ..] = [u(q1)(3) th(q1)(3) u(q2)(3) th(q2)(3)], q1 next to beam_conn[2b], q2 next to beam_conn[2b+1]; an absent
 * segment gives the end node's own values. */
int pl_node_mod(pl_handle h, const double *u, double *out);

/* Strain energy 1/2 u^T K u (LatticeOpti.compute_compliance, lattice_opti.py:645-663 uses u^T K u). */
int pl_energy(pl_handle h, const double *u, double *energy);

/* Dense condensation of a node subset: S = K_BB - K_BI K_II^-1 K_IB for the boundary nodes listed, computed
 * column by column with the device PCG.  Replaces SchurComplement.calculate_schur_complement
 * (schur_complement.py:75-147).  S is [6*nb x 6*nb] row-major. */
int pl_schur(pl_handle h, const int32_t *boundary_nodes, int32_t nb, double rtol, int32_t max_iter, double *S);

/* Debug / test access to the condensed per-strut records: rec[8*B] = (a, c, e1, e2, e3, dx, dy, dz). */
int pl_get_records(pl_handle h, double *rec);

/* Measurement hooks (bench.py): run `reps` launches of one kernel on the handle's stream between two HIP
 * events and return the average milliseconds.  which: 0 = K*p (PCG operator), 1 = record build,
 * 2 = BSR fill, 3 = one full PCG iteration, 4 = BSR SpMV; on a multi-GPU handle also 5 = the interface all-reduce of
 * one K*p (staging kernels + RCCL) and 6 = the coarse-residual all-reduce - collective calls, every rank must make them.
 * With the multi-level PCG: 7 = K*p on fp32-stored vectors, 8 = one iteration of the fp32 inner PCG (precision 1),
 * 9 = one iteration of the mixed PCG (precision 2); 10 = the operator exactly as the next pl_solve applies it (BOTH passes
 * under node elimination, fp32-stored operands in the fp32 solver modes) - what a roofline of "K*p" must be priced with;
 * 11 = one whole PCG iteration exactly as the next pl_solve runs it (storage width and node elimination of its plan). */
int pl_time_kernel(pl_handle h, int which, int reps, double *avg_ms);
/* Algorithmic byte counts of SURVEY.md section 8(d) for this handle: out[0]=spmv, out[1]=pcg_iter, out[2]=bsr - with the
 * storage widths the next pl_solve uses: strut records and the explicit K are fp64 in every mode, the PCG vectors are 4 bytes
 * wide where opts.precision stores them in fp32 (precision 1: all of them; 2: p and K*p). */
int pl_algorithmic_bytes(pl_handle h, double *out3);
/* Forget what earlier solves taught this handle: the iteration count that places the first look at the residual history of
 * the next solve (pl_solve) and the previous solution a warm start would begin from.  bench.py uses it
 * to report the cold value of a loop of identical solves beside the ordinary one. */
int pl_forget_history(pl_handle h);

/* Test hook for the device dense SPD solver behind the two-level preconditioner (blocked Cholesky + inverse factor):
 * solves A x = b for a host SPD matrix A[n*n] (row-major) on `device`; quad (may be NULL) gets b^T A^-1 b.
 * fp32_factor != 0 stores the inverse factor W = L^-1 in fp32, as the preconditioner does (x is then W32^T W32 b:
 * accurate to ~1e-6 cond(A)^(1/2), which is all a preconditioner needs); 0 keeps it in fp64.
 * Returns PL_ERR_ARG if A is not positive definite. */
int pl_debug_spd_solve(int device, int32_t n, const double *A, const double *b, double *x, double *quad,
                       int32_t fp32_factor);

/* ---- multi-GPU (slab partition, RCCL) ---------------------------------------------------------------- */
/* Size of the opaque RCCL unique id the ranks must share (rank 0 fills it with pl_dist_unique_id). */
int pl_dist_unique_id_bytes(void);
int pl_dist_unique_id(void *id_out);
/* Loopback transport: an id (same size as the RCCL one) that makes pl_dist_init attach the handle to an IN-PROCESS group
 * instead of an RCCL communicator - `world` handles of one process on one device, every collective of the multi-GPU path
 * (interface rows, scalar blocks, coarse operator band) summed through device buffers.  Each rank must then be driven by
 * its own host thread (collective calls meet at a host barrier: pl_dist_init, pl_set_bc on an assembled handle,
 * pl_assemble, pl_solve, pl_spmv*, pl_reactions, pl_time_kernel), exactly as each rank of an RCCL run is driven by its
 * own process.  This is how the whole multi-rank solver path runs with world = 2 ... 16 on a one-GPU box (tests/
 * test_gpu_loopback.py); it also serves to run several sub-domains on one GPU.  A rank that never arrives at a
 * collective makes the others fail with PL_ERR_HIP after 120 s instead of hanging. */
int pl_dist_loopback_id(void *id_out);
/* Loopback groups only (no-op otherwise): declare the group broken, so that ranks waiting at a collective return
 * PL_ERR_HIP at once - what a driver calls when one of its rank threads failed outside a collective. */
int pl_dist_abort(pl_handle h);
/* Attach this handle (one per rank/GPU) to a communicator.  shared_nodes lists, for each node of THIS rank's
 * sub-lattice that also exists on other ranks, its local index and a global id (dense 0..n_shared_global-1);
 * partial forces on those nodes are summed across ranks after every local K*x. */
int pl_dist_init(pl_handle h, int rank, int world, const void *unique_id, const int32_t *shared_local,
                 const int32_t *shared_global, int32_t n_shared, int32_t n_shared_global);

/* ---- host-side lattice generation (no GPU involved) -------------------------------------------------------- */
/* Lattice.generate_lattice + Cell.generate_beams + define_beam_node_index (lattice.py:421-483,665-698,
 * cell.py:293-382) on flat arrays, multi-threaded: for every cell c and template strut s the two end points
 * tmpl[6s + 3e + k] * cell_size[3c + k] + cell_coord[3c + k]; nodes and struts de-duplicated through coordinates
 * rounded to 9 decimals (first creator wins); nodes numbered in (x, y, z) order, struts in (lower end, upper end) order.
 * cell_radii[c * n_geom + g] is the radius the struts of geometry g get in cell c (gradient already applied),
 * tmpl_type[s] the geometry of template strut s.  The result lives in a library-owned object: read the sizes from
 * *info, fetch into caller arrays of those sizes (any pointer may be NULL), free.  created_nodes[2 (c n_tmpl + s) + e] /
 * created_beam[c n_tmpl + s] give node and strut of every CREATED strut end (the hybrid-collision pass needs them).
 * Returns PL_ERR_STATE when the lattice is too irregular for the direct-address node table (caller falls back). */
typedef struct pl_lattice pl_lattice;
typedef struct {
  int64_t n_nodes, n_beams, n_cell_beam, n_cell_node, n_created;
} pl_lattice_info_t;
int pl_generate_lattice(int64_t n_cells, const double *cell_coord, const double *cell_size, const double *cell_radii,
                        int32_t n_geom, int32_t n_tmpl, const double *tmpl, const int32_t *tmpl_type,
                        pl_lattice **out, pl_lattice_info_t *info);
int pl_lattice_fetch(const pl_lattice *L, double *node_xyz, int32_t *beam_conn, double *beam_radius, int32_t *beam_type,
                     int32_t *beam_cell0, int64_t *cell_beam_ptr /*[n_cells+1]*/, int64_t *cell_beam_idx,
                     int64_t *cell_node_ptr /*[n_cells+1]*/, int64_t *cell_node_idx, int32_t *created_nodes,
                     int32_t *created_beam);
void pl_lattice_free(pl_lattice *L);

/* LatticeSim.set_penalized_beams (lattice_sim.py:245-308) on arrays: every strut becomes [pen(L_zone at point1) | middle
 * | pen(L_zone at point2)]; new points at end + (other - end) / round(length, 4) * L_zone (beam.py:135,300-312).  Outputs
 * per strut: the three geometric segment lengths (0 = absent), gmsh's element count int(len / mesh_size + 0.99) per
 * segment (lattice_generation.py:50-64) and the two penalisation points (NaN where absent).  lzone = NULL: no
 * penalisation (one segment per strut).  Host code, multi-threaded. */
int pl_penalize(int64_t n_beams, const double *node_xyz, const int32_t *beam_conn, const double *lzone /*[2B] or NULL*/,
                double mesh_size, double *seg_len /*[3B]*/, int32_t *seg_nsub /*[3B]*/, double *pen_xyz /*[6B]*/);

/* LatticeSim.define_node_index_boundary (lattice_sim.py:546-563) with the visit order of get_global_displacement
 * (:502-542) on arrays: a node gets a boundary index when it lies on the box of one of its cells (exact coordinate
 * comparison with the cell's origin / origin + size); indices are handed out in the order cells (ascending) then nodes of
 * the cell (ascending node index = coordinate order) first meet them.  index_boundary[n_nodes] (-1 elsewhere),
 * visit[<= n_nodes] = the nodes in that order, *n_visit their number.  Host code, multi-threaded. */
int pl_boundary_index(int64_t n_cells, const int64_t *cell_node_ptr, const int64_t *cell_node_idx, int64_t n_nodes,
                      const double *node_xyz, const double *cell_coord, const double *cell_size, int64_t *index_boundary,
                      int64_t *visit, int64_t *n_visit);

/* The same for the reference's own rows (LatticeSim(reference_compat=True): the rows of a cell are its design nodes AND the
 * penalisation points of its struts, visited in (round(x, 9), round(y, 9), round(z, 9), index) order - _sorted_nodes,
 * lattice_sim.py:193-199): every cell's rows are sorted by that key here, all cells in parallel. */
int pl_boundary_index_rows(int64_t n_cells, const int64_t *cell_node_ptr, const int64_t *cell_node_idx, int64_t n_nodes,
                           const double *node_xyz, const double *cell_coord, const double *cell_size,
                           int64_t *index_boundary, int64_t *visit, int64_t *n_visit);

/* Neighbour halo exchange ("sum of interface-node partial forces with the two neighbouring
 * slabs"): shared_peer[i] = the rank that holds the other copy of shared entry i of pl_dist_init (a node shared with
 * several ranks is listed once per peer there).  After this call the interface rows of every K*x travel by grouped
 * ncclSend / ncclRecv between neighbours - each rank moves its own planes only, concurrently - instead of the all-reduce
 * over all planes; the dot-product slots take a small all-reduce of their own.  Both ranks of a pair order their
 * common nodes by global interface id, so no further handshake is needed.  Collective: every rank must call it. */
int pl_dist_set_peers(pl_handle h, const int32_t *shared_peer /*[n_shared]*/);

#ifdef __cplusplus
}
#endif
#endif /* PYLATTICE_HIP_H */
