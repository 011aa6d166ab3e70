// libpylattice_hip.so - the C ABI declared in include/pylattice_hip.h (gfx950 / MI355X).  One translation unit:
// pl_context.h (handle state) <- pl_ops.h (operator launches) <- pl_assembly.h <- pl_solver.h <- this file.
#include "pl_solver.h"


// ==========================================================================================================
// C ABI
// ==========================================================================================================
namespace {
template <typename WT>
int debug_spd_solve_t(int device, int32_t n, const double *A, const double *b, double *x, double *quad) {
  if (n <= 0 || !A || !b || !x) return fail(PL_ERR_ARG, "pl_debug_spd_solve: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(PL_ERR_NODEVICE, "no HIP device visible");
  PL_HIP(hipSetDevice(device));
  const int np = (n + pl::kNB - 1) / pl::kNB * pl::kNB;
  std::vector<double> Ap((size_t)np * np, 0.0), bp(np, 0.0);
  for (int i = 0; i < np; ++i) {
    if (i < n) {
      std::memcpy(&Ap[(size_t)i * np], A + (size_t)i * n, n * sizeof(double));
      bp[i] = b[i];
    } else {
      Ap[(size_t)i * np + i] = 1.0;
    }
  }
  DevBuf<double> dA, dL, dD, db, dt, dy, dq;
  DevBuf<WT> dW, dWt;
  DevBuf<int> dinfo;
  PL_HIP(dA.alloc(Ap.size()));
  PL_HIP(dW.alloc(Ap.size()));
  PL_HIP(dL.alloc(Ap.size()));
  PL_HIP(dWt.alloc(Ap.size()));
  PL_HIP(hipMemset(dWt.p, 0, Ap.size() * sizeof(WT)));
  PL_HIP(dD.alloc((size_t)np * pl::kNB));
  PL_HIP(db.alloc(np));
  PL_HIP(dt.alloc(np));
  PL_HIP(dy.alloc(np));
  PL_HIP(dq.alloc(pl::kSlots));
  PL_HIP(dinfo.alloc(2));
  PL_HIP(hipMemcpy(dA.p, Ap.data(), Ap.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(db.p, bp.data(), np * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemset(dW.p, 0, Ap.size() * sizeof(WT)));
  PL_HIP(hipMemset(dq.p, 0, pl::kSlots * sizeof(double)));
  PL_HIP(hipMemset(dinfo.p, 0, 2 * sizeof(int)));
  pl::dense_factor_inverse(dA.p, dL.p, dW.p, dWt.p, dD.p, np, np, dinfo.p, 0, nullptr);
  pl::dense_apply(dW.p, dWt.p, np, np, db.p, dt.p, dy.p, dq.p, nullptr, nullptr);
  PL_HIP(hipGetLastError());
  PL_HIP(hipDeviceSynchronize());
  int info[2];
  PL_HIP(hipMemcpy(info, dinfo.p, sizeof(info), hipMemcpyDeviceToHost));
  if (info[0] != 0) return fail(PL_ERR_ARG, "pl_debug_spd_solve: matrix is not positive definite");
  std::vector<double> y(np), q(pl::kSlots);
  PL_HIP(hipMemcpy(y.data(), dy.p, np * sizeof(double), hipMemcpyDeviceToHost));
  PL_HIP(hipMemcpy(q.data(), dq.p, pl::kSlots * sizeof(double), hipMemcpyDeviceToHost));
  std::memcpy(x, y.data(), n * sizeof(double));
  if (quad) {
    *quad = 0.0;
    for (double v : q) *quad += v;
  }
  return PL_OK;
}

// ---- multi-GPU ------------------------------------------------------------------------------------------
}  // namespace

extern "C" {

const char *pl_last_error(void) { return g_err.c_str(); }
#ifndef PL_KP_HASH
#define PL_KP_HASH "unknown"
#endif
const char *pl_version(void) { return "pylattice_hip 0.2 (gfx950) kp=" PL_KP_HASH; }

uint32_t pl_opts_size(void) { return (uint32_t)sizeof(pl_opts_t); }
uint32_t pl_stats_size(void) { return (uint32_t)sizeof(pl_stats_t); }
uint32_t pl_abi_version(void) { return PL_ABI_VERSION; }

int pl_default_opts(pl_opts_t *o, uint32_t struct_size) {
  if (!o) return fail(PL_ERR_ARG, "pl_default_opts: null argument");
  if (struct_size != sizeof(pl_opts_t))
    return fail(PL_ERR_ARG, "pl_default_opts: the caller's pl_opts_t has " + std::to_string(struct_size) +
                                " bytes, this library's " + std::to_string(sizeof(pl_opts_t)) +
                                " (ABI version " + std::to_string(PL_ABI_VERSION) + "): rebuild the binding against include/pylattice_hip.h");
  std::memset(o, 0, sizeof(*o));
  o->struct_size = (uint32_t)sizeof(pl_opts_t);
  o->abi_version = PL_ABI_VERSION;
  o->young = 1013.0;   // VeroClear
  o->poisson = 0.3;
  o->kappa = 0.9;
  o->pen_coef = 1.5;
  o->device = 0;
  o->spmv_kernel = 0;
  o->precond = 1;
  o->reorder = 1;
  o->check_every = 0;   // adaptive
  return PL_OK;
}

namespace {
// pl_opts_t must carry the stamp of pl_default_opts of THIS library (include/pylattice_hip.h, "ABI handshake")
int check_opts_abi(const pl_opts_t *o, const char *who) {
  if (o->struct_size != sizeof(pl_opts_t) || o->abi_version != PL_ABI_VERSION)
    return fail(PL_ERR_ARG, std::string(who) + ": pl_opts_t was not initialised by pl_default_opts of this library (struct_size " +
                                std::to_string(o->struct_size) + " / abi " + std::to_string(o->abi_version) + ", expected " +
                                std::to_string(sizeof(pl_opts_t)) + " / " + std::to_string(PL_ABI_VERSION) + ")");
  return PL_OK;
}
inline bool multi_rank_handle(const pl_opts_t *o) { return o->grid_nodes > 0; }
// modes per aggregate of the dense level: 12 (rigid + strains) needs the 12-mode tile level, i.e. precond = 3 (either CG form)
inline int coarse_modes_of(const pl_opts_t *o, int64_t N = -1) {
  const bool tile12 = o->precond == 3 && o->tile_modes != 6;
  if (!tile12 || o->coarse_modes == 6) return 6;
  if (multi_rank_handle(o)) N = o->grid_nodes;       // every rank must decide alike: the node count of the whole lattice
  // automatic: from a quarter of a million nodes on (below, the longer set-up of the richer level costs what its
  // iterations save: 32^3 Octet 178 M beams/s with 6 modes, 169 M with 12)
  return (o->coarse_modes == 12 || N < 0 || N >= 250000) ? 12 : 6;
}
// dofs the dense level may have (see the comment at its set-up in pl_create)
inline int coarse_budget(const pl_opts_t *o, int64_t N) {
  const bool multi_rank = o->grid_nodes > 0;
  if (o->coarse_max_dofs > 0) return o->coarse_max_dofs;
  // 12 modes per aggregate: 5^3 aggregates (1 500 dofs) beat 7^3 x 6 (2 058) below a million nodes - 120 against 126
  // iterations at 50^3 Octet, 24 chain links instead of 33 (measured: 1 536 -> 203, 2 600 -> 200, 800 -> 193 M beams/s)
  // (the level's cost does not grow with the lattice, its benefit does: 100^3 BCC 3 072 -> 59.7, 6 144 -> 61.4 M beams/s;
  // 200 x 200 x 50 BCC + Octet 3 072 -> 104.9, 6 144 -> 136.4 M beams/s)
  // several GPUs: replicated on every rank and all-reduced per assembly, decided on the node count of the WHOLE lattice so
  // that every rank decides alike.  Emulated 4 / 8 ranks (50 x 50N x 50 Octet): 161 / 179 iterations with 6 modes and 3 072
  // dofs, 139 / 174 with 12 modes and 3 072, 120 / 139 with 12 modes and 6 144
  if (multi_rank) return (coarse_modes_of(o, N) == 12 && o->grid_nodes >= 2000000) ? 6144 : 3072;
  if (coarse_modes_of(o, N) == 12) return N >= 2000000 ? 6144 : (N >= 1000000 ? 3072 : (N >= 250000 ? 1536 : 768));
  return N >= 1000000 ? 3072 : 2100;
}
}  // namespace

int pl_create(const pl_mesh_t *m, const pl_opts_t *o, pl_handle *out) {
  if (!m || !o || !out) return fail(PL_ERR_ARG, "pl_create: null argument");
  if (int rc_abi = check_opts_abi(o, "pl_create")) return rc_abi;
  StageTimer stage("pl_create");
  if (m->n_nodes <= 0 || m->n_beams <= 0) return fail(PL_ERR_ARG, "pl_create: empty mesh");
  if (m->n_nodes >= (1LL << 31) - 64 || m->n_beams >= (1LL << 31) - 64)
    return fail(PL_ERR_ARG, "pl_create: more than 2^31 nodes/struts per handle");
  if (!m->node_xyz || !m->beam_conn || !m->beam_radius || !m->seg_len || !m->seg_nsub)
    return fail(PL_ERR_ARG, "pl_create: null mesh array");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PL_ERR_NODEVICE, "pl_create: no HIP device visible (libpylattice_hip has no CPU fallback)");
  if (o->device < 0 || o->device >= ndev) return fail(PL_ERR_ARG, "pl_create: bad device ordinal");
  if (o->precond == 5 && 6 * m->n_nodes > PL_DDM_DENSE_MAX)
    return fail(PL_ERR_ARG, "pl_create: precond = 5 (dense factorisation) serves at most PL_DDM_DENSE_MAX dofs");
  const int64_t N = m->n_nodes, B = m->n_beams;
  {
    std::atomic<int64_t> bad_ends{-1}, bad_radius{-1}, bad_len{-1}, bad_seg{-1};
    pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
      for (int64_t b = b0; b < b1; ++b) {
        const int32_t a = m->beam_conn[2 * b], d = m->beam_conn[2 * b + 1];
        if (a < 0 || d < 0 || a >= N || d >= N || a == d) bad_ends = b;
        if (!(m->beam_radius[b] > 0.0)) bad_radius = b;
        const double L = m->seg_len[3 * b] + m->seg_len[3 * b + 1] + m->seg_len[3 * b + 2];
        if (!(L > 0.0)) bad_len = b;
        for (int k = 0; k < 3; ++k)
          if (m->seg_len[3 * b + k] < 0.0 || (m->seg_len[3 * b + k] > 0.0 && m->seg_nsub[3 * b + k] < 1)) bad_seg = b;
      }
    });
    if (bad_ends >= 0)
      return fail(PL_ERR_ARG, "pl_create: strut " + std::to_string(bad_ends.load()) + " has invalid end nodes");
    if (bad_radius >= 0) return fail(PL_ERR_ARG, "pl_create: non-positive radius");
    if (bad_len >= 0) return fail(PL_ERR_ARG, "pl_create: strut with zero length");
    if (bad_seg >= 0) return fail(PL_ERR_ARG, "pl_create: bad segment data on strut " + std::to_string(bad_seg.load()));
  }
  stage.mark("validate");
  PL_HIP(hipSetDevice(o->device));
  pl_context *c = new pl_context();
  c->opt = *o;
  c->N = N;
  c->B = B;
  c->mat = {o->young, o->young / (2.0 * (1.0 + o->poisson)), o->kappa, o->pen_coef};
  c->lpn = o->lanes_per_node > 0 ? o->lanes_per_node : pl::kDefaultLPN;
  if (c->lpn != 1 && c->lpn != 2 && c->lpn != 4 && c->lpn != 8 && c->lpn != 16) {
    delete c;
    return fail(PL_ERR_ARG, "pl_create: lanes_per_node must be 0 (default), 1, 2, 4, 8 or 16");
  }
  auto bail = [&](int rc) {
    delete c;
    return rc;
  };
#define PL_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc) return bail(_rc);    \
  } while (0)
#define PL_HIPC(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return bail(fail(PL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  {   // the main stream carries the latency-bound chains (dense factorisation, PCG): it goes ahead of the bulk fills
    int prio_low = 0, prio_high = 0;
    PL_HIPC(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_low));
  }
  PL_HIPC(hipEventCreate(&c->ev0));
  PL_HIPC(hipEventCreate(&c->ev1));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_chol, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t0, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_p0, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_p1, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_fill1, hipEventDisableTiming));
  PL_HIPC(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
  {
    // The explicit K (2.5 GB of traffic) is filled BESIDE the latency-bound factorisation chain, on a stream whose CU mask
    // leaves k of every 8 CUs to the chain (next to an unmasked fill every link of the chain went from 44 to 57 us, which is
    // why the fill used to wait for the chain's end: assembly 2.20 -> 1.93 ms at 50^3 Octet with k = 4 or 5, 1.97 with 1,
    // 2.03 with 6).  The CUs are left out on a diagonal - bit i when (i / 8 + i) % 8 < k - which is k CUs of every XCD
    // whether the mask bits run XCD-major or round-robin over the XCDs.  PL_BSR_CUMASK=0: the old order.
    // The mask (8 words = 256 CUs in 8 XCDs) and the two rates of the split heuristic in pl_assemble were tuned on MI355X:
    // on a device with another CU count the masked stream is not created and the fill keeps the old order.
    const char *e = std::getenv("PL_BSR_CUMASK");
    const int k = e ? std::atoi(e) : 4;
    hipDeviceProp_t prop;
    const bool tuned_layout = hipGetDeviceProperties(&prop, o->device) == hipSuccess && prop.multiProcessorCount == 256;
    if (k > 0 && k < 8 && tuned_layout) {
      uint32_t mask[8];
      for (int w = 0; w < 8; ++w) {
        mask[w] = 0u;
        for (int b = 0; b < 32; ++b) {
          const int i = 32 * w + b;
          if ((i / 8 + i) % 8 >= k) mask[w] |= 1u << b;
        }
      }
      if (hipExtStreamCreateWithCUMask(&c->side_cu, 8, mask) != hipSuccess) {   // (no CU masks here: the old order)
        (void)hipGetLastError();
        c->side_cu = nullptr;
      }
    }
  }

  // node ordering on the device
  const double global_grid[7] = {o->grid_lo[0], o->grid_lo[1], o->grid_lo[2], o->grid_hi[0], o->grid_hi[1],
                                 o->grid_hi[2], (double)o->grid_nodes};
  c->perm.resize(N);
  std::iota(c->perm.begin(), c->perm.end(), 0);
  std::vector<int32_t> tile_start, tile_of;
  std::vector<int64_t> tile_brick;
  pl::BrickGrid grid;
  if (o->reorder == 1) {
    pl::spatial_order(m->node_xyz, N, c->perm, tile_start,
                      (double)(o->tile_nodes > 0 ? std::min(o->tile_nodes, pl::kTileMaxNodes) : 256), tile_brick,
                      grid, o->grid_nodes > 0 ? global_grid : nullptr,
                      (o->precond >= 2 && o->precond <= 4) ? coarse_budget(o, N) * 6 / coarse_modes_of(o, N) : 0);
    c->reordered = true;
    // Inside a tile: nodes of one KIND together (kind = the set of directions of a node's struts: in a periodic lattice
    // the corner nodes, the face centres of each orientation, ...; original order within a kind).  Struts of one
    // direction then join consecutive rows on both ends, which is what keeps the LDS-resident K*p (pl_tile.h) free of
    // bank conflicts; the other kernels do not care about the order inside a tile.
    {
      std::vector<uint64_t> kind((size_t)N, 0);
      pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t b = b0; b < b1; ++b) {
          const int32_t u = m->beam_conn[2 * b], v = m->beam_conn[2 * b + 1];
          uint64_t d = 0;
          for (int k = 0; k < 3; ++k) {
            const int64_t q = (int64_t)std::llround((m->node_xyz[3 * (size_t)v + k] - m->node_xyz[3 * (size_t)u + k]) * 4096.0);
            d = d * 0x9E3779B97F4A7C15ull + (uint64_t)(q + (1 << 20));
          }
          auto mix = [](uint64_t h) { h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; return h; };
          __atomic_fetch_add(&kind[u], mix(d), __ATOMIC_RELAXED);
          __atomic_fetch_add(&kind[v], mix(~d), __ATOMIC_RELAXED);
        }
      }, 1 << 16);
      const int64_t T = (int64_t)tile_start.size() - 1;
      pl::parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
        for (int64_t t = t0; t < t1; ++t)
          std::stable_sort(c->perm.begin() + tile_start[t], c->perm.begin() + tile_start[t + 1],
                           [&](int32_t l, int32_t r) { return kind[l] < kind[r]; });
      }, 16);
    }
  } else {
    pl::chunk_tiles(N, tile_start);
  }
  stage.mark("spatial order");
  c->iperm.resize(N);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i) c->iperm[c->perm[i]] = (int32_t)i;
  }, 1 << 16);
  if (o->condense >= 0 && o->precond >= 2 && o->precond <= 4 && o->precision != 2 && o->reorder == 1) {
    // Candidates for exact elimination inside the PCG (opts.condense): a greedy maximal independent set of the node
    // graph (no two share a strut; at least three struts each).  Inside every tile they are numbered LAST, so that the
    // vector kernels, which skip them, skip one contiguous run of rows per tile.
    std::vector<int64_t> aptr((size_t)N + 1, 0);
    pl::parallel_for(2 * B, [&](int64_t k0, int64_t k1, unsigned) {
      for (int64_t k = k0; k < k1; ++k)
        __atomic_fetch_add(&aptr[c->iperm[m->beam_conn[k]] + 1], (int64_t)1, __ATOMIC_RELAXED);
    }, 1 << 16);
    for (int64_t i = 0; i < N; ++i) aptr[i + 1] += aptr[i];
    std::vector<int32_t> adj((size_t)2 * B);
    {   // neighbour lists in any order: the greedy passes below only ask whether a neighbour is taken
      std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
      pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
        for (int64_t b = b0; b < b1; ++b) {
          const int32_t u = c->iperm[m->beam_conn[2 * b]], v = c->iperm[m->beam_conn[2 * b + 1]];
          adj[__atomic_fetch_add(&fill[u], (int64_t)1, __ATOMIC_RELAXED)] = v;
          adj[__atomic_fetch_add(&fill[v], (int64_t)1, __ATOMIC_RELAXED)] = u;
        }
      }, 1 << 16);
    }
    // two greedy passes: nodes strictly inside the bounding box first, then the ones on it.  Boundary conditions sit
    // on the faces (a node with a Dirichlet dof cannot be eliminated) and face nodes have fewer struts; on BCC this
    // picks the cell centres rather than the corners (measured: 702 -> 468 iterations at 50^3 against 575 the other way)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int64_t i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], m->node_xyz[3 * i + k]);
        hi[k] = std::max(hi[k], m->node_xyz[3 * i + k]);
      }
    std::vector<uint8_t> state((size_t)N, 0);
    for (int pass = 0; pass < 2; ++pass)
      for (int64_t i = 0; i < N; ++i) {
        if (state[i] || aptr[i + 1] - aptr[i] < 3) continue;
        const double *q3 = m->node_xyz + 3 * (size_t)c->perm[i];
        const bool on_box = q3[0] <= lo[0] || q3[0] >= hi[0] || q3[1] <= lo[1] || q3[1] >= hi[1] || q3[2] <= lo[2] ||
                            q3[2] >= hi[2];
        if (on_box != (pass == 1)) continue;
        state[i] = 1;
        for (int64_t q = aptr[i]; q < aptr[i + 1]; ++q)
          if (!state[adj[q]]) state[adj[q]] = 2;
      }
    int64_t n_cand = 0;
    for (int64_t i = 0; i < N; ++i) n_cand += state[i] == 1;
    // automatic mode: only when close to half of the unknowns can go (bipartite node graphs such as BCC: measured
    // 1.1-1.2 x faster solves; Octet, a quarter of the nodes: 1.2 x slower - every iteration pays a second K*p)
    const bool use = o->condense > 0 || (double)n_cand >= 0.45 * (double)N;
    const int64_t T = use ? (int64_t)tile_start.size() - 1 : 0;
    std::vector<int32_t> np(c->perm.size());
    if (use) c->h_cand.assign((size_t)N, 0);
    for (int64_t t = 0; t < T; ++t) {
      int64_t w = tile_start[t];
      for (int pass = 0; pass < 2; ++pass)
        for (int64_t i = tile_start[t]; i < tile_start[t + 1]; ++i)
          if ((state[i] == 1) == (pass == 1)) {
            c->h_cand[w] = (uint8_t)pass;
            np[w++] = c->perm[i];
          }
    }
    if (use) {
      c->perm.swap(np);
      for (int64_t i = 0; i < N; ++i) c->iperm[c->perm[i]] = (int32_t)i;
    }
  }

  stage.mark("candidates");
  std::vector<double> xyz((size_t)N * 3);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i)
      std::memcpy(&xyz[3 * i], m->node_xyz + 3 * (size_t)c->perm[i], 3 * sizeof(double));
  });
  std::vector<int32_t> conn((size_t)B * 2);
  {
    std::vector<int32_t> conn0((size_t)B * 2);
    pl::parallel_for(2 * B, [&](int64_t k0, int64_t k1, unsigned) {
      for (int64_t k = k0; k < k1; ++k) conn0[k] = c->iperm[m->beam_conn[k]];
    });
    pl::tile_strut_order(conn0, N, B, tile_start, tile_of, c->bperm, xyz.data());
    pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
      for (int64_t b = b0; b < b1; ++b) {
        conn[2 * b] = conn0[2 * (size_t)c->bperm[b]];
        conn[2 * b + 1] = conn0[2 * (size_t)c->bperm[b] + 1];
      }
    });
  }
  stage.mark("strut order");
  std::vector<double> radius(B), seg_len((size_t)B * 3);
  std::vector<int32_t> seg_nsub((size_t)B * 3);
  pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
    for (int64_t b = b0; b < b1; ++b) {
      const size_t ob = (size_t)c->bperm[b];
      radius[b] = m->beam_radius[ob];
      for (int k = 0; k < 3; ++k) {
        seg_len[3 * b + k] = m->seg_len[3 * ob + k];
        seg_nsub[3 * b + k] = m->seg_nsub[3 * ob + k];
      }
    }
  });

  PL_HIPC(c->xyz.alloc(N * 3));
  PL_HIPC(c->conn.alloc(B * 2));
  PL_HIPC(c->radius.alloc(B));
  PL_HIPC(c->seg_len.alloc(B * 3));
  PL_HIPC(c->seg_nsub.alloc(B * 3));
  PL_HIPC(c->rec.alloc(B));
  // opts.compact_records (default on): the tile K*p streams 40-byte records when no palette applies
  if (o->compact_records >= 0 && (o->spmv_kernel == 0 || o->spmv_kernel == 3)) PL_HIPC(c->rec5.alloc((size_t)B * 5));
  PL_HIPC(hipMemcpy(c->xyz.p, xyz.data(), xyz.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->conn.p, conn.data(), conn.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->radius.p, radius.data(), B * sizeof(double), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->seg_len.p, seg_len.data(), 3 * B * sizeof(double), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->seg_nsub.p, seg_nsub.data(), 3 * B * sizeof(int32_t), hipMemcpyHostToDevice));
  stage.mark("permute + upload");
  PL_TRY(build_incidence(c, conn));
  stage.mark("incidence + BSR pattern");
  {
    int rc = pl::build_tile_plan(c->tile, conn, N, B, tile_start, tile_of, xyz.data());
    if (rc) return bail(fail(PL_ERR_HIP, "pl_create: building the LDS tile plan failed (" + std::to_string(rc) + ")"));
  }
  if (o->palette && c->tile.vis_ready && std::getenv("PL_ROWS")) {   // K*p by rows (pl_rows.h, opt-in): palette form only
    int rc = pl::build_row_plan(c->rows, conn, N, B, tile_start, xyz.data());
    if (rc) return bail(fail(PL_ERR_HIP, "pl_create: building the row plan failed (" + std::to_string(rc) + ")"));
  }
  c->h_tile_start = tile_start;
  stage.mark("tile + row plan");
  if (o->precond >= 2 && o->precond <= 4) {
    if (!c->reordered) return bail(fail(PL_ERR_ARG, "pl_create: precond = 2/3/4 (multi-level) needs reorder = 1"));
    c->coarse.tile_level = (o->precond >= 3);
    // 12-mode tile level (rigid + uniform strains): single-GPU handles in the ordinary CG form; opts.tile_modes = 6 keeps
    // the rigid-body blocks
    c->coarse.tile_modes = (o->tile_modes == 6 || o->precond == 4) ? 6 : 12;
    // default size of the dense level: its factorisation is a ~45 us-per-64-dofs latency chain in every assembly, its
    // benefit grows with the cost of an iteration - up to 1 M nodes on one GPU the optimum is ~2 000 dofs (measured on
    // 50^3 Octet: 7^3 aggregates 18.9 ms per step, 8^3 20.7 ms), beyond that and in multi-rank runs (collectives in
    // every iteration) the full 3 072
    const bool multi_rank = o->grid_nodes > 0;
    const int max_dofs = coarse_budget(o, N);
    int rc = pl::coarse_setup(c->coarse, tile_start, tile_brick, grid, xyz.data(), N, max_dofs, conn, false, multi_rank,
                              coarse_modes_of(o, N));
    if (rc == 4)
      return bail(fail(PL_ERR_ARG, "pl_create: a strut spans more than neighbouring aggregates; the band-packed "
                                   "all-reduce of the coarse operator of a multi-GPU handle cannot hold it (use "
                                   "precond = 1 or a smaller coarse_max_dofs)"));
    if (rc) return bail(fail(PL_ERR_HIP, "pl_create: coarse-space setup failed (" + std::to_string(rc) + ")"));
    if (o->coarse_storage != 0 && o->coarse_storage != 16 && o->coarse_storage != 32)
      return bail(fail(PL_ERR_ARG, "pl_create: coarse_storage must be 0 (automatic), 16 (bfloat16) or 32 (fp32)"));
    c->coarse.w16 = o->coarse_storage == 16 || (o->coarse_storage == 0 && c->coarse.ncp >= 1024);
    PL_HIPC(c->sharedbits.alloc(N));
    PL_HIPC(c->maskL.alloc(N));
    PL_HIPC(hipMemset(c->sharedbits.p, 0, N));
    if (o->precond == 4) {
      const int maxL = o->local_max_dofs > 0 ? o->local_max_dofs : 3072;
      rc = pl::coarse_setup(c->coarseL, tile_start, tile_brick, grid, xyz.data(), N, maxL, conn, true);
      if (rc) return bail(fail(PL_ERR_HIP, "pl_create: local coarse-space setup failed (" + std::to_string(rc) + ")"));
      c->coarseL.tile_level = false;
      c->coarseL.w16 = o->coarse_storage == 16 || (o->coarse_storage == 0 && c->coarseL.ncp >= 1024);
    }
  }

  stage.mark("coarse setup");
  const size_t n6 = (size_t)N * 6;
  PL_HIPC(c->fixed.alloc(n6));
  PL_HIPC(c->fixedbits.alloc(N));
  PL_HIPC(c->ubar.alloc(n6));
  PL_HIPC(c->f.alloc(n6));
  // (precond = 5: padded to the block size of the dense solver and zeroed - its GEMVs read / write whole blocks)
  const size_t vpad = o->precond == 5 ? (size_t)pl::kNB : 0;
  for (DevBuf<double> *v : {&c->diag, &c->dinv, &c->x, &c->r, &c->z, &c->p, &c->Ap, &c->tmp, &c->tmp2}) {
    PL_HIPC(v->alloc(n6 + vpad));
    if (vpad) PL_HIPC(hipMemset(v->p, 0, (n6 + vpad) * sizeof(double)));
  }
  PL_HIPC(c->scal.alloc(2 * pl::S_COUNT * pl::kSlots));   // two sets, selected by iteration parity
  PL_HIPC(hipMemset(c->fixed.p, 0, n6));
  PL_HIPC(hipMemset(c->fixedbits.p, 0, N));
  PL_HIPC(hipMemset(c->ubar.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipMemset(c->f.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipDeviceSynchronize());
#undef PL_TRY
#undef PL_HIPC
  stage.mark("vector buffers");
  *out = c;
  return PL_OK;
}

// Everything of a DDM handle that depends on WHICH matrix a cell uses (not on the matrix values): the cells sorted by matrix
// id (k_ddm_cell_product_lds), and the tiles of 16 cells of one class with their gather positions (k_ddm_cell_product_mfma).
static int ddm_build_classes(pl_context *c) {
  const int64_t n_cells = c->ddm_cells;
  const int nb = c->ddm_nb, m = 6 * nb;
  const int32_t *cell_S = c->h_ddm_cell_S.data(), *cell_nodes = c->h_ddm_cell_nodes.data();
  std::vector<int32_t> order((size_t)n_cells);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t l, int32_t r) { return cell_S[l] < cell_S[r]; });
  PL_HIP(c->ddm_order.alloc(order.size()));
  PL_HIP(hipMemcpy(c->ddm_order.p, order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  // tiles of 16 cells of one matrix class for the matrix-pipe product (k_ddm_cell_product_mfma)
  std::vector<int32_t> tiles, tile_S;
  for (size_t q = 0; q < order.size();) {
    const int32_t id = cell_S[order[q]];
    size_t e = q;
    while (e < order.size() && cell_S[order[e]] == id) ++e;
    for (size_t a = q; a < e; a += 16) {
      for (size_t k = 0; k < 16; ++k) tiles.push_back(a + k < e ? order[a + k] : -1);
      tile_S.push_back(id);
    }
    q = e;
  }
  c->ddm_n_tiles = (int64_t)tile_S.size();
  if (m <= 192) {
    const int KS = pl::ddm_mfma_ks(m);
    std::vector<int32_t> gidx((size_t)c->ddm_n_tiles * KS * 64, -1);
    for (int64_t t = 0; t < c->ddm_n_tiles; ++t)
      for (int kk = 0; kk < KS; ++kk)
        for (int lane = 0; lane < 64; ++lane) {
          const int32_t cell = tiles[16 * t + (lane & 15)];
          const int k = 4 * kk + (lane >> 4);
          if (cell >= 0 && k < m)
            gidx[((size_t)t * KS + kk) * 64 + lane] = 6 * cell_nodes[(int64_t)cell * nb + k / 6] + k % 6;
        }
    PL_HIP(c->ddm_tile_gidx.alloc(std::max<size_t>(1, gidx.size())));
    if (!gidx.empty())
      PL_HIP(hipMemcpy(c->ddm_tile_gidx.p, gidx.data(), gidx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  PL_HIP(c->ddm_tiles.alloc(std::max<size_t>(1, tiles.size())));
  PL_HIP(c->ddm_tile_S.alloc(std::max<size_t>(1, tile_S.size())));
  if (!tiles.empty()) {
    PL_HIP(hipMemcpy(c->ddm_tiles.p, tiles.data(), tiles.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    PL_HIP(hipMemcpy(c->ddm_tile_S.p, tile_S.data(), tile_S.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  return PL_OK;
}

// New cell matrices on an existing DDM handle (a design loop changes the radii, hence every S_c, between two solves; the
// topology - which nodes a cell couples - stays): uploads the palette again and, only when the cells' matrix ids changed,
// re-cuts the class tiles.  What it saves is pl_destroy + pl_create_ddm per design iteration (streams, events, a dozen
// allocations, the node -> cell incidence).  The preconditioner data of the handle is dropped: pl_assemble before pl_solve.
int pl_ddm_update_matrices(pl_handle h, int32_t n_S, const double *S, const int32_t *cell_S) {
  if (!valid(h) || !S || !cell_S || n_S <= 0) return fail(PL_ERR_ARG, "pl_ddm_update_matrices: bad argument");
  if (h->opkind != 1) return fail(PL_ERR_STATE, "pl_ddm_update_matrices: not a DDM handle");
  for (int64_t c = 0; c < h->ddm_cells; ++c)
    if (cell_S[c] < 0 || cell_S[c] >= n_S) return fail(PL_ERR_ARG, "pl_ddm_update_matrices: matrix id out of range");
  PL_HIP(hipSetDevice(h->opt.device));
  PL_HIP(hipStreamSynchronize(h->stream));
  const int m = 6 * h->ddm_nb;
  const size_t cnt = (size_t)n_S * m * m;
  if (h->ddm_St.n != cnt) PL_HIP(h->ddm_St.alloc(cnt));
  if (h->ddm_Sraw.n < cnt) PL_HIP(h->ddm_Sraw.alloc(cnt));
  PL_HIP(hipMemcpyAsync(h->ddm_Sraw.p, S, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(pl::k_ddm_transpose_palette, dim3(grid_for((int64_t)cnt)), dim3(pl::kBlock), 0, h->stream, (int64_t)n_S, m,
                     (const double *)h->ddm_Sraw.p, h->ddm_St.p);
  PL_HIP(hipStreamSynchronize(h->stream));
  PL_HIP(hipGetLastError());
  h->ddm_n_S = n_S;
  if (!std::equal(h->h_ddm_cell_S.begin(), h->h_ddm_cell_S.end(), cell_S)) {
    h->h_ddm_cell_S.assign(cell_S, cell_S + h->ddm_cells);
    PL_HIP(hipMemcpy(h->ddm_cell_S.p, cell_S, h->ddm_cells * sizeof(int32_t), hipMemcpyHostToDevice));
    int rc = ddm_build_classes(h);
    if (rc) return rc;
  }
  h->assembled = false;
  h->dd_ready = false;
  h->dd_blocks = false;
  h->dd2_ready = false;
  return PL_OK;
}

int pl_create_ddm(int64_t n_nodes, int64_t n_cells, int32_t nb, const int32_t *cell_nodes, int32_t n_S,
                  const double *S, const int32_t *cell_S, const pl_opts_t *o, pl_handle *out) {
  if (!cell_nodes || !S || !cell_S || !o || !out) return fail(PL_ERR_ARG, "pl_create_ddm: null argument");
  if (int rc_abi = check_opts_abi(o, "pl_create_ddm")) return rc_abi;
  if (n_nodes <= 0 || n_cells <= 0 || nb <= 0 || n_S <= 0) return fail(PL_ERR_ARG, "pl_create_ddm: empty problem");
  if (6 * nb > pl::kDdmMaxM) return fail(PL_ERR_ARG, "pl_create_ddm: more than 27 boundary nodes per cell");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PL_ERR_NODEVICE, "pl_create_ddm: no HIP device visible (libpylattice_hip has no CPU fallback)");
  if (o->device < 0 || o->device >= ndev) return fail(PL_ERR_ARG, "pl_create_ddm: bad device ordinal");
  for (int64_t k = 0; k < n_cells * nb; ++k)
    if (cell_nodes[k] < 0 || cell_nodes[k] >= n_nodes) return fail(PL_ERR_ARG, "pl_create_ddm: node id out of range");
  for (int64_t c = 0; c < n_cells; ++c)
    if (cell_S[c] < 0 || cell_S[c] >= n_S) return fail(PL_ERR_ARG, "pl_create_ddm: matrix id out of range");
  PL_HIP(hipSetDevice(o->device));
  pl_context *c = new pl_context();
  c->opt = *o;
  // 0: the reference's plain CG; 1: Jacobi on the assembled Schur diagonal; 2: the reference's own preconditioner, the
  // factorised assembled Schur matrix (dense Cholesky on the device, hence the size limit)
  c->opt.precond = (o->precond >= 1 && o->precond <= 4) ? o->precond : 0;     // 3: node-block Jacobi (6 x 6 blocks of G);
                                                                              // 4: + a dense level on node aggregates (pl_ddm_set_geometry)
  if (c->opt.precond == 2 && 6 * n_nodes > PL_DDM_DENSE_MAX) {
    delete c;
    return fail(PL_ERR_ARG, "pl_create_ddm: precond = 2 factorises a dense (6 n_nodes)^2 matrix; limit is " +
                                std::to_string(PL_DDM_DENSE_MAX) + " dofs (use precond = 3 or 1)");
  }
  c->opkind = 1;
  c->N = n_nodes;
  c->B = 0;
  c->ddm_cells = n_cells;
  c->ddm_nb = nb;
  {   // dofs coupled by a cell are at most 6 * (max - min node id) + 5 apart: the band of the assembled matrix
    int64_t span = 0;
    for (int64_t cc = 0; cc < n_cells; ++cc) {
      const int32_t *nd = cell_nodes + cc * nb;
      span = std::max<int64_t>(span, *std::max_element(nd, nd + nb) - *std::min_element(nd, nd + nb));
    }
    c->dd_bw = (int)((6 * span + 5) / pl::kNB + 1);
  }
  auto bail = [&](int rc) {
    delete c;
    return rc;
  };
#define PL_HIPC(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess)                                                              \
      return bail(fail(PL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  {   // the main stream carries the latency-bound chains (dense factorisation, PCG): it goes ahead of the bulk fills
    int prio_low = 0, prio_high = 0;
    PL_HIPC(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high));
    PL_HIPC(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_low));
  }
  PL_HIPC(hipEventCreate(&c->ev0));
  PL_HIPC(hipEventCreate(&c->ev1));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_chol, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t0, hipEventDisableTiming));
  PL_HIPC(hipEventCreateWithFlags(&c->ev_t1, hipEventDisableTiming));
  PL_HIPC(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
  c->perm.resize(n_nodes);
  std::iota(c->perm.begin(), c->perm.end(), 0);
  c->iperm = c->perm;
  const int m = 6 * nb;
  std::vector<double> St((size_t)n_S * m * m);
  for (int s = 0; s < n_S; ++s)
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) St[((size_t)s * m + j) * m + i] = S[((size_t)s * m + i) * m + j];
  PL_HIPC(c->ddm_cell_nodes.alloc((size_t)n_cells * nb));
  PL_HIPC(c->ddm_cell_S.alloc(n_cells));
  PL_HIPC(c->ddm_St.alloc(St.size()));
  PL_HIPC(hipMemcpy(c->ddm_cell_nodes.p, cell_nodes, (size_t)n_cells * nb * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->ddm_cell_S.p, cell_S, n_cells * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIPC(hipMemcpy(c->ddm_St.p, St.data(), St.size() * sizeof(double), hipMemcpyHostToDevice));
  {
    if (n_cells * nb >= (1LL << 31)) return bail(fail(PL_ERR_ARG, "pl_create_ddm: more than 2^31 cell-node entries"));
    std::vector<int64_t> nptr((size_t)n_nodes + 1, 0);
    for (int64_t k = 0; k < n_cells * nb; ++k) nptr[cell_nodes[k] + 1]++;
    for (int64_t i = 0; i < n_nodes; ++i) nptr[i + 1] += nptr[i];
    std::vector<int32_t> nent((size_t)n_cells * nb);
    std::vector<int64_t> fill(nptr.begin(), nptr.end() - 1);
    for (int64_t k = 0; k < n_cells * nb; ++k) nent[fill[cell_nodes[k]]++] = (int32_t)k;   // cell-major: fixed sum order
    PL_HIPC(c->ddm_node_ptr.alloc(nptr.size()));
    PL_HIPC(c->ddm_node_ent.alloc(nent.size()));
    PL_HIPC(c->ddm_stage.alloc((size_t)n_cells * m));
    PL_HIPC(hipMemcpy(c->ddm_node_ptr.p, nptr.data(), nptr.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    PL_HIPC(hipMemcpy(c->ddm_node_ent.p, nent.data(), nent.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    c->h_ddm_cell_nodes.assign(cell_nodes, cell_nodes + n_cells * nb);
    c->h_ddm_cell_S.assign(cell_S, cell_S + n_cells);
    c->ddm_n_S = n_S;
    {
      int rcc = ddm_build_classes(c);
      if (rcc) return bail(rcc);
    }
  }
  const size_t n6 = (size_t)n_nodes * 6;
  PL_HIPC(c->fixed.alloc(n6));
  PL_HIPC(c->fixedbits.alloc(n_nodes));
  PL_HIPC(c->ubar.alloc(n6));
  PL_HIPC(c->f.alloc(n6));
  // (vectors padded to the block size of the dense solver and zeroed: its GEMVs read / write whole blocks)
  for (DevBuf<double> *v : {&c->diag, &c->dinv, &c->x, &c->r, &c->z, &c->p, &c->Ap, &c->tmp, &c->tmp2}) {
    PL_HIPC(v->alloc(n6 + pl::kNB));
    PL_HIPC(hipMemset(v->p, 0, (n6 + pl::kNB) * sizeof(double)));
  }
  PL_HIPC(c->scal.alloc(2 * pl::S_COUNT * pl::kSlots));
  PL_HIPC(hipMemset(c->fixed.p, 0, n6));
  PL_HIPC(hipMemset(c->fixedbits.p, 0, n_nodes));
  PL_HIPC(hipMemset(c->ubar.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipMemset(c->f.p, 0, n6 * sizeof(double)));
  PL_HIPC(hipDeviceSynchronize());
#undef PL_HIPC
  *out = c;
  return PL_OK;
}

// Node positions of a DDM handle -> aggregates of the two-level preconditioner (opts.precond = 4): boxes of a regular grid over
// the bounding box, as many as opts.coarse_max_dofs / 12 allows (default 1 536 dofs = 128 aggregates) but at least ~27 nodes
// each, numbered with the longest axis slowest (narrow band of A_c); the band is taken from the cells themselves.
int pl_ddm_set_geometry(pl_handle h, const double *node_xyz) {
  if (!valid(h) || !node_xyz) return fail(PL_ERR_ARG, "pl_ddm_set_geometry: null argument");
  if (h->opkind != 1) return fail(PL_ERR_STATE, "pl_ddm_set_geometry: not a DDM handle");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t N = h->N;
  h->dd2_plan = false;
  h->dd2_ready = false;
  h->assembled = false;
  double lo[3], hi[3];
  for (int k = 0; k < 3; ++k) lo[k] = hi[k] = node_xyz[k];
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < 3; ++k) {
      const double v = node_xyz[3 * i + k];
      if (!std::isfinite(v)) return fail(PL_ERR_ARG, "pl_ddm_set_geometry: non-finite coordinate");
      lo[k] = std::min(lo[k], v);
      hi[k] = std::max(hi[k], v);
    }
  const int budget = h->opt.coarse_max_dofs > 0 ? h->opt.coarse_max_dofs : 1536;
  const int64_t n_target = std::max<int64_t>(1, std::min<int64_t>(budget / pl::kDdmModes, N / 27));
  double ext[3];
  for (int k = 0; k < 3; ++k) ext[k] = hi[k] - lo[k];
  const double emax = std::max(ext[0], std::max(ext[1], ext[2]));
  int64_t na[3] = {1, 1, 1};
  if (emax > 0.0) {     // boxes as close to cubes as the count allows: na_k ~ ext_k / side, side shrunk while the product fits
    double side = emax;
    for (int it = 0; it < 200; ++it) {
      int64_t t[3];
      for (int k = 0; k < 3; ++k) t[k] = std::max<int64_t>(1, (int64_t)std::floor(ext[k] / (side * 0.96) + 0.5));
      if (t[0] * t[1] * t[2] > n_target) break;
      for (int k = 0; k < 3; ++k) na[k] = t[k];
      side *= 0.96;
    }
  }
  const int n_agg = (int)(na[0] * na[1] * na[2]);
  int ax[3] = {0, 1, 2};
  std::stable_sort(ax, ax + 3, [&](int l, int r) { return na[l] > na[r]; });
  std::vector<int32_t> agg((size_t)N), ptr((size_t)n_agg + 1, 0), nodes((size_t)N);
  for (int64_t i = 0; i < N; ++i) {
    int64_t b[3];
    for (int k = 0; k < 3; ++k) {
      b[k] = ext[k] > 0.0 ? (int64_t)std::floor((node_xyz[3 * i + k] - lo[k]) / ext[k] * (double)na[k]) : 0;
      b[k] = std::min<int64_t>(std::max<int64_t>(b[k], 0), na[k] - 1);
    }
    agg[(size_t)i] = (int32_t)((b[ax[0]] * na[ax[1]] + b[ax[1]]) * na[ax[2]] + b[ax[2]]);
    ptr[(size_t)agg[(size_t)i] + 1]++;
  }
  for (int a = 0; a < n_agg; ++a) ptr[(size_t)a + 1] += ptr[(size_t)a];
  {
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < N; ++i) nodes[(size_t)fill[(size_t)agg[(size_t)i]]++] = (int32_t)i;
  }
  std::vector<double> cen((size_t)n_agg * 3);
  for (int a = 0; a < n_agg; ++a) {
    int64_t b[3];
    b[ax[0]] = a / (na[ax[1]] * na[ax[2]]);
    b[ax[1]] = (a / na[ax[2]]) % na[ax[1]];
    b[ax[2]] = a % na[ax[2]];
    for (int k = 0; k < 3; ++k) cen[3 * (size_t)a + k] = lo[k] + ((double)b[k] + 0.5) * ext[k] / (double)na[k];
  }
  // band of A_c: the largest difference of aggregate numbers inside one cell
  std::vector<int32_t> cn((size_t)h->ddm_cells * h->ddm_nb);
  PL_HIP(hipMemcpy(cn.data(), h->ddm_cell_nodes.p, cn.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  int64_t max_diff = 0;
  for (int64_t c = 0; c < h->ddm_cells; ++c) {
    int32_t amin = n_agg, amax = -1;
    for (int i = 0; i < h->ddm_nb; ++i) {
      const int32_t a = agg[(size_t)cn[(size_t)c * h->ddm_nb + i]];
      amin = std::min(amin, a);
      amax = std::max(amax, a);
    }
    max_diff = std::max<int64_t>(max_diff, amax - amin);
  }
  pl::Coarse &cs = h->dd2;
  const int nc = pl::kDdmModes * n_agg, ncp = (nc + pl::kNB - 1) / pl::kNB * pl::kNB;
  if (cs.ncp != ncp) {
    for (void **q : {(void **)&cs.Ac, (void **)&cs.Lf, (void **)&cs.W, (void **)&cs.Wt, (void **)&cs.Dinv, (void **)&cs.rc,
                     (void **)&cs.yc, (void **)&cs.tv, (void **)&cs.info, (void **)&cs.bar, (void **)&cs.Ainv})
      if (*q) {
        (void)hipFree(*q);
        *q = nullptr;
      }
    const size_t n2 = (size_t)ncp * ncp;
    PL_HIP(hipMalloc((void **)&cs.Ac, n2 * sizeof(double)));
    PL_HIP(hipMalloc((void **)&cs.Lf, n2 * sizeof(double)));
    PL_HIP(hipMalloc((void **)&cs.W, n2 * sizeof(float)));
    PL_HIP(hipMalloc((void **)&cs.Wt, n2 * sizeof(float)));
    PL_HIP(hipMalloc((void **)&cs.Dinv, (size_t)ncp * pl::kNB * sizeof(double)));
    PL_HIP(hipMalloc((void **)&cs.rc, (size_t)(ncp + 2 * pl::kSlots) * sizeof(double)));
    PL_HIP(hipMalloc((void **)&cs.yc, (size_t)ncp * sizeof(double)));
    PL_HIP(hipMalloc((void **)&cs.tv, (size_t)ncp * sizeof(double)));
    PL_HIP(hipMalloc((void **)&cs.info, 2 * sizeof(int)));
    PL_HIP(hipMalloc((void **)&cs.bar, sizeof(unsigned)));
    if (ncp <= pl::kOneGemvMaxDofs) PL_HIP(hipMalloc((void **)&cs.Ainv, n2 * sizeof(float)));
    PL_HIP(hipMemset(cs.Lf, 0, n2 * sizeof(double)));
    PL_HIP(hipMemset(cs.W, 0, n2 * sizeof(float)));
    PL_HIP(hipMemset(cs.Wt, 0, n2 * sizeof(float)));
    PL_HIP(hipMemset(cs.rc, 0, (size_t)(ncp + 2 * pl::kSlots) * sizeof(double)));
    PL_HIP(hipMemset(cs.yc, 0, (size_t)ncp * sizeof(double)));
  }
  cs.n_agg = n_agg;
  cs.nc = nc;
  cs.ncp = ncp;
  cs.cm = pl::kDdmModes;
  cs.bw_blocks = (int)((pl::kDdmModes * (max_diff + 1) + pl::kNB - 1) / pl::kNB + 1);
  cs.w16 = h->opt.coarse_storage == 16 || (h->opt.coarse_storage == 0 && ncp >= 1024);
  cs.ainv_ready = false;
  h->dd2_n_agg = n_agg;
  PL_HIP(h->dd2_xyz.alloc((size_t)N * 3));
  PL_HIP(h->dd2_cen.alloc(cen.size()));
  PL_HIP(h->dd2_agg.alloc(agg.size()));
  PL_HIP(h->dd2_ptr.alloc(ptr.size()));
  PL_HIP(h->dd2_nodes.alloc(nodes.size()));
  PL_HIP(hipMemcpy(h->dd2_xyz.p, node_xyz, (size_t)N * 3 * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->dd2_cen.p, cen.data(), cen.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->dd2_agg.p, agg.data(), agg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->dd2_ptr.p, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->dd2_nodes.p, nodes.data(), nodes.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  h->dd2_plan = true;
  return PL_OK;
}

int pl_ddm_set_preconditioner(pl_handle h, int32_t n_S, const double *S, const int32_t *cell_S) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_ddm_set_preconditioner: null handle");
  if (h->opkind != 1) return fail(PL_ERR_STATE, "pl_ddm_set_preconditioner: not a DDM handle");
  PL_HIP(hipSetDevice(h->opt.device));
  h->assembled = false;
  if (!S) {   // back to the operator's own matrices
    h->ddm_have_P = false;
    return PL_OK;
  }
  if (n_S <= 0 || !cell_S) return fail(PL_ERR_ARG, "pl_ddm_set_preconditioner: bad argument");
  for (int64_t c = 0; c < h->ddm_cells; ++c)
    if (cell_S[c] < 0 || cell_S[c] >= n_S) return fail(PL_ERR_ARG, "pl_ddm_set_preconditioner: matrix id out of range");
  const int m = 6 * h->ddm_nb;
  std::vector<double> Pt((size_t)n_S * m * m);
  for (int s = 0; s < n_S; ++s)
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) Pt[((size_t)s * m + j) * m + i] = S[((size_t)s * m + i) * m + j];
  PL_HIP(h->ddm_Pt.alloc(Pt.size()));
  PL_HIP(h->ddm_cell_P.alloc(h->ddm_cells));
  PL_HIP(hipMemcpy(h->ddm_Pt.p, Pt.data(), Pt.size() * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->ddm_cell_P.p, cell_S, h->ddm_cells * sizeof(int32_t), hipMemcpyHostToDevice));
  h->ddm_have_P = true;
  return PL_OK;
}

// G = sum_c B^T Shat B on the free dofs (unit diagonal elsewhere), Cholesky + explicit inverse factor on the device.
static int ensure_bsr_buffers(pl_context *h) {
  if (h->bsr_vals.p) return PL_OK;
  PL_HIP(h->bsr_rowptr.alloc(h->N + 1));
  PL_HIP(h->bsr_col.alloc(h->nblk));
  PL_HIP(h->bsr_vals.alloc((size_t)h->nblk * 36));
  PL_HIP(hipMemcpy(h->bsr_rowptr.p, h->h_rowptr.data(), (h->N + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->bsr_col.p, h->h_col.data(), h->nblk * sizeof(int32_t), hipMemcpyHostToDevice));
  return PL_OK;
}

static int ddm_factor_preconditioner(pl_context *h) {
  const int64_t n6 = h->N * 6;
  const int np = (int)((n6 + pl::kNB - 1) / pl::kNB * pl::kNB);
  if (h->dd_n != np) {
    const size_t nn = (size_t)np * np;
    PL_HIP(h->dd_A.alloc(nn));
    PL_HIP(h->dd_Lf.alloc(nn));
    PL_HIP(h->dd_W.alloc(nn));
    PL_HIP(h->dd_Wt.alloc(nn));
    PL_HIP(h->dd_Dinv.alloc((size_t)np * pl::kNB));
    PL_HIP(h->dd_tv.alloc(np));
    PL_HIP(h->dd_info.alloc(2));
    PL_HIP(hipMemsetAsync(h->dd_W.p, 0, nn * sizeof(double), h->stream));
    PL_HIP(hipMemsetAsync(h->dd_Wt.p, 0, nn * sizeof(double), h->stream));
    h->dd_n = np;
  }
  h->dd_ready = false;
  PL_HIP(hipMemsetAsync(h->dd_A.p, 0, (size_t)np * np * sizeof(double), h->stream));
  PL_HIP(hipMemsetAsync(h->dd_info.p, 0, 2 * sizeof(int), h->stream));
  const int m = 6 * h->ddm_nb;
  const int64_t ne = h->ddm_cells * m * m;
  const uint8_t *fx = h->have_bc ? h->fixed.p : (const uint8_t *)nullptr;
  hipLaunchKernelGGL(pl::k_ddm_dense_assemble, dim3(grid_for(ne)), dim3(pl::kBlock), 0, h->stream, h->ddm_cells,
                     h->ddm_nb, h->ddm_cell_nodes.p, h->ddm_have_P ? h->ddm_cell_P.p : h->ddm_cell_S.p,
                     h->ddm_have_P ? h->ddm_Pt.p : h->ddm_St.p, fx, np, h->dd_A.p);
  hipLaunchKernelGGL(pl::k_ddm_dense_unit, dim3((np + 255) / 256), dim3(256), 0, h->stream, n6, np, fx, h->dd_A.p);
  pl::dense_factor_inverse(h->dd_A.p, h->dd_Lf.p, h->dd_W.p, h->dd_Wt.p, h->dd_Dinv.p, np, np, h->dd_info.p,
                           h->dd_bw, h->stream);
  PL_HIP(hipGetLastError());
  int info[2] = {0, 0};
  PL_HIP(hipMemcpyAsync(info, h->dd_info.p, sizeof(info), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  h->dd_ready = (info[0] == 0);   // not positive definite (e.g. an indefinite surrogate matrix): caller falls back
  return PL_OK;
}

// precond = 5 on a strut-operator handle: P K P + (I - P) as a dense matrix from the BSR(6 x 6) blocks, factorised by the same
// device Cholesky as the DDM path's assembled Schur matrix (fp64 inverse factor); the PCG then converges in one or two
// iterations.  What PETSc's preonly / LU is to the reference (simulation_base.py:501-511), for lattices of a few hundred
// nodes - the sizes of the reference's own presets - where hundreds of PCG iterations cost more than the factorisation.
static int fem_factor_dense(pl_context *h) {
  const int64_t n6 = h->N * 6;
  const int np = (int)((n6 + pl::kNB - 1) / pl::kNB * pl::kNB);
  if (h->dd_n != np) {
    const size_t nn = (size_t)np * np;
    PL_HIP(h->dd_A.alloc(nn));
    PL_HIP(h->dd_Lf.alloc(nn));
    PL_HIP(h->dd_W.alloc(nn));
    PL_HIP(h->dd_Wt.alloc(nn));
    PL_HIP(h->dd_Dinv.alloc((size_t)np * pl::kNB));
    PL_HIP(h->dd_tv.alloc(np));
    PL_HIP(h->dd_info.alloc(2));
    PL_HIP(hipMemsetAsync(h->dd_W.p, 0, nn * sizeof(double), h->stream));
    PL_HIP(hipMemsetAsync(h->dd_Wt.p, 0, nn * sizeof(double), h->stream));
    h->dd_n = np;
    // block bandwidth of K in the device numbering (spatial tiles: a banded matrix)
    int64_t band = 0;
    for (int64_t i = 0; i < h->N; ++i)
      for (int64_t q = h->h_rowptr[i]; q < h->h_rowptr[i + 1]; ++q) band = std::max<int64_t>(band, std::llabs((long long)(h->h_col[q] - i)));
    h->dd_bw = (int)((6 * (band + 1) + pl::kNB - 1) / pl::kNB + 1);
  }
  h->dd_ready = false;
  int rc = ensure_bsr_buffers(h);
  if (rc) return rc;
  PL_HIP(hipMemsetAsync(h->dd_A.p, 0, (size_t)np * np * sizeof(double), h->stream));
  PL_HIP(hipMemsetAsync(h->dd_info.p, 0, 2 * sizeof(int), h->stream));
  rc = launch_bsr_fill(h, 1, h->stream);            // dolfinx's Dirichlet treatment: constrained rows / columns zero, unit diagonal
  if (rc) return rc;
  h->have_bsr = false;                              // (the caller's explicit matrix, if any, is refreshed by pl_assemble)
  hipLaunchKernelGGL(pl::k_bsr_to_dense, dim3(grid_for(h->nblk * 36)), dim3(pl::kBlock), 0, h->stream, h->N, h->nblk,
                     h->bsr_rowptr.p, h->bsr_col.p, (const double *)h->bsr_vals.p, np, h->dd_A.p);
  hipLaunchKernelGGL(pl::k_ddm_dense_unit, dim3((np + 255) / 256), dim3(256), 0, h->stream, n6, np, (const uint8_t *)nullptr,
                     h->dd_A.p);
  pl::dense_factor_inverse(h->dd_A.p, h->dd_Lf.p, h->dd_W.p, h->dd_Wt.p, h->dd_Dinv.p, np, np, h->dd_info.p, h->dd_bw,
                           h->stream);
  PL_HIP(hipGetLastError());
  int info[2] = {0, 0};
  PL_HIP(hipMemcpyAsync(info, h->dd_info.p, sizeof(info), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  h->dd_ready = (info[0] == 0);   // not positive definite (a mechanism): the solve falls back to Jacobi
  if (h->dd_ready) PL_HIP(hipMemsetAsync(h->dinv.p, 0, n6 * sizeof(double), h->stream));     // z comes from the dense solve
  return PL_OK;
}

void pl_destroy(pl_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->opt.device);
  (void)hipStreamSynchronize(h->stream);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  pl::dist_destroy(h->dist);
  if (h->pal_host_flags != h->pal_fallback_flags) (void)hipHostFree(h->pal_host_flags);
  if (h->cls_host_flag) (void)hipHostFree(h->cls_host_flag);
  delete h;
}

int pl_set_bc(pl_handle h, const uint8_t *fixed, const double *ubar, const double *f) {
  if (!valid(h) || !fixed) return fail(PL_ERR_ARG, "pl_set_bc: null argument");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t N = h->N;
  const size_t n6 = (size_t)N * 6;
  std::vector<uint8_t> fx(n6), bits(N);
  double *st = nullptr;          // ubar, then f, through the handle's pinned staging buffer
  if (int rcs = staging(h, &st)) return rcs;
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i) {
      const size_t src = 6 * (size_t)h->perm[i];
      uint8_t b = 0;
      for (int k = 0; k < 6; ++k) {
        const uint8_t v = fixed[src + k] ? 1 : 0;
        fx[6 * i + k] = v;
        b |= (uint8_t)(v << k);
        st[6 * i + k] = (ubar && v) ? ubar[src + k] : 0.0;
      }
      bits[i] = b;
    }
  }, 1 << 15);
  PL_HIP(hipMemcpyAsync(h->ubar.p, st, n6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  PL_HIP(hipMemcpyAsync(h->fixed.p, fx.data(), n6, hipMemcpyHostToDevice, h->stream));
  PL_HIP(hipMemcpyAsync(h->fixedbits.p, bits.data(), N, hipMemcpyHostToDevice, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i) {
      const size_t src = 6 * (size_t)h->perm[i];
      for (int k = 0; k < 6; ++k) st[6 * i + k] = f ? f[src + k] : 0.0;
    }
  }, 1 << 15);
  PL_HIP(hipMemcpyAsync(h->f.p, st, n6 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  // The same Dirichlet set as before (only prescribed values / loads changed - the columns of pl_schur, the adjoint solves of
  // a design loop): everything that depends on the mask (Jacobi inverse, coarse levels, eliminated nodes, dense factor) stands
  const bool same_mask = h->have_bc && !h->dist.active && h->h_fixedbits == bits;
  h->h_fixedbits = bits;
  if (same_mask) return PL_OK;
  h->have_bc = true;
  h->coarse.n_fix = -1;
  h->coarseL.n_fix = -1;
  {
    int rcs = select_condensed(h, bits);
    if (rcs) return rcs;
  }
  if (h->assembled && h->opkind == 1) {
    if (h->opt.precond >= 2 && h->opt.precond <= 4) {
      // the factorised G / the node blocks (and the dense level on top of them) were built for the old Dirichlet mask (and
      // dinv must stay 0 next to them): build them again
      h->assembled = false;
      h->dd_ready = false;
      h->dd_blocks = false;
      h->dd2_ready = false;
    } else {
      pl::launch_invert_diag(h->N * 6, h->diag.p, h->fixed.p, h->dinv.p, h->stream);
      PL_HIP(hipStreamSynchronize(h->stream));
    }
  } else if (h->assembled) {   // the Jacobi inverse and the coarse operator depend on the mask
    int rc = launch_diag(h, h->stream);
    if (rc) return rc;
    rc = finish_diag_dist(h);
    if (rc) return rc;
    rc = launch_local_mask(h);
    if (rc) return rc;
    rc = launch_tile_blocks(h, h->stream);
    if (rc) return rc;
    rc = build_coarse(h);
    if (rc) return rc;
    rc = launch_dinv32(h);
    if (rc) return rc;
    rc = agree_condensed(h);
    if (rc) return rc;
    rc = launch_condensed_blocks(h, h->stream);
    if (rc) return rc;
    PL_HIP(hipStreamSynchronize(h->stream));
    {
      const bool keep = h->pal_ready;        // (the record palette is untouched by a new mask)
      finish_condensed_classes(h);
      (void)keep;
    }
    if (h->bsr_with_bc) h->have_bsr = false;   // an explicit matrix built with the old mask is stale
    if (h->opt.precond == 5 && !h->dist.active) {   // the dense factor depends on the mask
      rc = fem_factor_dense(h);
      if (rc) return rc;
    }
  }
  return PL_OK;
}

int pl_set_periodic(pl_handle h, const int32_t *master) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_set_periodic: null handle");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_set_periodic: not available on a DDM handle");
  h->n_per_groups = 0;
  if (!master) return PL_OK;
  if (h->opt.precond != 1 || h->dist.active)
    return fail(PL_ERR_STATE, "pl_set_periodic: periodic constraints are served by the Jacobi PCG of a single-GPU handle (opts.precond = 1)");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t N = h->N;
  std::vector<int32_t> inv((size_t)N);
  for (int64_t i = 0; i < N; ++i) inv[(size_t)h->perm[i]] = (int32_t)i;          // caller node -> device node
  std::vector<int32_t> cnt((size_t)N, 0);
  for (int64_t i = 0; i < N; ++i) {
    const int32_t m = master[i];
    if (m < 0 || m >= N || master[m] != m) return fail(PL_ERR_ARG, "pl_set_periodic: master[i] must name a node that is its own master");
    cnt[(size_t)m]++;
  }
  std::vector<int32_t> ptr(1, 0), nodes, start((size_t)N, -1);
  for (int64_t m = 0; m < N; ++m)
    if (cnt[(size_t)m] >= 2) {
      start[(size_t)m] = (int32_t)ptr.size() - 1;
      ptr.push_back(ptr.back() + cnt[(size_t)m]);
    }
  nodes.resize((size_t)ptr.back());
  std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
  for (int64_t i = 0; i < N; ++i) {
    const int32_t g = start[(size_t)master[i]];
    if (g >= 0) nodes[(size_t)fill[(size_t)g]++] = inv[(size_t)i];
  }
  const int64_t ng = (int64_t)ptr.size() - 1;
  if (ng == 0) return PL_OK;
  PL_HIP(h->per_ptr.alloc(ptr.size()));
  PL_HIP(h->per_nodes.alloc(nodes.size()));
  PL_HIP(hipMemcpy(h->per_ptr.p, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->per_nodes.p, nodes.data(), nodes.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  h->n_per_groups = ng;
  return PL_OK;
}

int pl_update_radii(pl_handle h, const double *beam_radius) {
  if (!valid(h) || !beam_radius) return fail(PL_ERR_ARG, "pl_update_radii: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_update_radii: not available on a DDM handle");
  for (int64_t b = 0; b < h->B; ++b)
    if (!(beam_radius[b] > 0.0)) return fail(PL_ERR_ARG, "pl_update_radii: non-positive radius");
  PL_HIP(hipSetDevice(h->opt.device));
  double *stg = nullptr;
  int rc = stagingB(h, &stg);
  if (rc) return rc;
  pl::parallel_for(h->B, [&](int64_t b0, int64_t b1, unsigned) {
    for (int64_t b = b0; b < b1; ++b) stg[b] = beam_radius[h->bperm[b]];
  }, 1 << 16);
  PL_HIP(hipMemcpyAsync(h->radius.p, stg, h->B * sizeof(double), hipMemcpyHostToDevice, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  h->assembled = false;
  h->have_bsr = false;
  return PL_OK;
}

int pl_set_multiplicity(pl_handle h, const double *beam_mult) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_set_multiplicity: null handle");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_set_multiplicity: not available on a DDM handle");
  PL_HIP(hipSetDevice(h->opt.device));
  if (!beam_mult) {
    h->mult.release();
  } else {
    for (int64_t b = 0; b < h->B; ++b)
      if (!(beam_mult[b] > 0.0)) return fail(PL_ERR_ARG, "pl_set_multiplicity: non-positive multiplicity");
    std::vector<double> m(h->B);
    for (int64_t b = 0; b < h->B; ++b) m[b] = beam_mult[h->bperm[b]];
    if (!h->mult.p) PL_HIP(h->mult.alloc(h->B));
    PL_HIP(hipMemcpy(h->mult.p, m.data(), h->B * sizeof(double), hipMemcpyHostToDevice));
  }
  h->assembled = false;
  h->have_bsr = false;
  return PL_OK;
}

int pl_update_segments(pl_handle h, const double *seg_len, const int32_t *seg_nsub) {
  if (!valid(h) || !seg_len || !seg_nsub) return fail(PL_ERR_ARG, "pl_update_segments: null argument");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> sl((size_t)h->B * 3);
  std::vector<int32_t> sn((size_t)h->B * 3);
  for (int64_t b = 0; b < h->B; ++b)
    for (int k = 0; k < 3; ++k) {
      sl[3 * b + k] = seg_len[3 * (size_t)h->bperm[b] + k];
      sn[3 * b + k] = seg_nsub[3 * (size_t)h->bperm[b] + k];
    }
  PL_HIP(hipMemcpy(h->seg_len.p, sl.data(), 3 * h->B * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(h->seg_nsub.p, sn.data(), 3 * h->B * sizeof(int32_t), hipMemcpyHostToDevice));
  h->assembled = false;
  h->have_bsr = false;
  return PL_OK;
}

int pl_assemble(pl_handle h) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_assemble: null handle");
  PL_HIP(hipSetDevice(h->opt.device));
  if (h->opkind == 1) {   // DDM operator: nothing to build; plain CG as the reference, Jacobi, or the factorised matrix
    // (everything that changes what is built here - pl_set_bc with another Dirichlet set, pl_ddm_update_matrices,
    //  pl_ddm_set_preconditioner, pl_ddm_set_geometry - clears `assembled`: a second call on an unchanged handle, e.g. solve_DDM
    //  for another load case, has nothing to do.  32^3 cells: 1.6 of the 9.5 ms of a repeated solve_DDM)
    if (h->assembled && (h->opt.precond == 0 || (h->opt.precond == 1 && !h->dd_ready && !h->dd_blocks) || h->dd_ready || h->dd_blocks)) {
      h->ms_assembly = 0.0;
      return PL_OK;
    }
    h->dd_ready = false;
    h->dd_blocks = false;
    h->dd2_ready = false;
    if (h->opt.precond == 4 && !h->dd2_plan)
      return fail(PL_ERR_STATE, "pl_assemble: precond = 4 on a DDM handle needs the node positions (pl_ddm_set_geometry)");
    if (h->opt.precond == 3 || h->opt.precond == 4) {      // node-block Jacobi: the 6 x 6 diagonal blocks of the assembled matrix, inverted
      PL_HIP(hipEventRecord(h->ev0, h->stream));
      if (!h->dd_B.p) PL_HIP(h->dd_B.alloc((size_t)h->N * 36));
      PL_HIP(hipMemsetAsync(h->dd_B.p, 0, (size_t)h->N * 36 * sizeof(double), h->stream));
      const int64_t ne = (int64_t)h->ddm_cells * h->ddm_nb * 36;
      hipLaunchKernelGGL(pl::k_ddm_node_blocks, dim3(grid_for(ne)), dim3(pl::kBlock), 0, h->stream, h->ddm_cells, h->ddm_nb,
                         h->ddm_cell_nodes.p, h->ddm_have_P ? h->ddm_cell_P.p : h->ddm_cell_S.p,
                         h->ddm_have_P ? h->ddm_Pt.p : h->ddm_St.p, h->dd_B.p);
      hipLaunchKernelGGL(pl::k_ddm_node_blocks_invert, dim3(grid_for(h->N)), dim3(pl::kBlock), 0, h->stream, h->N,
                         h->have_bc ? (const uint8_t *)h->fixed.p : (const uint8_t *)nullptr, h->dd_B.p);
      PL_HIP(hipMemsetAsync(h->dinv.p, 0, h->N * 6 * sizeof(double), h->stream));      // z comes from the node blocks
      int info2[2] = {0, 0};
      if (h->opt.precond == 4) {    // + the dense level: A_c = Z^T P G P Z cell by cell, Cholesky + inverse factor (pl_dense.h)
        pl::Coarse &cs = h->dd2;
        const int n = cs.ncp, m = 6 * h->ddm_nb;
        PL_HIP(hipMemsetAsync(cs.Ac, 0, (size_t)n * n * sizeof(double), h->stream));
        PL_HIP(hipMemsetAsync(cs.info, 0, 2 * sizeof(int), h->stream));
        hipLaunchKernelGGL(pl::k_ddm_coarse_cells, dim3((unsigned)h->ddm_cells), dim3(pl::kWave),
                           (size_t)2 * m * pl::kDdmModes * sizeof(double), h->stream, h->ddm_cells, h->ddm_nb,
                           h->ddm_cell_nodes.p, h->ddm_have_P ? h->ddm_cell_P.p : h->ddm_cell_S.p,
                           h->ddm_have_P ? h->ddm_Pt.p : h->ddm_St.p, h->dd2_agg.p, h->dd2_cen.p, h->dd2_xyz.p,
                           h->have_bc ? (const uint8_t *)h->fixed.p : (const uint8_t *)nullptr, n, cs.Ac);
        hipLaunchKernelGGL(pl::k_coarse_regularize, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, cs.Ac);
        pl::TrtriPhases ph;          // the inverse factor in row ranges on a second stream, behind the chain (as pl_assembly.h)
        ph.rows = 3;
        ph.stream = h->side2;
        ph.ev_go = h->ev_t0;
        ph.ev_done = h->ev_t1;
        pl::coarse_factor(cs, n, h->stream, nullptr, (unsigned *)nullptr, ph);
        cs.ainv_ready = false;
        if (cs.Ainv) {
          const long tiles = (long)(n / 32) * (n / 32 + 1) / 2;
          if (cs.w16)
            hipLaunchKernelGGL(pl::k_dense_explicit_inverse<pl::bf16_t>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0,
                               h->stream, n, reinterpret_cast<const pl::bf16_t *>(cs.W), n, cs.Ainv);
          else
            hipLaunchKernelGGL(pl::k_dense_explicit_inverse<float>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, h->stream,
                               n, (const float *)cs.W, n, cs.Ainv);
          cs.ainv_ready = true;
        }
        PL_HIP(hipMemcpyAsync(info2, cs.info, sizeof(info2), hipMemcpyDeviceToHost, h->stream));
      }
      PL_HIP(hipEventRecord(h->ev1, h->stream));
      PL_HIP(hipEventSynchronize(h->ev1));
      PL_HIP(hipGetLastError());
      float ms = 0.f;
      PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
      h->ms_assembly = ms;
      h->dd2_ready = h->opt.precond == 4 && info2[0] == 0;     // (A_c not positive definite: the node blocks alone; precond_used = 3)
      h->dd_blocks = true;
      h->assembled = true;
      return PL_OK;
    }
    if (h->opt.precond == 2) {
      PL_HIP(hipEventRecord(h->ev0, h->stream));
      int rcp = ddm_factor_preconditioner(h);
      if (rcp) return rcp;
      if (h->dd_ready) {
        PL_HIP(hipMemsetAsync(h->dinv.p, 0, h->N * 6 * sizeof(double), h->stream));   // z comes from the dense solve
        PL_HIP(hipEventRecord(h->ev1, h->stream));
        PL_HIP(hipEventSynchronize(h->ev1));
        float ms = 0.f;
        PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        h->ms_assembly = ms;
        h->assembled = true;
        return PL_OK;
      }
      // G is not positive definite (the reference falls back from LU to ILU when its factorisation fails,
      // lattice_sim.py:1406-1413): Jacobi on the assembled diagonal; pl_stats_t.precond_used tells the caller
    }
    if (h->opt.precond >= 1) {
      PL_HIP(hipMemsetAsync(h->diag.p, 0, h->N * 6 * sizeof(double), h->stream));
      const int64_t m = (int64_t)h->ddm_cells * h->ddm_nb * 6;
      hipLaunchKernelGGL(pl::k_ddm_diag, dim3(grid_for(m)), dim3(pl::kBlock), 0, h->stream, h->ddm_cells, h->ddm_nb,
                         h->ddm_cell_nodes.p, h->ddm_cell_S.p, h->ddm_St.p, h->diag.p);
    } else {
      pl::launch_fill(h->N * 6, 1.0, h->diag.p, h->stream);
    }
    pl::launch_invert_diag(h->N * 6, h->diag.p, h->have_bc ? h->fixed.p : nullptr, h->dinv.p, h->stream);
    PL_HIP(hipStreamSynchronize(h->stream));
    h->ms_assembly = 0.0;
    h->assembled = true;
    return PL_OK;
  }
  int rc = agree_condensed(h);      // (multi-GPU, first assembly after a pl_set_bc: a host round trip, outside the timing)
  if (rc) return rc;
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  rc = launch_records(h);
  if (rc) return rc;
  rc = launch_local_mask(h);
  if (rc) return rc;
  // fork: everything that only streams the records runs on the side stream while the main stream walks the
  // latency-bound chain of the coarse factorisation
  PL_HIP(hipEventRecord(h->ev_fork, h->stream));
  PL_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  rc = launch_palette(h, h->side);
  if (rc) return rc;
  rc = launch_diag(h, h->dist.active ? h->stream : h->side);
  if (rc) return rc;
  rc = finish_diag_dist(h);
  if (rc) return rc;
  rc = launch_tile_blocks(h, h->side);
  if (rc) return rc;
  rc = launch_condensed_blocks(h, h->side);
  if (rc) return rc;
  const bool refresh_bsr = h->want_bsr && (!h->bsr_with_bc || h->have_bc);
  // The explicit K is the one bulk item of the assembly (2.5 GB of traffic): next to the Cholesky chain it slows every
  // link of that latency-bound chain (44 -> 57 us), so it starts when the chain's last link is queued and runs beside
  // the single-launch inverse factor instead.
  int rc_fill = PL_OK;
  bool fill_queued = false;
  // How much of the fill fits beside the chain: the masked stream fills 7 - 9 M blocks per millisecond (50^3 Octet: 6.6 M blocks in
  // 0.72 ms; 100^3 BCC: 18 M in 2.6 ms), a link of the chain takes ~40 us.  What does not fit (200 x 200 x 50 BCC + Octet:
  // 138 M blocks against a 3.8-ms chain - all of it on the masked half of the chip cost 7 ms) runs behind the chain on the
  // plain side stream, on the whole chip, as until round 4.
  int64_t split = 0;            // slices [0, split) beside the chain
  if (refresh_bsr && h->side_cu && h->coarse.enabled && h->have_bc) {
    // (PL_CHAIN_LINK_MS / PL_FILL_BLOCKS_PER_MS: the two measured rates, for A/B runs on another box)
    static const double kChainLinkMs = [] { const char *e = std::getenv("PL_CHAIN_LINK_MS"); return e ? std::atof(e) : 0.04; }();
    static const double kFillBlocksPerMs = [] { const char *e = std::getenv("PL_FILL_BLOCKS_PER_MS"); return e ? std::atof(e) : 8.5e6; }();
    const double chain_ms = kChainLinkMs * std::max(0, h->coarse.ncp / pl::kNB - 1);
    const double fit = chain_ms * kFillBlocksPerMs / std::max<double>(1.0, (double)h->nblk);
    split = fit >= 1.0 ? h->n_slices : (int64_t)(fit * (double)h->n_slices);
    if (split < h->n_slices / 16) split = 0;         // (not worth a launch)
  }
  bool part1_queued = false;
  auto queue_fill = [&](int phase) {
    if (!refresh_bsr || fill_queued) return;
    if (phase == 0) {              // head of the chain: the part that fits beside it, on the stream that leaves it CUs
      if (split <= 0 || part1_queued) return;
      part1_queued = true;
      if (hipEventRecord(h->ev_chol, h->stream) != hipSuccess || hipStreamWaitEvent(h->side_cu, h->ev_chol, 0) != hipSuccess ||
          hipStreamWaitEvent(h->side_cu, h->ev_join, 0) != hipSuccess) {
        rc_fill = fail(PL_ERR_HIP, "pl_assemble: could not order the BSR fill beside the factorisation");
        return;
      }
      rc_fill = launch_bsr_fill(h, h->bsr_with_bc, h->side_cu, 0, split);
      if (hipEventRecord(h->ev_fill1, h->side_cu) != hipSuccess) rc_fill = fail(PL_ERR_HIP, "pl_assemble: event record failed");
      return;
    }
    fill_queued = true;            // behind the chain: the rest (everything without a masked stream), on the whole chip
    if (hipEventRecord(h->ev_chol, h->stream) != hipSuccess || hipStreamWaitEvent(h->side, h->ev_chol, 0) != hipSuccess) {
      rc_fill = fail(PL_ERR_HIP, "pl_assemble: could not order the BSR fill behind the factorisation");
      return;
    }
    if (!rc_fill) rc_fill = launch_bsr_fill(h, h->bsr_with_bc, h->side, part1_queued ? split : 0, -1);
    if (part1_queued && hipStreamWaitEvent(h->side, h->ev_fill1, 0) != hipSuccess)
      rc_fill = fail(PL_ERR_HIP, "pl_assemble: event wait failed");
    if (hipEventRecord(h->ev_join, h->side) != hipSuccess) rc_fill = fail(PL_ERR_HIP, "pl_assemble: event record failed");
  };
  // (round 3, tried: the fill queued HERE, beside the tile-block front of the assembly instead of behind the chain's last
  // link: assembly 2.27 -> 2.46 ms in two alternating pairs of runs - it slows the front and the first links)
  PL_HIP(hipEventRecord(h->ev_join, h->side));
  h->dd_ready = false;
  if (h->opt.precond == 5 && h->have_bc && !h->dist.active) {
    PL_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));       // (the Jacobi diagonal first: the fallback, and dinv is cleared after)
    rc = fem_factor_dense(h);
    if (rc) return rc;
  }
  rc = build_coarse(h, h->coarse.enabled && h->have_bc ? std::function<void(int)>(queue_fill) : std::function<void(int)>());
  if (rc) return rc;
  queue_fill(1);                                 // (no dense level: queue it now)
  if (rc_fill) return rc_fill;
  PL_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  rc = launch_dinv32(h);
  if (rc) return rc;
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  finish_palette(h);
  finish_condensed_classes(h);
  // the factorisation has consumed A_c: zero it now, behind the caller's back, instead of at the head of the next
  // assembly's critical chain (34 MB; it would sit in front of the coarse assembly there)
  for (pl::Coarse *cs : {&h->coarse, &h->coarseL})
    if (cs->ready && cs->Ac) {
      PL_HIP(hipMemsetAsync(cs->Ac, 0, (size_t)cs->ncp * cs->ncp * sizeof(double), h->stream));
      cs->ac_clean = true;
    }
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->ms_assembly = ms;
  h->assembled = true;
  h->have_bsr = refresh_bsr;
  return PL_OK;
}

int pl_assemble_bsr(pl_handle h, int with_bc, int64_t *n_block_rows, int64_t *n_blocks) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_assemble_bsr: null handle");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_assemble_bsr: not available on a DDM handle");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_assemble_bsr: call pl_assemble first");
  if (with_bc && !h->have_bc) return fail(PL_ERR_STATE, "pl_assemble_bsr: with_bc needs pl_set_bc");
  PL_HIP(hipSetDevice(h->opt.device));
  if (int rcb = ensure_bsr_buffers(h)) return rcb;
  const bool fresh = h->have_bsr && h->want_bsr && h->bsr_with_bc == (with_bc ? 1 : 0);
  h->want_bsr = true;
  h->bsr_with_bc = with_bc ? 1 : 0;
  if (fresh) {   // pl_assemble already rebuilt it (overlapped with the coarse factorisation)
    if (n_block_rows) *n_block_rows = h->N;
    if (n_blocks) *n_blocks = h->nblk;
    return PL_OK;
  }
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  int rc = launch_bsr_fill(h, with_bc, h->stream);
  if (rc) return rc;
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->ms_assembly += ms;
  h->have_bsr = true;
  if (n_block_rows) *n_block_rows = h->N;
  if (n_blocks) *n_blocks = h->nblk;
  return PL_OK;
}

int pl_get_bsr(pl_handle h, int64_t *rowptr, int32_t *colidx, double *vals) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_get_bsr: null handle");
  if (!h->have_bsr) return fail(PL_ERR_STATE, "pl_get_bsr: call pl_assemble_bsr first");
  PL_HIP(hipSetDevice(h->opt.device));
  // returned in CALLER numbering: row i of the device matrix is caller node perm[i]
  std::vector<double> v((size_t)h->nblk * 36);
  PL_HIP(hipMemcpy(v.data(), h->bsr_vals.p, v.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (!h->reordered) {
    if (rowptr) std::memcpy(rowptr, h->h_rowptr.data(), (h->N + 1) * sizeof(int64_t));
    if (colidx) std::memcpy(colidx, h->h_col.data(), h->nblk * sizeof(int32_t));
    if (vals) std::memcpy(vals, v.data(), v.size() * sizeof(double));
    return PL_OK;
  }
  // permuted: rebuild rows in caller order, columns re-sorted
  std::vector<int64_t> rp(h->N + 1, 0);
  for (int64_t ci = 0; ci < h->N; ++ci) {
    const int64_t di = h->iperm[ci];
    rp[ci + 1] = rp[ci] + (h->h_rowptr[di + 1] - h->h_rowptr[di]);
  }
  if (rowptr) std::memcpy(rowptr, rp.data(), (h->N + 1) * sizeof(int64_t));
  std::vector<std::pair<int32_t, int64_t>> row;
  for (int64_t ci = 0; ci < h->N; ++ci) {
    const int64_t di = h->iperm[ci];
    row.clear();
    for (int64_t p = h->h_rowptr[di]; p < h->h_rowptr[di + 1]; ++p) row.push_back({h->perm[h->h_col[p]], p});
    std::stable_sort(row.begin(), row.end(), [](auto &l, auto &r) { return l.first < r.first; });
    for (size_t k = 0; k < row.size(); ++k) {
      if (colidx) colidx[rp[ci] + k] = row[k].first;
      if (vals) std::memcpy(vals + 36 * (rp[ci] + k), &v[36 * row[k].second], 36 * sizeof(double));
    }
  }
  return PL_OK;
}

static int spmv_common(pl_handle h, const double *x, double *y, bool masked, const char *who) {
  if (!valid(h) || !x || !y) return fail(PL_ERR_ARG, std::string(who) + ": null argument");
  if (!h->assembled) return fail(PL_ERR_STATE, std::string(who) + ": call pl_assemble first");
  if (masked && !h->have_bc) return fail(PL_ERR_STATE, std::string(who) + ": call pl_set_bc first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, x, h->tmp.p, stage);
  if (rc) return rc;
  if (masked) {
    // enforce the contract "x is zero on fixed dofs" for arbitrary caller input
    hipLaunchKernelGGL(pl::k_mask_dot, dim3(grid_stream(h->N * 6)), dim3(pl::kBlock), 0, h->stream, h->N * 6,
                       h->fixed.p, h->tmp.p, h->tmp.p, (double *)nullptr);
  }
  rc = launch_spmv(h, h->tmp.p, h->tmp2.p, masked, nullptr);
  if (rc) return rc;
  return download6(h, h->tmp2.p, y);
}

int pl_spmv(pl_handle h, const double *x, double *y) { return spmv_common(h, x, y, false, "pl_spmv"); }
int pl_spmv_free(pl_handle h, const double *x, double *y) { return spmv_common(h, x, y, true, "pl_spmv_free"); }

int pl_spmv_bsr(pl_handle h, const double *x, double *y) {
  if (!valid(h) || !x || !y) return fail(PL_ERR_ARG, "pl_spmv_bsr: null argument");
  if (!h->have_bsr) return fail(PL_ERR_STATE, "pl_spmv_bsr: call pl_assemble_bsr first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, x, h->tmp.p, stage);
  if (rc) return rc;
  hipLaunchKernelGGL(pl::k_bsr_spmv, dim3(grid_for(h->N)), dim3(pl::kBlock), 0, h->stream, h->N, h->bsr_rowptr.p,
                     h->bsr_col.p, h->bsr_vals.p, h->tmp.p, h->tmp2.p);
  PL_HIP(hipGetLastError());
  return download6(h, h->tmp2.p, y);
}

int pl_solve(pl_handle h, double rtol, int32_t max_iter, double *u, pl_stats_t *stats) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_solve: null handle");   // u == NULL: leave the solution on the device
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_solve: call pl_assemble first");
  if (!h->have_bc) return fail(PL_ERR_STATE, "pl_solve: call pl_set_bc first");
  if (!(rtol > 0.0) || max_iter <= 0) return fail(PL_ERR_ARG, "pl_solve: rtol and max_iter must be positive");
  if (stats && stats->struct_size != sizeof(pl_stats_t))
    return fail(PL_ERR_ARG, "pl_solve: stats->struct_size is " + std::to_string(stats->struct_size) + ", this library's pl_stats_t has " +
                                std::to_string(sizeof(pl_stats_t)) + " bytes (set it to sizeof(pl_stats_t) before the call)");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t n6 = h->N * 6;
  pl_stats_t st{};
  st.struct_size = (uint32_t)sizeof(pl_stats_t);
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  // lifting: tmp = K ubar (ubar is zero on free dofs)
  int rc = launch_spmv(h, h->ubar.p, h->tmp.p, false, nullptr);
  if (rc) return rc;
  if (h->n_per_groups > 0 && (h->coarse.ready || ref_cg(h) || h->dist.active))
    return fail(PL_ERR_STATE, "pl_solve: periodic constraints (pl_set_periodic) need the plain Jacobi PCG");
  // fp32 solver modes need the multi-level preconditioner on the tile kernel; anything else runs the fp64 PCG
  const bool mp = mp_applies(h);
  solver_plan(h);
  const bool cg1 = !mp && cg1_applies(h);
  st.cg_form_used = cg1 ? 1.0 : 0.0;
  if (cg1) rc = pcg_solve_cg1(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  else if (mp && h->opt.precision == 1) rc = pcg_solve_mp_t<float>(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  else if (mp) rc = pcg_solve_mp_t<double>(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  else rc = pcg_solve(h, h->f.p, h->tmp.p, rtol, max_iter, &st);
  if (rc) return rc;
  st.precision_used = mp ? (double)h->opt.precision : 0.0;
  st.kp_form = (double)kp_form_of(h);
  st.comm_world = h->dist.active ? (double)h->dist.comm_count : 0.0;
  st.comm_rank = h->dist.active ? (double)h->dist.comm_user_rank : 0.0;
  st.condensed_nodes = h->cond_use ? (double)h->n_cond : 0.0;
  if (st.converged) st.info = 0.0;
  else if (st.info != 2.0) st.info = 1.0;     // precision mode the solve ran in
  if (!h->usol.p) PL_HIP(h->usol.alloc((size_t)n6));
  h->usol_valid = false;
  hipLaunchKernelGGL(pl::k_compose_solution, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, h->stream, n6, h->fixed.p,
                     h->ubar.p, h->x.p, h->usol.p);
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  st.ms_solve = ms;
  st.ms_assembly = h->ms_assembly;
  st.precond_used = h->opkind == 1 ? (h->dd_ready ? 2 : h->dd2_ready ? 4 : h->dd_blocks ? 3 : h->opt.precond >= 1 ? 1 : 0)
                                   : (h->coarse.ready ? h->opt.precond : (h->dd_ready ? 5 : 1));
  h->usol_valid = true;
  if (u) {
    rc = download6(h, h->usol.p, u);
    if (rc) return rc;
  }
  h->last = st;
  if (stats) *stats = st;
  if (!st.converged) return fail(PL_ERR_NOCONV, "pl_solve: PCG did not reach rtol within max_iter");
  return PL_OK;
}

int pl_reactions(pl_handle h, const double *u, double *R) { return spmv_common(h, u, R, false, "pl_reactions"); }

int pl_sens(pl_handle h, const double *u, const double *lam, double *dCdr) {
  if (!valid(h) || !dCdr) return fail(PL_ERR_ARG, "pl_sens: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_sens: not available on a DDM handle");
  // u == NULL: the solution of the last pl_solve, still on the device (a design loop evaluates the sensitivities of the
  // displacement field it has just computed: no 6N-vector upload)
  if (!u && !(h->usol.p && h->usol_valid)) return fail(PL_ERR_STATE, "pl_sens: u = NULL needs a pl_solve on this handle first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = PL_OK;
  const double *u_dev = h->usol.p;
  if (u) {
    rc = upload6(h, u, h->tmp.p, stage);
    if (rc) return rc;
    u_dev = h->tmp.p;
  }
  const double *lam_dev = u_dev;
  if (lam) {
    rc = upload6(h, lam, h->tmp2.p, stage);
    if (rc) return rc;
    lam_dev = h->tmp2.p;
  }
  if (!h->sens_out.p) PL_HIP(h->sens_out.alloc(h->B));
  hipLaunchKernelGGL(pl::k_sens, dim3(grid_for(h->B)), dim3(pl::kBlock), 0, h->stream, h->B, h->xyz.p, h->conn.p,
                     h->radius.p, h->seg_len.p, h->seg_nsub.p, h->mult.p, h->mat, u_dev, lam_dev, h->sens_out.p);
  PL_HIP(hipGetLastError());
  double *stg = nullptr;
  rc = stagingB(h, &stg);
  if (rc) return rc;
  PL_HIP(hipMemcpyAsync(stg, h->sens_out.p, h->B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  pl::parallel_for(h->B, [&](int64_t b0, int64_t b1, unsigned) {
    for (int64_t b = b0; b < b1; ++b) dCdr[h->bperm[b]] = stg[b];
  }, 1 << 16);
  return PL_OK;
}

int pl_node_mod(pl_handle h, const double *u, double *out) {
  if (!valid(h) || !u || !out) return fail(PL_ERR_ARG, "pl_node_mod: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_node_mod: not available on a DDM handle");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_node_mod: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, u, h->tmp.p, stage);
  if (rc) return rc;
  DevBuf<double> dev;
  PL_HIP(dev.alloc((size_t)h->B * 12));
  hipLaunchKernelGGL(pl::k_node_mod, dim3(grid_for(h->B)), dim3(pl::kBlock), 0, h->stream, h->B, h->xyz.p, h->conn.p,
                     h->radius.p, h->seg_len.p, h->seg_nsub.p, h->mult.p, h->mat, h->rec.p, h->tmp.p, dev.p);
  PL_HIP(hipGetLastError());
  std::vector<double> tmp((size_t)h->B * 12);
  PL_HIP(hipMemcpyAsync(tmp.data(), dev.p, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  for (int64_t b = 0; b < h->B; ++b) std::memcpy(out + 12 * (size_t)h->bperm[b], &tmp[12 * (size_t)b], 12 * sizeof(double));
  return PL_OK;
}

int pl_energy(pl_handle h, const double *u, double *energy) {
  if (!valid(h) || !u || !energy) return fail(PL_ERR_ARG, "pl_energy: null argument");
  if (h->opkind != 0) return fail(PL_ERR_STATE, "pl_energy: not available on a DDM handle");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_energy: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<double> stage;
  int rc = upload6(h, u, h->tmp.p, stage);
  if (rc) return rc;
  double *aux = h->scal.p + pl::S_AUX * pl::kSlots;
  double h_aux[pl::kSlots];
  PL_HIP(hipMemsetAsync(aux, 0, sizeof(h_aux), h->stream));
  hipLaunchKernelGGL(pl::k_energy, dim3(grid_for(h->B)), dim3(pl::kBlock), 0, h->stream, h->B, h->conn.p, h->rec.p,
                     h->tmp.p, aux);
  PL_HIP(hipGetLastError());
  PL_HIP(hipMemcpyAsync(h_aux, aux, sizeof(h_aux), hipMemcpyDeviceToHost, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  *energy = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) *energy += h_aux[k];
  return PL_OK;
}

int pl_schur(pl_handle h, const int32_t *boundary_nodes, int32_t nb, double rtol, int32_t max_iter, double *S) {
  if (!valid(h) || !boundary_nodes || !S || nb <= 0) return fail(PL_ERR_ARG, "pl_schur: bad argument");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_schur: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t N = h->N;
  const size_t n6 = (size_t)N * 6;
  for (int i = 0; i < nb; ++i)
    if (boundary_nodes[i] < 0 || boundary_nodes[i] >= N) return fail(PL_ERR_ARG, "pl_schur: node index out of range");
  // Column j of S = reaction on the boundary dofs when boundary dof j = 1, the other boundary dofs = 0 and the
  // interior is in equilibrium: exactly S = K_BB - K_BI K_II^-1 K_IB.
  std::vector<uint8_t> fixed(n6, 0);
  for (int i = 0; i < nb; ++i)
    for (int k = 0; k < 6; ++k) fixed[6 * (size_t)boundary_nodes[i] + k] = 1;
  std::vector<double> ubar(n6, 0.0), u(n6), R(n6);
  const int m = nb * 6;
  for (int j = 0; j < m; ++j) {
    const size_t dofj = 6 * (size_t)boundary_nodes[j / 6] + (j % 6);
    ubar[dofj] = 1.0;
    int rc = pl_set_bc(h, fixed.data(), ubar.data(), nullptr);
    if (rc) return rc;
    pl_stats_t st;
    st.struct_size = (uint32_t)sizeof(pl_stats_t);
    rc = pl_solve(h, rtol, max_iter, u.data(), &st);
    if (rc) return rc;
    rc = pl_reactions(h, u.data(), R.data());
    if (rc) return rc;
    for (int i = 0; i < m; ++i) S[(size_t)i * m + j] = R[6 * (size_t)boundary_nodes[i / 6] + (i % 6)];
    ubar[dofj] = 0.0;
  }
  return PL_OK;
}

int pl_get_records(pl_handle h, double *rec) {
  if (!valid(h) || !rec) return fail(PL_ERR_ARG, "pl_get_records: null argument");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_get_records: call pl_assemble first");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<pl::Record> tmp(h->B);
  PL_HIP(hipMemcpy(tmp.data(), h->rec.p, h->B * sizeof(pl::Record), hipMemcpyDeviceToHost));
  for (int64_t b = 0; b < h->B; ++b) std::memcpy(rec + 8 * (size_t)h->bperm[b], &tmp[b], sizeof(pl::Record));
  return PL_OK;
}

int pl_algorithmic_bytes(pl_handle h, double *out3) {
  if (!valid(h) || !out3) return fail(PL_ERR_ARG, "pl_algorithmic_bytes: null argument");
  // SURVEY.md 8(d) with the storage widths the next pl_solve really uses: records are fp64 in every mode (w = 8); the PCG
  // vectors are fp32-stored in precision = 1 (x, r, p, K*p: wv = 4) and, for p and K*p only, in precision = 2.
  const double w = 8.0, B = (double)h->B, N = (double)h->N;
  const bool mp = h->assembled && mp_applies(h);
  const double wv = mp ? 4.0 : 8.0;                                    // p and K*p
  const double wx = (mp && h->opt.precision == 1) ? 4.0 : 8.0;         // x, r (and the other vector passes)
  out3[0] = B * (8.0 + 8.0 * w) + N * 6.0 * wv * 2.0;       // bytes_spmv
  out3[1] = out3[0] + 10.0 * (6.0 * N * wx);                // bytes_pcg_iter
  out3[2] = B * (8.0 + 8.0 * w) + (N + 2.0 * B) * (36.0 * w + 4.0);   // bytes_assembly_bsr (the explicit K is fp64)
  return PL_OK;
}

int pl_forget_history(pl_handle h) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_forget_history: null handle");
  h->last_iterations = 0;       // the next solve looks at the residual history every 32 iterations again
  h->xprev_valid = false;       // and starts from zero even with opts.warm_start
  h->xprev2_valid = false;
  h->xprev3_valid = false;
  h->gh_count = 0;
  return PL_OK;
}

int pl_time_kernel(pl_handle h, int which, int reps, double *avg_ms) {
  if (!valid(h) || !avg_ms || reps <= 0) return fail(PL_ERR_ARG, "pl_time_kernel: bad argument");
  if (h->opkind != 0 && which != 0 && which != 3 && which != 10 && which != 11) return fail(PL_ERR_STATE, "pl_time_kernel: DDM handles time K*p / PCG only");
  if (!h->assembled) return fail(PL_ERR_STATE, "pl_time_kernel: call pl_assemble first");
  if ((which == 0 || which == 3 || which == 10 || which == 11) && !h->have_bc) return fail(PL_ERR_STATE, "pl_time_kernel: call pl_set_bc first");
  if ((which == 2 || which == 4) && !h->have_bsr) return fail(PL_ERR_STATE, "pl_time_kernel: needs pl_assemble_bsr");
  PL_HIP(hipSetDevice(h->opt.device));
  const int64_t n6 = h->N * 6;
  int rc = ensure_hist(h, reps + 1);
  if (rc) return rc;
  solver_plan(h);   // which == 3 times the iteration the next pl_solve would run, whatever was called before
  h->stop_use = false;   // (a finished solve has stopped itself on the device: time the iteration, not its early return)
  if (h->small_use && (which == 3 || which == 11)) {     // short form: its buffers (zeroed operands: the timing does not care)
    rc = small_prepare(h);
    if (rc) return rc;
  }
  // a well-defined operand: p = dinv (free dofs) -> nonzero everywhere that matters
  PL_HIP(hipMemcpyAsync(h->p.p, h->dinv.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (which >= 7 && which <= 11) {   // the same operand, fp32-stored, in the z buffer; zeroed fp32 x / r in tmp
    hipLaunchKernelGGL(pl::k_to_float, dim3(grid_for(n6)), dim3(pl::kBlock), 0, h->stream, n6, h->dinv.p,
                       reinterpret_cast<float *>(h->z.p));
    PL_HIP(hipMemsetAsync(h->tmp.p, 0, n6 * sizeof(double), h->stream));
  }
  auto one = [&](int k) -> int {
    switch (which) {
      case 0: return launch_spmv(h, h->p.p, h->Ap.p, true, h->scal.p + pl::S_PAP * pl::kSlots);
      case 1: {
        int r1 = launch_records(h);
        return r1 ? r1 : build_palette(h);
      }
      case 2: return launch_bsr_fill(h, 0, h->stream);
      case 3: return pcg_iteration(h, k);
      case 4:
        hipLaunchKernelGGL(pl::k_bsr_spmv, dim3(grid_for(h->N)), dim3(pl::kBlock), 0, h->stream, h->N,
                           h->bsr_rowptr.p, h->bsr_col.p, h->bsr_vals.p, h->p.p, h->Ap.p);
        return PL_OK;
      case 5:   // multi-GPU: the interface all-reduce of a K*p alone (pack, RCCL, unpack) - collective call
        if (!h->dist.active) return fail(PL_ERR_STATE, "pl_time_kernel: 5/6 need pl_dist_init");
        return pl::dist_sum_shared(h->dist, h->Ap.p, h->stream, h->scal.p + pl::S_PAP * pl::kSlots, pl::kSlots)
                   ? fail(PL_ERR_HIP, "RCCL all-reduce failed") : PL_OK;
      case 6:   // multi-GPU: the coarse-residual all-reduce alone - collective call
        if (!h->dist.active || !h->coarse.ready) return fail(PL_ERR_STATE, "pl_time_kernel: 6 needs a coarse level");
        return pl::dist_sum_scalars(h->dist, h->coarse.tv, h->coarse.ncp, h->stream)
                   ? fail(PL_ERR_HIP, "RCCL all-reduce failed") : PL_OK;
      case 7:   // K*p on fp32-stored vectors (the operator of opts.precision = 1 / 2)
      case 8:   // one whole iteration of the fp32 inner PCG (precision = 1)
      case 9: { // one whole iteration of the mixed PCG (precision = 2: p, K*p fp32; x, r fp64)
        if (!(h->coarse.ready && h->tile.ready && choose_kernel(h) == 3))
          return fail(PL_ERR_STATE, "pl_time_kernel: 7/8/9 need the multi-level PCG on the tile kernel");
        float *p32 = reinterpret_cast<float *>(h->z.p), *Ap32 = p32 + n6;
        double *cur = h->scal.p + (k & 1) * pl::S_COUNT * pl::kSlots, *nxt = h->scal.p + ((k + 1) & 1) * pl::S_COUNT * pl::kSlots;
        int r7 = launch_spmv_f32(h, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots);
        if (r7 || which == 7) return r7;
        if (which == 8) {
          float *x32 = reinterpret_cast<float *>(h->tmp.p), *r32 = x32 + n6;
          return pcg_tail_coarse_t<float, float>(h, cur, nxt, k, p32, (const float *)Ap32, x32, r32);
        }
        return pcg_tail_coarse_t<float, double>(h, cur, nxt, k, p32, (const float *)Ap32, h->x.p, h->r.p);
      }
      case 10: {   // the operator exactly as the next pl_solve applies it: both passes under node elimination, fp32-stored
                   // operands in the fp32 solver modes
        if (mp_applies(h)) {
          float *p32 = reinterpret_cast<float *>(h->z.p), *Ap32 = p32 + n6;
          if (h->cond_use) {
            int r0 = launch_spmv_f32(h, p32, p32, false, nullptr, nullptr, pl::kEndsCondensedSolve);
            return r0 ? r0 : launch_spmv_f32(h, p32, Ap32, true, h->scal.p + pl::S_PAP * pl::kSlots, h->maskC.p, pl::kEndsOthers);
          }
          return launch_spmv_f32(h, p32, Ap32, true, h->scal.p + pl::S_PAP * pl::kSlots);
        }
        if (h->cond_use) {
          int r0 = launch_spmv(h, h->p.p, h->p.p, false, nullptr, nullptr, pl::kEndsCondensedSolve);
          return r0 ? r0 : launch_spmv(h, h->p.p, h->Ap.p, true, h->scal.p + pl::S_PAP * pl::kSlots, h->maskC.p, pl::kEndsOthers);
        }
        return launch_spmv(h, h->p.p, h->Ap.p, true, h->scal.p + pl::S_PAP * pl::kSlots);
      }
      case 11: {   // one whole iteration exactly as the next pl_solve runs it (storage width and node elimination of its plan)
        if (!(mp_applies(h) && h->opt.precision == 1)) return pcg_iteration(h, k);
        float *p32 = reinterpret_cast<float *>(h->z.p), *Ap32 = p32 + n6;
        float *x32 = reinterpret_cast<float *>(h->tmp.p), *r32 = x32 + n6;
        double *cur = h->scal.p + (k & 1) * pl::S_COUNT * pl::kSlots, *nxt = h->scal.p + ((k + 1) & 1) * pl::S_COUNT * pl::kSlots;
        int r11;
        if (h->cond_use) {
          r11 = launch_spmv_f32(h, p32, p32, false, nullptr, nullptr, pl::kEndsCondensedSolve);
          if (!r11) r11 = launch_spmv_f32(h, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots, h->maskC.p, pl::kEndsOthers);
        } else {
          r11 = launch_spmv_f32(h, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots);
        }
        return r11 ? r11 : pcg_tail_coarse_t<float, float>(h, cur, nxt, k, p32, (const float *)Ap32, x32, r32);
      }
      default: return fail(PL_ERR_ARG, "pl_time_kernel: unknown kernel id");
    }
  };
  for (int k = 0; k < 3; ++k) {   // warm-up
    rc = one(0);
    if (rc) return rc;
  }
  PL_HIP(hipEventRecord(h->ev0, h->stream));
  for (int k = 0; k < reps; ++k) {
    rc = one(k);
    if (rc) return rc;
  }
  PL_HIP(hipEventRecord(h->ev1, h->stream));
  PL_HIP(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  PL_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = (double)ms / reps;
  return PL_OK;
}

int pl_debug_spd_solve(int device, int32_t n, const double *A, const double *b, double *x, double *quad,
                       int32_t fp32_factor) {
  return fp32_factor ? debug_spd_solve_t<float>(device, n, A, b, x, quad)
                     : debug_spd_solve_t<double>(device, n, A, b, x, quad);
}

int pl_lzone(int device, int64_t n_nodes, int64_t n_beams, const double *node_xyz, const int32_t *beam_conn,
             const double *beam_radius, double *lzone) {
  if (n_nodes <= 0 || n_beams <= 0 || !node_xyz || !beam_conn || !beam_radius || !lzone)
    return fail(PL_ERR_ARG, "pl_lzone: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(PL_ERR_NODEVICE, "pl_lzone: no HIP device visible (libpylattice_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(PL_ERR_ARG, "pl_lzone: bad device ordinal");
  const int64_t nh = 2 * n_beams;
  std::vector<int64_t> ptr((size_t)n_nodes + 1, 0);
  for (int64_t h = 0; h < nh; ++h) {
    const int32_t v = beam_conn[h];
    if (v < 0 || v >= n_nodes) return fail(PL_ERR_ARG, "pl_lzone: node id out of range");
    ptr[(size_t)v + 1]++;
  }
  for (int64_t i = 0; i < n_nodes; ++i) ptr[i + 1] += ptr[i];
  std::vector<int32_t> half((size_t)nh);
  {
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t h = 0; h < nh; ++h) half[(size_t)fill[beam_conn[h]]++] = (int32_t)h;
  }
  PL_HIP(hipSetDevice(device));
  DevBuf<double> dx, dr, dl;
  DevBuf<int32_t> dc, dh;
  DevBuf<int64_t> dp;
  PL_HIP(dx.alloc((size_t)n_nodes * 3));
  PL_HIP(dr.alloc(n_beams));
  PL_HIP(dl.alloc(nh));
  PL_HIP(dc.alloc(nh));
  PL_HIP(dh.alloc(nh));
  PL_HIP(dp.alloc(n_nodes + 1));
  PL_HIP(hipMemcpy(dx.p, node_xyz, (size_t)n_nodes * 3 * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dr.p, beam_radius, n_beams * sizeof(double), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dc.p, beam_conn, nh * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dh.p, half.data(), nh * sizeof(int32_t), hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(dp.p, ptr.data(), (n_nodes + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(pl::k_lzone, dim3(grid_for(nh)), dim3(pl::kBlock), 0, nullptr, nh, dx.p, dc.p, dr.p, dp.p, dh.p,
                     dl.p);
  PL_HIP(hipGetLastError());
  PL_HIP(hipMemcpy(lzone, dl.p, nh * sizeof(double), hipMemcpyDeviceToHost));
  return PL_OK;
}

#ifdef PL_TILE_STAMPS
// experiment builds only (not in the header): the clock stamps of the last tile K*p launch, out[8 * 4096]
int pl_debug_tile_stamps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pl::g_tile_stamps), sizeof(unsigned long long) * 8 * 4096) == hipSuccess ? 0 : -2;
}
#endif

int pl_dist_unique_id_bytes(void) { return pl::dist_unique_id_bytes(); }
int pl_dist_unique_id(void *id_out) {
  if (!id_out) return fail(PL_ERR_ARG, "pl_dist_unique_id: null argument");
  return pl::dist_unique_id(id_out) ? fail(PL_ERR_HIP, "ncclGetUniqueId failed") : PL_OK;
}
int pl_dist_loopback_id(void *id_out) {
  if (!id_out) return fail(PL_ERR_ARG, "pl_dist_loopback_id: null argument");
  pl::dist_loopback_id(id_out);
  return PL_OK;
}
int pl_dist_init(pl_handle h, int rank, int world, const void *unique_id, const int32_t *shared_local,
                 const int32_t *shared_global, int32_t n_shared, int32_t n_shared_global) {
  if (!valid(h) || !unique_id || world < 1 || rank < 0 || rank >= world || n_shared < 0)
    return fail(PL_ERR_ARG, "pl_dist_init: bad argument");
  if (n_shared > 0 && (!shared_local || !shared_global)) return fail(PL_ERR_ARG, "pl_dist_init: null index array");
  PL_HIP(hipSetDevice(h->opt.device));
  std::vector<int32_t> loc(n_shared);
  for (int i = 0; i < n_shared; ++i) {
    if (shared_local[i] < 0 || shared_local[i] >= h->N || shared_global[i] < 0 || shared_global[i] >= n_shared_global)
      return fail(PL_ERR_ARG, "pl_dist_init: shared node index out of range");
    loc[i] = h->iperm[shared_local[i]];
  }
  int rc = pl::dist_init(h->dist, rank, world, unique_id, loc.data(), shared_global, n_shared, n_shared_global, h->N,
                         h->stream);
  if (rc == 6)
    return fail(PL_ERR_ARG, "pl_dist_init: loopback group mismatch (all ranks of a loopback id must share one device and one "
                            "world size <= " + std::to_string(pl::kLoopMaxWorld) + ", each rank attaches once)");
  if (rc) return fail(PL_ERR_HIP, "pl_dist_init: communicator setup failed (" + std::to_string(rc) + ")");
  h->h_shared.assign((size_t)h->N, 0);
  for (int i = 0; i < n_shared; ++i) h->h_shared[loc[i]] = 1;
  h->ov_ready = false;
  if (h->coarse.enabled) {   // the tile level and the rank-local dense level leave shared nodes out
    h->cond_ready = false;   // (re-selected at the next pl_set_bc; until then nothing is condensed)
    h->n_cond = 0;
    h->cond_agree = -1;
    PL_HIP(hipMemcpy(h->sharedbits.p, h->h_shared.data(), h->h_shared.size(), hipMemcpyHostToDevice));
    h->coarseL.n_fix = -1;
  }
  h->assembled = false;
  return PL_OK;
}

int pl_dist_abort(pl_handle h) {
  if (!valid(h)) return fail(PL_ERR_ARG, "pl_dist_abort: null handle");
  if (h->dist.loop) {
    std::lock_guard<std::mutex> lk(h->dist.loop->m);
    h->dist.loop->broken = true;
    h->dist.loop->cv.notify_all();
  }
  return PL_OK;
}

int pl_dist_set_peers(pl_handle h, const int32_t *shared_peer) {
  if (!valid(h) || !shared_peer) return fail(PL_ERR_ARG, "pl_dist_set_peers: null argument");
  if (!h->dist.active) return fail(PL_ERR_STATE, "pl_dist_set_peers: call pl_dist_init first");
  PL_HIP(hipSetDevice(h->opt.device));
  int rc = pl::dist_set_peers(h->dist, shared_peer);
  if (rc) return fail(rc == 2 ? PL_ERR_ARG : PL_ERR_HIP, "pl_dist_set_peers: setup failed (" + std::to_string(rc) + ")");
  h->assembled = false;
  // exchange / compute overlap of the tile K*p: which tiles own interface rows
  h->ov_ready = false;
  if (h->opt.overlap >= 0 && h->tile.ready && h->opkind == 0 && !h->h_tile_start.empty()) {
    const int64_t T = (int64_t)h->h_tile_start.size() - 1;
    std::vector<int32_t> iface, inner;
    for (int64_t t = 0; t < T; ++t) {
      bool has = false;
      for (int32_t i = h->h_tile_start[t]; i < h->h_tile_start[t + 1] && !has; ++i) has = h->h_shared[i] != 0;
      (has ? iface : inner).push_back((int32_t)t);
    }
    if (!iface.empty() && !inner.empty()) {
      PL_HIP(h->ov_iface.alloc(iface.size()));
      PL_HIP(h->ov_inner.alloc(inner.size()));
      PL_HIP(hipMemcpy(h->ov_iface.p, iface.data(), iface.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      PL_HIP(hipMemcpy(h->ov_inner.p, inner.data(), inner.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      h->n_ov_iface = (int64_t)iface.size();
      h->n_ov_inner = (int64_t)inner.size();
      if (!h->comm_stream) {
        int prio_low = 0, prio_high = 0;
        PL_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        PL_HIP(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, prio_high));
        PL_HIP(hipEventCreateWithFlags(&h->ev_ov_a, hipEventDisableTiming));
        PL_HIP(hipEventCreateWithFlags(&h->ev_ov_x, hipEventDisableTiming));
      }
      h->ov_ready = true;
    }
  }
  return PL_OK;
}

}  // extern "C"
