// PCG solvers of libpylattice_hip: ordinary multi-level / Jacobi / reference-CG loop, single-reduction form
// (Chronopoulos-Gear), fp32 solver modes.
#pragma once
#include "pl_assembly.h"

namespace {

// Warm start (opts.warm_start, design loops): p <- the previous converged solution of this handle with the eliminated nodes'
// rows zeroed, then r -= (operator p), x = p on the rows that are unknowns of the CG.
__global__ __launch_bounds__(pl::kBlock) void k_warm_mask(int64_t N, const uint8_t *__restrict__ cflag /* may be null */,
                                                         const uint8_t *__restrict__ fixed, double *__restrict__ p) {
  const int64_t t = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (t < 6 * N && ((cflag && cflag[t / 6]) || fixed[t])) p[t] = 0.0;     // (the Dirichlet set may have changed since)
}
// opts.warm_start = 2: p <- 2 x_prev - x_prev2, the linear extrapolation of the last two solutions of a design loop
__global__ __launch_bounds__(pl::kBlock) void k_warm_extrapolate(int64_t n6, const double *__restrict__ a /* x_prev */,
                                                                 const double *__restrict__ b /* x_prev2 */, double *__restrict__ p) {
  const int64_t t = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (t < n6) p[t] = 2.0 * a[t] - b[t];
}
__global__ __launch_bounds__(pl::kBlock) void k_warm_extrapolate2(int64_t n6, const double *__restrict__ a, const double *__restrict__ b,
                                                                  const double *__restrict__ c3, double *__restrict__ p) {
  const int64_t t = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (t < n6) p[t] = 3.0 * (a[t] - b[t]) + c3[t];
}
// opts.warm_start = 4 (Galerkin start): out[i] += v_i . Ap for the m stored vectors, out[m] += vj . r - rows that are unknowns
// of the CG only (not eliminated, not constrained: v is masked, Ap is masked)
constexpr int kWarmMax = pl_context::kWarmMax;
struct WarmVecs {
  const double *v[kWarmMax];
};
struct WarmCoef {
  double c[kWarmMax];
};
__global__ __launch_bounds__(pl::kBlock) void k_warm_dots(int64_t N, int m, WarmVecs V, int j, const uint8_t *__restrict__ cflag,
                                                          const double *__restrict__ Ap, const double *__restrict__ r,
                                                          double *__restrict__ out /* [m + 1] */) {
  __shared__ double red[kWarmMax + 1][pl::kBlock / pl::kWave];
  double acc[kWarmMax + 1];
#pragma unroll
  for (int i = 0; i <= kWarmMax; ++i) acc[i] = 0.0;
  for (int64_t t = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; t < 6 * N; t += (int64_t)gridDim.x * pl::kBlock) {
    if (cflag && cflag[t / 6]) continue;
    const double a = Ap[t];
#pragma unroll
    for (int i = 0; i < kWarmMax; ++i)
      if (i < m) acc[i] += V.v[i][t] * a;
    acc[kWarmMax] += V.v[j][t] * r[t];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i <= kWarmMax; ++i) {
    const double s = pl::wave_sum(acc[i]);
    if (lane == 0) red[i][wv] = s;
  }
  __syncthreads();
  if (threadIdx.x <= kWarmMax) {
    double s = 0.0;
    for (int w = 0; w < pl::kBlock / pl::kWave; ++w) s += red[threadIdx.x][w];
    const int i = threadIdx.x;
    if (i < m) unsafeAtomicAdd(out + i, s);
    else if (i == kWarmMax) unsafeAtomicAdd(out + m, s);
  }
}
// p = sum_i c_i v_i
__global__ __launch_bounds__(pl::kBlock) void k_warm_combine(int64_t n6, int m, WarmVecs V, WarmCoef C, double *__restrict__ p) {
  const int64_t t = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (t >= n6) return;
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kWarmMax; ++i)
    if (i < m) s += C.c[i] * V.v[i][t];
  p[t] = s;
}
__global__ __launch_bounds__(pl::kBlock) void k_warm_apply(int64_t N, const uint8_t *__restrict__ cflag /* may be null */,
                                                          const double *__restrict__ Ap, const double *__restrict__ p,
                                                          double *__restrict__ r, double *__restrict__ x) {
  const int64_t t = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (t >= 6 * N) return;
  if (cflag && cflag[t / 6]) return;
  r[t] -= Ap[t];
  x[t] = p[t];
}

// Everything of a two-level PCG iteration after K*p: update + restriction, coarse solve, new direction.
template <typename PT, typename RT>
int pcg_tail_coarse_t(pl_context *c, double *cur, double *nxt, int hist_slot, PT *p, const PT *Ap, RT *x, RT *r) {
  pl::Coarse &cs = c->coarse, &cl = c->coarseL;
  const bool useL = cl.ready;
#define PL_UPD(TM, MULTI, LOCAL)                                                                                    \
  hipLaunchKernelGGL((pl::k_pcg_update_tile<PT, RT, TM, MULTI, LOCAL>), dim3((unsigned)cs.n_tiles), dim3(cs.vblock), \
                     0, c->stream,                                                                                   \
                     c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p, Ap, cs.dinv32,                       \
                     c->dist.active ? (const double *)c->dist.weight.p : (const double *)nullptr, r, cur,             \
                     cs.rc, cs.tile_level ? (const double *)cs.Bt_inv : (const double *)nullptr, cs.yt,               \
                     useL ? (const int32_t *)cl.agg_of_tile.p : (const int32_t *)nullptr, cl.cen.p,                   \
                     (c->dist.active || useL) ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr, cl.rc,   \
                     cs.ncp, c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, cs.cm, rel32)
  // lever arms as exact floats where the lattice allows it (Coarse::rel32); the rank-local level has its own reference points
  const float *rel32 = cs.rel_exact ? (const float *)cs.rel32.p : (const float *)nullptr;
  if (useL) {
    if (tile_modes_now(c) == 12) PL_UPD(12, true, true);
    else PL_UPD(6, true, true);
  } else if (c->dist.active) {
    if (tile_modes_now(c) == 12) PL_UPD(12, true, false);
    else PL_UPD(6, true, false);
  } else if (tile_modes_now(c) == 12) PL_UPD(12, false, false);
  else PL_UPD(6, false, false);
#undef PL_UPD
  if (useL)   // rank-local level: y_L is never communicated, but its share of r.z, r_L . A_L^-1 r_L, is a per-rank
              // partial sum: it joins the r.D^-1 r slots BEFORE they travel in the collective below
    pl::coarse_apply(cl, cl.rc, cl.tv, cl.yc, cs.rc + cs.ncp + pl::kSlots, (const double *)nullptr, c->stream);
  if (c->dist.active) {   // one collective: [Z^T r | r.r slots | r.D^-1 r slots]; the coarse solve is then redundant per rank
    if (pl::dist_sum_scalars(c->dist, cs.rc, cs.ncp + 2 * pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the coarse residual failed");
  }
  pl::coarse_apply(cs, cs.rc, cs.tv, cs.yc, cur + pl::S_RZ_NEW * pl::kSlots, cs.rc + cs.ncp + pl::kSlots, c->stream);
#define PL_DIR(TM, MULTI, LOCAL)                                                                                         \
  hipLaunchKernelGGL((pl::k_pcg_direction_coarse<PT, RT, TM, MULTI, LOCAL>), dim3((unsigned)cs.n_tiles), dim3(cs.vblock), \
                     0, c->stream,                                                                                        \
                     c->tile.tile_start.p, (const RT *)r, cs.dinv32, c->xyz.p, cs.agg_of_tile.p, cs.cen.p, cs.yc,         \
                     cs.tile_level ? (const double *)cs.yt : (const double *)nullptr, c->fixedbits.p, p, x, cur, nxt,     \
                     c->hist.p, hist_slot, cs.rc, cs.ncp,                                                                \
                     useL ? (const int32_t *)cl.agg_of_tile.p : (const int32_t *)nullptr, cl.cen.p, cl.yc,                \
                     (c->dist.active || useL) ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr, cl.rc,       \
                     cl.ncp, c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, cs.cm, rel32)
  const int64_t n_flat = c->cond_use ? c->N - c->n_cond : c->N;       // nodes that are unknowns of this CG
#define PL_DIRF(TM)                                                                                                      \
  hipLaunchKernelGGL((pl::k_pcg_direction_flat<PT, RT, TM>), dim3((unsigned)((3 * n_flat + pl::kBlock - 1) / pl::kBlock)), \
                     dim3(pl::kBlock), 0, c->stream, n_flat, cs.tile_of_node.p, (const RT *)r, cs.dinv32, c->xyz.p,       \
                     cs.agg_of_tile.p, cs.cen.p, cs.yc, cs.tile_level ? (const double *)cs.yt : (const double *)nullptr,  \
                     c->fixedbits.p, p, x, cur, nxt, c->hist.p, hist_slot, cs.rc, cs.ncp,                                \
                     c->cond_use ? (const int32_t *)c->ckeep.p : (const int32_t *)nullptr, cs.cm,                        \
                     c->dist.active ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr, rel32)
  // flat mapping, contiguous per wave: fp64 p without a rank-local level (measured on one box, 50^3 Octet: 26.2 -> 24.5 us;
  // fp32 p / fp64 r the same either way, fp32 p / fp32 r 19.0 -> 22.0 us - 8-byte loads per lane are too few in flight)
  if (!useL && sizeof(PT) == 8) {
    if (tile_modes_now(c) == 12) PL_DIRF(12);
    else PL_DIRF(6);
  } else if (useL) {
    if (tile_modes_now(c) == 12) PL_DIR(12, true, true);
    else PL_DIR(6, true, true);
  } else if (c->dist.active) {
    if (tile_modes_now(c) == 12) PL_DIR(12, true, false);
    else PL_DIR(6, true, false);
  } else if (tile_modes_now(c) == 12) PL_DIR(12, false, false);
  else PL_DIR(6, false, false);
#undef PL_DIRF
#undef PL_DIR
  PL_HIP(hipGetLastError());
  return PL_OK;
}

int pcg_tail_coarse(pl_context *c, double *cur, double *nxt, int hist_slot) {
  return pcg_tail_coarse_t<double, double>(c, cur, nxt, hist_slot, c->p.p, (const double *)c->Ap.p, c->x.p, c->r.p);
}

// z = G^-1 r of a DDM handle (dense factor of the assembled matrix, or its inverted node blocks); dot[kSlots] += r.z
// (restricted = true: r_c already holds Z^T r - the iteration's fused update has summed it, k_ddm_update_restrict)
inline void ddm_precondition(pl_context *c, double *dot, bool restricted = false) {
  if (c->dd2_ready) {   // two-level: r_c = Z^T r, y_c = A_c^-1 r_c (dot += r_c . y_c), z = B^-1 r + P Z y_c (dot += r . B^-1 r)
    pl::Coarse &cs = c->dd2;
    const uint8_t *fx = c->have_bc ? (const uint8_t *)c->fixed.p : (const uint8_t *)nullptr;
    if (!restricted)
      hipLaunchKernelGGL(pl::k_ddm_restrict, dim3((unsigned)c->dd2_n_agg), dim3(pl::kBlock), 0, c->stream, c->dd2_ptr.p,
                         c->dd2_nodes.p, (const double *)c->dd2_cen.p, (const double *)c->dd2_xyz.p, fx,
                         (const double *)c->r.p, cs.rc);
    pl::coarse_apply(cs, cs.rc, cs.tv, cs.yc, dot, (const double *)nullptr, c->stream);
    hipLaunchKernelGGL(pl::k_ddm_two_level_apply, dim3(grid_for(c->N * 6)), dim3(pl::kBlock), 0, c->stream, c->N,
                       (const double *)c->dd_B.p, c->dd2_agg.p, (const double *)c->dd2_cen.p, (const double *)c->dd2_xyz.p, fx,
                       (const double *)cs.yc, (const double *)c->r.p, c->z.p, dot, cs.rc, cs.ncp);
    return;
  }
  if (c->dd_ready)
    pl::dense_apply(c->dd_W.p, c->dd_Wt.p, c->dd_n, c->dd_n, c->r.p, c->dd_tv.p, c->z.p, dot, (const double *)nullptr,
                    c->stream);
  else
    hipLaunchKernelGGL(pl::k_ddm_node_blocks_apply, dim3(grid_for(c->N * 6)), dim3(pl::kBlock), 0, c->stream, c->N,
                       (const double *)c->dd_B.p, (const double *)c->r.p, c->z.p, dot);
}

// ----------------------------------------------------------------------------------------------------------
// Short form of the iteration (pl_small.h): K*p with the direction formed in the kernel, update, z-kernel.
// ----------------------------------------------------------------------------------------------------------
inline int small_set_size() { return pl::S_COUNT * pl::kSlots; }
inline double *small_set(pl_context *c, int k) { return c->small_scal.p + (size_t)((k + 4) & 3) * small_set_size(); }
inline double *small_rc(pl_context *c, int parity) { return c->small_rc.p + (size_t)(parity & 1) * (c->coarse.ncp + 2 * pl::kSlots); }

// buffers of the short form, zeroed: scalar ring, both r_c buffers, p_old of iteration 0
int small_prepare(pl_context *c) {
  const int64_t n6 = c->N * 6;
  const size_t nrc = 2 * (size_t)(c->coarse.ncp + 2 * pl::kSlots);
  if (!c->p2.p || c->p2.n < (size_t)n6) PL_HIP(c->p2.alloc((size_t)n6));
  if (!c->small_scal.p) PL_HIP(c->small_scal.alloc(4 * (size_t)small_set_size()));
  if (!c->small_rc.p || c->small_rc.n < nrc) PL_HIP(c->small_rc.alloc(nrc));
  PL_HIP(hipMemsetAsync(c->small_scal.p, 0, 4 * (size_t)small_set_size() * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->small_rc.p, 0, nrc * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->p.p, 0, n6 * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->p2.p, 0, n6 * sizeof(double), c->stream));
  return PL_OK;
}

// update (r -= alpha Ap with the scalars of set k, restriction into the r_c buffer of parity k + 1, tile solve) and the
// z-kernel of iteration k (k = -1: the pass that prepares iteration 0, alpha = 0; its history entry goes to hist_slot)
int small_tail(pl_context *c, int k, int hist_slot) {
  pl::Coarse &cs = c->coarse;
  const uint8_t *skip = c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr;
  const double *Bt = cs.tile_level ? (const double *)cs.Bt_inv : (const double *)nullptr;
  double *rc_new = small_rc(c, k + 1), *rc_old = small_rc(c, k);
#define PL_SUPD(TM)                                                                                                         \
  hipLaunchKernelGGL((pl::k_pcg_update_tile<double, double, TM, false, false>), dim3((unsigned)cs.n_tiles), dim3(cs.vblock),  \
                     0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p, (const double *)c->Ap.p,     \
                     cs.dinv32, (const double *)nullptr, c->r.p, small_set(c, k), rc_new, Bt, cs.yt,                        \
                     (const int32_t *)nullptr, (const double *)nullptr, (const uint8_t *)nullptr, (double *)nullptr, cs.ncp, \
                     skip, cs.cm)
#define PL_SZ(TM)                                                                                                           \
  hipLaunchKernelGGL((pl::k_small_z<TM>), dim3((unsigned)cs.n_tiles), dim3(cs.vblock), 0, c->stream, c->tile.tile_start.p,  \
                     cs.agg_of_tile.p, cs.cen.p, c->xyz.p, (const double *)c->r.p, cs.dinv32, c->fixedbits.p, skip,         \
                     (const float *)cs.Ainv, cs.ncp, cs.cm, (const double *)rc_new, rc_old,                                 \
                     cs.tile_level ? (const double *)cs.yt : (const double *)nullptr, c->z.p, small_set(c, k + 1),          \
                     small_set(c, k + 2), c->hist.p, hist_slot)
  if (tile_modes_now(c) == 12) {
    PL_SUPD(12);
    PL_SZ(12);
  } else {
    PL_SUPD(6);
    PL_SZ(6);
  }
#undef PL_SUPD
#undef PL_SZ
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// p_k = z_k + beta p_{k-1} is formed by the K*p launch of iteration k and stored in buffer k & 1 (p_{k-1} in the other one)
inline double *small_p_of(pl_context *c, int k) { return (k & 1) ? c->p2.p : c->p.p; }

int small_iteration(pl_context *c, int k) {
  pl::Defer df;
  df.z = c->z.p;
  df.p_old = small_p_of(c, k - 1);
  df.p_new = small_p_of(c, k);
  df.xsol = c->x.p;
  df.sc_prev = small_set(c, k - 1);
  df.sc_cur = small_set(c, k);
  double *dot = small_set(c, k) + pl::S_PAP * pl::kSlots;
  const bool stream_form = !c->pal_lds;
  const uint32_t *vw = stream_form ? c->vword_dir.p : c->vword.p;
  const void *tab = stream_form ? (const void *)c->tile.dir_table.p : (const void *)c->pal_dense.p;
  const int n_tab = stream_form ? c->tile.n_dir : c->pal_entries;
  const pl::Rec5 *r5 = stream_form ? reinterpret_cast<const pl::Rec5 *>(c->rec5.p) : (const pl::Rec5 *)nullptr;
  if (c->cond_use) {
    const pl::CondSolve cs = cond_solve(c, pl::kEndsCondensedSolve);
    if (!pl::launch_tile_spmv_lds_defer(c->tile, vw, tab, n_tab, r5, nullptr, df.p_new, nullptr, c->stream,
                                        pl::kEndsCondensedSolve, c->cflag.p, cs, df))
      return fail(PL_ERR_STATE, "short iteration: the LDS-resident K*p does not fit this lattice");
    int rc = launch_spmv(c, df.p_new, c->Ap.p, true, dot, c->maskC.p, pl::kEndsOthers);
    if (rc) return rc;
  } else {
    if (!pl::launch_tile_spmv_lds_defer(c->tile, vw, tab, n_tab, r5, c->fixedbits.p, c->Ap.p, dot, c->stream, pl::kEndsAll,
                                        nullptr, pl::CondSolve(), df))
      return fail(PL_ERR_STATE, "short iteration: the LDS-resident K*p does not fit this lattice");
  }
  return small_tail(c, k, k);
}

// x += alpha_{K-1} p_{K-1} after K iterations (the K*p launch of iteration K would have made it)
int small_finish(pl_context *c, int K) {
  if (K <= 0) return PL_OK;
  hipLaunchKernelGGL(pl::k_small_final_x, dim3(grid_stream(c->N * 6)), dim3(pl::kBlock), 0, c->stream, c->N,
                     c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr,
                     (const double *)small_p_of(c, K - 1), (const double *)small_set(c, K - 1), c->x.p);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// v <- Q v (periodic constraints, pl_set_periodic)
inline void periodic_average(pl_context *c, double *v) {
  if (c->n_per_groups > 0)
    hipLaunchKernelGGL(pl::k_periodic_average, dim3(grid_for(6 * c->n_per_groups)), dim3(pl::kBlock), 0, c->stream,
                       c->n_per_groups, c->per_ptr.p, c->per_nodes.p, v);
}

// One PCG iteration (k = iteration index: selects the scalar set by parity and the residual-history slot).
int pcg_iteration(pl_context *c, int k) {
  if (c->small_use) return small_iteration(c, k);
  const int64_t n6 = c->N * 6;
  const int set = pl::S_COUNT * pl::kSlots;
  double *cur = c->scal.p + (k & 1) * set, *nxt = c->scal.p + ((k + 1) & 1) * set;
  if (c->cond_use) {
    // S p: the condensed nodes take their equilibrium position under p (first pass, their rows of p are 0 on entry),
    // then the ordinary product with their rows masked like Dirichlet rows (second pass, with p.Ap)
    // (first pass fused with the 6 x 6 solves: every tile writes -K_cc^-1 (K p_v)_c into the p rows of its condensed nodes)
    int rc = launch_spmv(c, c->p.p, c->p.p, false, nullptr, nullptr, pl::kEndsCondensedSolve);
    if (rc) return rc;
    rc = launch_spmv(c, c->p.p, c->Ap.p, true, cur + pl::S_PAP * pl::kSlots, c->maskC.p, pl::kEndsOthers);
    if (rc) return rc;
    return pcg_tail_coarse(c, cur, nxt, k);
  }
  int rc = launch_spmv(c, c->p.p, c->Ap.p, true, cur + pl::S_PAP * pl::kSlots);
  if (rc) return rc;
  if (c->coarse.ready) return pcg_tail_coarse(c, cur, nxt, k);
  periodic_average(c, c->Ap.p);        // (periodic constraints: the operator is Q K Q; p.Kp above is already p.QKQp)
  // reference-CG mode (conjugate_gradient_solver.py:79-109): every restart_every-th iteration the direction is rebuilt
  // on the PREVIOUS z (with a preconditioner; kept in tmp) or on the updated residual (without one: z aliases r there)
  const bool ref = ref_cg(c) && !c->dist.active;
  const bool restart = ref && c->opt.restart_every > 0 && k > 0 && (k % c->opt.restart_every) == 0;
  const bool has_M = c->dd_ready || c->dd_blocks || c->opt.precond >= 1;
  const double *pn = c->p.p, *psrc = nullptr;
  if (restart) {
    if (has_M) {
      PL_HIP(hipMemcpyAsync(c->tmp.p, c->z.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      pn = psrc = c->tmp.p;
    } else {
      pn = nullptr;        // ||r_new||
      psrc = c->z.p;       // = r_new once the update kernel has run (dinv = 1 on free dofs)
    }
  }
  const int hcap = ref ? c->hist_cap : 0;
  int *stop = c->stop_use ? c->stop_flag.p : (int *)nullptr;     // (DDM handles: the device stops itself, k_pcg_direction)
  const double stop_mintol = ref ? c->opt.mintol : 0.0;
  if (c->dd2_ready) {                  // two-level DDM preconditioner: update and restriction in one pass over the aggregates
    const uint8_t *fx = c->have_bc ? (const uint8_t *)c->fixed.p : (const uint8_t *)nullptr;
    const dim3 g((unsigned)(c->dd2_n_agg * pl::kDdmSplit));
    if (ref)
      hipLaunchKernelGGL(pl::k_ddm_update_restrict<true>, g, dim3(pl::kBlock), 0, c->stream, c->dd2_ptr.p, c->dd2_nodes.p,
                         (const double *)c->dd2_cen.p, (const double *)c->dd2_xyz.p, fx, (const double *)c->p.p,
                         (const double *)c->Ap.p, c->x.p, c->r.p, cur, c->opt.alpha_max, pn, (const int *)stop, c->dd2.rc);
    else
      hipLaunchKernelGGL(pl::k_ddm_update_restrict<false>, g, dim3(pl::kBlock), 0, c->stream, c->dd2_ptr.p, c->dd2_nodes.p,
                         (const double *)c->dd2_cen.p, (const double *)c->dd2_xyz.p, fx, (const double *)c->p.p,
                         (const double *)c->Ap.p, c->x.p, c->r.p, cur, c->opt.alpha_max, (const double *)nullptr,
                         (const int *)stop, c->dd2.rc);
    ddm_precondition(c, cur + pl::S_RZ_NEW * pl::kSlots, true);
    hipLaunchKernelGGL(pl::k_pcg_direction, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->z.p,
                       c->p.p, cur, nxt, c->hist.p, k, psrc, hcap, stop, c->stop_thresh, stop_mintol);
    PL_HIP(hipGetLastError());
    return PL_OK;
  }
  if (c->dd_ready || c->dd_blocks) {   // DDM with the factorised assembled matrix / its node blocks: update leaves z = 0,
                                       // r.z = 0; then z = G^-1 r
    if (ref)
      hipLaunchKernelGGL(pl::k_pcg_update<true>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                         c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, pn, (const int *)stop);
    else
      hipLaunchKernelGGL(pl::k_pcg_update<false>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                         c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, (const double *)nullptr,
                         (const int *)stop);
    ddm_precondition(c, cur + pl::S_RZ_NEW * pl::kSlots);
    hipLaunchKernelGGL(pl::k_pcg_direction, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->z.p,
                       c->p.p, cur, nxt, c->hist.p, k, psrc, hcap, stop, c->stop_thresh, stop_mintol);
    PL_HIP(hipGetLastError());
    return PL_OK;
  }
  if (c->dist.active) {
    pl::launch_pcg_update_weighted(n6, c->p.p, c->Ap.p, c->dinv.p, c->dist.weight.p, c->x.p, c->r.p, c->z.p, cur,
                                   c->stream);
    if (pl::dist_sum_scalars(c->dist, cur + pl::S_RZ_NEW * pl::kSlots, 2 * pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the PCG scalars failed");
  } else if (ref) {
    hipLaunchKernelGGL(pl::k_pcg_update<true>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                       c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, pn, (const int *)stop);
  } else {
    hipLaunchKernelGGL(pl::k_pcg_update<false>, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->p.p,
                       c->Ap.p, c->dinv.p, c->x.p, c->r.p, c->z.p, cur, c->opt.alpha_max, (const double *)nullptr,
                       (const int *)stop);
  }
  periodic_average(c, c->z.p);         // z = Q D^-1 r (r.z was summed with the un-averaged D^-1 r: the same number, r = Q r)
  hipLaunchKernelGGL(pl::k_pcg_direction, dim3(grid_stream(n6 / 2)), dim3(pl::kBlock), 0, c->stream, n6, c->z.p,
                     c->p.p, cur, nxt, c->hist.p, k, psrc, hcap, stop, c->stop_thresh, stop_mintol);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// The PCG loop as one persistent launch (pl_persist.h): x, r hold the initial iterate and residual on entry, the result on exit.
int persist_solve(pl_context *c, double thresh, double bb, int max_iter, pl_stats_t *st) {
  pl::Coarse &cs = c->coarse;
  const int64_t n6 = c->N * 6;
  const int G = (int)c->tile.n_tiles;
  if (!c->ps_Ug.p) {
    PL_HIP(c->ps_Ug.alloc((size_t)n6));
    PL_HIP(c->ps_red.alloc((size_t)G * pl::kPersistRed));
    PL_HIP(c->ps_flags.alloc((size_t)2 * G + 4));
    PL_HIP(c->ps_dbg.alloc(8));
    // aggregate -> its tiles
    std::vector<int32_t> aot((size_t)G);
    PL_HIP(hipMemcpy(aot.data(), cs.agg_of_tile.p, (size_t)G * sizeof(int32_t), hipMemcpyDeviceToHost));
    int n_agg = 0;
    for (int32_t v : aot) n_agg = std::max(n_agg, v + 1);
    std::vector<int32_t> ptr((size_t)n_agg + 1, 0), idx((size_t)G);
    for (int32_t v : aot) ptr[(size_t)v + 1]++;
    for (int g = 0; g < n_agg; ++g) ptr[(size_t)g + 1] += ptr[(size_t)g];
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (int t = 0; t < G; ++t) idx[(size_t)fill[(size_t)aot[(size_t)t]]++] = t;
    PL_HIP(c->ps_agg_ptr.alloc(ptr.size()));
    PL_HIP(c->ps_agg_idx.alloc(idx.size()));
    PL_HIP(hipMemcpy(c->ps_agg_ptr.p, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    PL_HIP(hipMemcpy(c->ps_agg_idx.p, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    c->ps_n_agg = n_agg;
  }
  if ((int64_t)c->ps_n_agg * cs.cm > cs.ncp) return fail(PL_ERR_STATE, "persistent PCG: aggregate table does not match the dense level");
  PL_HIP(hipMemsetAsync(c->ps_flags.p, 0, ((size_t)2 * G + 4) * sizeof(unsigned), c->stream));
  const bool stream_form = !c->pal_lds;
  pl::PersistArgs a;
  a.tdesc = c->tile.tdesc.p;
  a.vword = stream_form ? c->vword_dir.p : c->vword.p;
  a.vother = c->tile.vother.p;
  a.tab = static_cast<const double2 *>(stream_form ? (const void *)c->tile.dir_table.p : (const void *)c->pal_dense.p);
  a.n_tab = stream_form ? c->tile.n_dir : c->pal_entries;
  a.rec5 = stream_form ? reinterpret_cast<const pl::Rec5 *>(c->rec5.p) : (const pl::Rec5 *)nullptr;
  a.foreign_idx = c->tile.foreign_idx.p;
  a.fixedbits = c->fixedbits.p;
  a.stride = c->tile.max_nodes | 1;
  a.agg_of_tile = cs.agg_of_tile.p;
  a.cen = cs.cen.p;
  a.xyz = c->xyz.p;
  a.dinv32 = cs.dinv32;
  a.Bt_inv = cs.tile_level ? (const double *)cs.Bt_inv : (const double *)nullptr;
  a.Ainv = cs.Ainv;
  a.agg_tile_ptr = c->ps_agg_ptr.p;
  a.agg_tile_idx = c->ps_agg_idx.p;
  a.ncp = cs.ncp;
  a.cm = cs.cm;
  a.n_agg = c->ps_n_agg;
  a.x = c->x.p;
  a.r = c->r.p;
  a.Ug = c->ps_Ug.p;
  a.red = c->ps_red.p;
  a.flagU = c->ps_flags.p;
  a.flagR = c->ps_flags.p + G;
  a.err = c->ps_flags.p + 2 * G;
  a.hist = c->hist.p;
  a.max_iter = max_iter;
  a.thresh = thresh;
  a.iters_out = reinterpret_cast<int *>(c->ps_flags.p + 2 * G + 1);
  a.dbg = c->ps_dbg.p;
  const size_t lds = ((size_t)12 * a.stride + (size_t)2 * a.n_tab * (stream_form ? 2 : pl::kPalLdsChunks) + (size_t)2 * cs.ncp +
                      (size_t)G * pl::kPersistRed) * sizeof(double) + ((size_t)c->ps_n_agg + 1 + G + 2) * sizeof(int32_t);
  if (lds > 150 * 1024) return fail(PL_ERR_STATE, "persistent PCG: the tile state does not fit the LDS");
  const bool tm12 = tile_modes_now(c) == 12;
#define PL_PS3(R, TM, NC)                                                                                               \
  do {                                                                                                                  \
    PL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pl::k_persist_cg1<R, TM, NC>),                           \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                   \
    hipLaunchKernelGGL((pl::k_persist_cg1<R, TM, NC>), dim3((unsigned)G), dim3(pl::kPersistBlock), lds, c->stream, a); \
  } while (0)
#define PL_PS(R, TM)                                                                   \
  do {                                                                                 \
    if (cs.ncp <= 2 * pl::kPersistBlock) PL_PS3(R, TM, 2);                             \
    else if (cs.ncp <= 3 * pl::kPersistBlock) PL_PS3(R, TM, 3);                        \
    else PL_PS3(R, TM, 4);                                                             \
  } while (0)
  if (stream_form) {
    if (tm12) PL_PS(pl::kRecCompact, 12);
    else PL_PS(pl::kRecCompact, 6);
  } else {
    if (tm12) PL_PS(pl::kRecPalette, 12);
    else PL_PS(pl::kRecPalette, 6);
  }
#undef PL_PS
#undef PL_PS3
  PL_HIP(hipGetLastError());
  unsigned tail[3] = {0, 0, 0};
  PL_HIP(hipMemcpyAsync(tail, c->ps_flags.p + 2 * G, sizeof(tail), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  if (tail[0] != 0) return fail(PL_ERR_HIP, "persistent PCG: a hand-off between workgroups timed out");
  if (const char *e = std::getenv("PL_PERSIST_DEBUG"); e && e[0] == '1') {
    unsigned long long ph[8];
    PL_HIP(hipMemcpy(ph, c->ps_dbg.p, sizeof(ph), hipMemcpyDeviceToHost));
    const double it = std::max(1u, tail[1]);
    std::fprintf(stderr, "[persist] %u iterations; us per iteration in workgroup 0: u + publish %.2f | wait u %.2f | K u %.2f | sums + publish "
                 "%.2f | wait + gather %.2f | scalars + recurrences %.2f\n", tail[1], ph[0] * 0.01 / it, ph[1] * 0.01 / it,
                 ph[2] * 0.01 / it, ph[3] * 0.01 / it, ph[4] * 0.01 / it, ph[5] * 0.01 / it);
  }
  const int its = (int)tail[1];
  st->iterations = its;
  st->converged = (int)tail[2];
  st->info = st->converged ? 0.0 : 1.0;
  st->short_iteration_used = 2.0;
  if (its >= 0 && its < c->hist_cap) {
    double rr = 0.0;
    PL_HIP(hipMemcpy(&rr, c->hist.p + std::min(its, max_iter - 1), sizeof(double), hipMemcpyDeviceToHost));
    if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
    st->rel_residual = std::sqrt(rr / bb);
  }
  return PL_OK;
}

// Solve P K P x = rhs (device rhs already masked), x0 = 0.  Result in c->x.  Returns iterations through stats.
int pcg_solve(pl_context *c, const double *f_dev, const double *Kubar_dev, double rtol, int max_iter,
              pl_stats_t *st) {
  const int64_t n6 = c->N * 6;
  int rc = ensure_hist(c, max_iter + 1);
  if (rc) return rc;
  PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * pl::S_COUNT * pl::kSlots * sizeof(double), c->stream));
  if (c->dist.active)
    pl::launch_pcg_init_weighted(n6, f_dev, Kubar_dev, c->fixed.p, c->dinv.p, c->dist.weight.p, c->x.p, c->r.p,
                                 c->z.p, c->p.p, c->scal.p, c->stream);
  else
    hipLaunchKernelGGL(pl::k_pcg_init, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       c->fixed.p, c->dinv.p, c->x.p, c->r.p, c->z.p, c->p.p, c->scal.p);
  PL_HIP(hipGetLastError());
  if (c->dist.active) {
    if (pl::dist_sum_scalars(c->dist, c->scal.p + pl::S_RZ_OLD * pl::kSlots, pl::kSlots, c->stream) ||
        pl::dist_sum_scalars(c->dist, c->scal.p + pl::S_BB * pl::kSlots, pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the initial PCG scalars failed");
  }
  if (c->n_per_groups > 0) {   // periodic constraints: b is Q b (the caller's part), z0 = Q D^-1 r0, p0 = z0
    periodic_average(c, c->z.p);
    PL_HIP(hipMemcpyAsync(c->p.p, c->z.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  if (c->dd_ready || c->dd_blocks) {   // z0 = p0 = G^-1 r0, rz_old = r0.z0 (k_pcg_init ran with dinv = 0)
    ddm_precondition(c, c->scal.p + pl::S_RZ_OLD * pl::kSlots);
    PL_HIP(hipMemcpyAsync(c->p.p, c->z.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  if (c->cond_use) {   // (decided by solver_plan: never with a K*p kernel that ignores the elimination masks)
    // start from the iterate whose condensed nodes are in equilibrium: x_c = K_cc^-1 r_c, r <- r - K x (rows of the
    // condensed nodes become exactly 0 and stay 0: every later step keeps them in equilibrium)
    // t_c = K_cc^-1 b_c (rows of z, zero elsewhere), r_v -= (K t)_v: the load the eliminated nodes pass on.  Their own
    // rows of r keep b_c, their rows of x stay 0 until the back-substitution after the loop.
    PL_HIP(hipMemsetAsync(c->z.p, 0, n6 * sizeof(double), c->stream));
    hipLaunchKernelGGL(pl::k_condense_solve<double>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                       c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const double *)c->r.p, c->z.p, 1.0);
    rc = launch_spmv(c, c->z.p, c->tmp2.p, true, nullptr, c->maskC.p, pl::kEndsOthers);
    if (rc) return rc;
    hipLaunchKernelGGL(pl::k_condense_subtract<double>, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N, c->cflag.p,
                       (const double *)c->tmp2.p, c->r.p);
    PL_HIP(hipGetLastError());
  }
  const bool warm = c->opt.warm_start >= 1 && c->coarse.ready && !c->dist.active;
  if (warm && c->xprev.p && c->xprev_valid) {
    // x0 = the previous solution: r0 = b - S x0 through the same operator the iterations apply (with node elimination: first
    // pass fills the eliminated rows of p, second pass takes the others); the tail below then builds z0, p0 from r0
    bool combined = false;
    if (c->opt.warm_start == 4 && c->gh_count >= 1) {
      // Galerkin start: the combination x0 = sum c_i v_i of the handle's last m solutions that is nearest to the solution
      // of the CURRENT system in its energy norm, (V^T S V) c = V^T b.  It contains the previous solution and both
      // extrapolations as candidates, and - a projection - can do no worse than any of them.  m applications of the operator,
      // m + 1 dot products each, one look at m^2 + m numbers on the host.
      static const int want = [] { const char *e = std::getenv("PL_WARM_VECTORS"); const int v = e ? std::atoi(e) : 6;       // (configs[3]: 183 / 170 / 160 / 155 iterations per solve with 3 / 4 / 6 / 8)
                                   return std::max(2, std::min(kWarmMax, v)); }();
      const int m = std::min(want, 1 + c->gh_count);
      const uint8_t *cf = c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr;
      WarmVecs V;
      for (int i = 0; i < kWarmMax; ++i) V.v[i] = nullptr;
      for (int i = 0; i < m; ++i) {
        const double *src = i == 0 ? c->xprev.p
                                   : c->gh[(c->gh_head - (i - 1) + 2 * (kWarmMax - 1)) % (kWarmMax - 1)].p;
        if (!c->gw[i].p || c->gw[i].n < (size_t)n6) PL_HIP(c->gw[i].alloc(n6));
        PL_HIP(hipMemcpyAsync(c->gw[i].p, src, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_warm_mask, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N, cf, (const uint8_t *)c->fixed.p,
                           c->gw[i].p);
        V.v[i] = c->gw[i].p;
      }
      constexpr int kRow = kWarmMax + 1;
      if (!c->gw_dots.p) PL_HIP(c->gw_dots.alloc(kWarmMax * kRow));
      PL_HIP(hipMemsetAsync(c->gw_dots.p, 0, kWarmMax * kRow * sizeof(double), c->stream));
      for (int j = 0; j < m; ++j) {
        // (with node elimination the first pass writes the eliminated rows of its operand: on a copy, the dots skip those rows)
        PL_HIP(hipMemcpyAsync(c->p.p, c->gw[j].p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        if (c->cond_use) {
          rc = launch_spmv(c, c->p.p, c->p.p, false, nullptr, nullptr, pl::kEndsCondensedSolve);
          if (rc) return rc;
          rc = launch_spmv(c, c->p.p, c->Ap.p, true, nullptr, c->maskC.p, pl::kEndsOthers);
        } else {
          rc = launch_spmv(c, c->p.p, c->Ap.p, true, nullptr);
        }
        if (rc) return rc;
        hipLaunchKernelGGL(k_warm_dots, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, c->N, m, V, j, cf,
                           (const double *)c->Ap.p, (const double *)c->r.p, c->gw_dots.p + kRow * j);
      }
      double hd[kWarmMax * kRow];
      PL_HIP(hipMemcpyAsync(hd, c->gw_dots.p, sizeof(hd), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipStreamSynchronize(c->stream));
      // G[i][j] = hd[kRow j + i] (symmetrised), g[j] = hd[kRow j + m]; Cholesky with a relative pivot floor: vectors that add
      // nothing new (a stalled design path) are dropped
      double G[kWarmMax][kWarmMax], g[kWarmMax], L[kWarmMax][kWarmMax] = {}, y[kWarmMax] = {};
      WarmCoef cc;
      for (int i = 0; i < kWarmMax; ++i) cc.c[i] = 0.0;
      bool ok = true, used[kWarmMax] = {};
      for (int i = 0; i < m; ++i) {
        g[i] = hd[kRow * i + m];
        for (int j = 0; j < m; ++j) G[i][j] = 0.5 * (hd[kRow * j + i] + hd[kRow * i + j]);
        if (!(G[i][i] > 0.0) || !std::isfinite(G[i][i]) || !std::isfinite(g[i])) ok = false;
      }
      if (ok) {
        for (int j = 0; j < m; ++j) {            // (in the order newest first: the newest vector is always kept)
          double d = G[j][j];
          for (int k = 0; k < j; ++k)
            if (used[k]) d -= L[j][k] * L[j][k];
          if (!(d > 1e-10 * G[j][j])) continue;
          used[j] = true;
          L[j][j] = std::sqrt(d);
          for (int i = j + 1; i < m; ++i) {
            double v = G[i][j];
            for (int k = 0; k < j; ++k)
              if (used[k]) v -= L[i][k] * L[j][k];
            L[i][j] = v / L[j][j];
          }
        }
        for (int i = 0; i < m; ++i) {
          if (!used[i]) continue;
          double v = g[i];
          for (int k = 0; k < i; ++k)
            if (used[k]) v -= L[i][k] * y[k];
          y[i] = v / L[i][i];
        }
        for (int i = m - 1; i >= 0; --i) {
          if (!used[i]) continue;
          double v = y[i];
          for (int k = i + 1; k < m; ++k)
            if (used[k]) v -= L[k][i] * cc.c[k];
          cc.c[i] = v / L[i][i];
        }
        for (int i = 0; i < m; ++i)
          if (!std::isfinite(cc.c[i])) ok = false;
      }
      if (ok) {
        hipLaunchKernelGGL(k_warm_combine, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, n6, m, V, cc, c->p.p);
        combined = true;
      }
    }
    if (combined) {
      // (p holds the combination of masked vectors)
    } else if (c->opt.warm_start == 3 && c->xprev2_valid && c->xprev3_valid)
      hipLaunchKernelGGL(k_warm_extrapolate2, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, n6, (const double *)c->xprev.p,
                         (const double *)c->xprev2.p, (const double *)c->xprev3.p, c->p.p);
    else if ((c->opt.warm_start == 2 || c->opt.warm_start == 3) && c->xprev2.p && c->xprev2_valid)
      hipLaunchKernelGGL(k_warm_extrapolate, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, n6, (const double *)c->xprev.p,
                         (const double *)c->xprev2.p, c->p.p);
    else
      PL_HIP(hipMemcpyAsync(c->p.p, c->xprev.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_warm_mask, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N,
                       c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, (const uint8_t *)c->fixed.p, c->p.p);
    if (c->cond_use) {
      rc = launch_spmv(c, c->p.p, c->p.p, false, nullptr, nullptr, pl::kEndsCondensedSolve);
      if (rc) return rc;
      rc = launch_spmv(c, c->p.p, c->Ap.p, true, nullptr, c->maskC.p, pl::kEndsOthers);
    } else {
      rc = launch_spmv(c, c->p.p, c->Ap.p, true, nullptr);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(k_warm_apply, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N,
                       c->cond_use ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr, (const double *)c->Ap.p,
                       (const double *)c->p.p, c->r.p, c->x.p);
    PL_HIP(hipGetLastError());
  }
  if (c->persist_use) {
    // (the persistent launch builds u0 = M^-1 r0 itself)
  } else if (c->small_use) {
    // short form: z0 = M^-1 r0 and r0.z0 through the tail of an iteration "-1" (alpha = 0: set -1 of the ring is zero);
    // iteration 0 then forms p0 = z0 + 0 p_old in its K*p launch
    rc = small_prepare(c);
    if (rc) return rc;
    PL_HIP(hipMemsetAsync(c->Ap.p, 0, n6 * sizeof(double), c->stream));
    rc = small_tail(c, -1, max_iter);
    if (rc) return rc;
  } else if (c->coarse.ready) {
    // z0 = M^-1 r0 needs the coarse solve: run the tail of an iteration "-1" with p = 0, alpha = 0 (p.Ap = 0) on
    // scalar set 1; its direction kernel leaves p = z0 and rz_old = r0.z0 in set 0, where iteration 0 starts.
    const int set = pl::S_COUNT * pl::kSlots;
    PL_HIP(hipMemsetAsync(c->p.p, 0, n6 * sizeof(double), c->stream));
    PL_HIP(hipMemsetAsync(c->Ap.p, 0, n6 * sizeof(double), c->stream));
    rc = pcg_tail_coarse(c, c->scal.p + set, c->scal.p, max_iter);
    if (rc) return rc;
  }
  double h_scal[pl::kSlots];
  PL_HIP(hipMemcpyAsync(h_scal, c->scal.p + pl::S_BB * pl::kSlots, sizeof(h_scal), hipMemcpyDeviceToHost,
                        c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  double bb = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) bb += h_scal[k];
  st->b_norm = std::sqrt(bb);
  st->iterations = 0;
  st->converged = 0;
  st->rel_residual = 0.0;
  if (!(bb > 0.0)) {   // zero right-hand side -> zero solution (a warm start above may have put the previous one into x)
    st->converged = 1;
    PL_HIP(hipMemsetAsync(c->x.p, 0, n6 * sizeof(double), c->stream));
    PL_HIP(hipStreamSynchronize(c->stream));
    return std::isnan(bb) ? fail(PL_ERR_NAN, "NaN in the right-hand side") : PL_OK;
  }
  const double thresh = rtol * rtol * bb;
  if (c->persist_use) {
    rc = persist_solve(c, thresh, bb, max_iter, st);
    if (rc) return rc;
    c->last_iterations = st->converged ? st->iterations : 0;
    if (c->opt.warm_start == 1 && st->converged) {      // keep the solution for the next solve (as below)
      if (!c->xprev.p) PL_HIP(c->xprev.alloc(n6));
      PL_HIP(hipMemcpyAsync(c->xprev.p, c->x.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      c->xprev_valid = true;
    }
    return PL_OK;
  }
  // The host looks at the residual history every `chunk` iterations.  With the default (check_every = 0) the chunk
  // adapts: 32 while far from the threshold, then what the observed decay rate predicts is still needed - a fixed
  // chunk overshoots by 16 iterations on average, 8 % of a 200-iteration solve.  (Every rank of a multi-GPU run sees
  // the same all-reduced history, hence takes the same decisions.)
  const bool adaptive = c->opt.check_every <= 0;
  const int chunk = adaptive ? 32 : c->opt.check_every;
  const bool ref = ref_cg(c) && !c->dist.active && !c->coarse.ready;
  const int hcap = c->hist_cap;
  // DDM handles: the device applies the stopping rules itself and freezes the iterate (k_pcg_direction), so the host may
  // queue as many iterations as it likes between two looks at the history and still hands back the reference's iterate
  c->stop_use = c->opkind == 1 && !c->dist.active && !c->coarse.ready && !c->small_use;
  if (c->stop_use) {
    if (!c->stop_flag.p) PL_HIP(c->stop_flag.alloc(1));
    PL_HIP(hipMemsetAsync(c->stop_flag.p, 0, sizeof(int), c->stream));
    c->stop_thresh = thresh;
  }
  // A design loop solves a slowly changing system over and over: the iteration count of the previous converged solve on
  // this handle (identical on every rank) is where the first look at the history is worth taking - three iterations
  // before it - instead of every 32 iterations on the way there (each look drains the stream: 30-50 us).
  // ... and a handle preconditioned by a dense FACTOR of its own matrix (precond = 5; the assembled Schur matrix of a DDM handle,
  // precond = 2) converges in one or two steps: look after two (32 queued iterations of a 9-node cell were 0.5 ms of the 0.57 ms
  // a column of pl_schur took)
  const bool direct = adaptive && c->dd_ready;
  const int first = direct ? std::min(2, max_iter)
                           : ((adaptive && !ref && c->last_iterations > 40) ? std::min(c->last_iterations - 3, max_iter) : chunk);
  const int hbuf = std::max(chunk, first);
  std::vector<double> h_hist(hbuf), h_pp(ref ? hbuf : 0), h_xx(ref ? hbuf : 0), h_al(ref ? hbuf : 0);
  st->info = 1.0;
  int k = 0, next = first;
  double rr_prev = bb;
  int k_prev = 0;
  while (k < max_iter) {
    const int todo = std::min(next, max_iter - k);
    for (int j = 0; j < todo; ++j) {
      rc = pcg_iteration(c, k + j);
      if (rc) return rc;
    }
    PL_HIP(hipMemcpyAsync(h_hist.data(), c->hist.p + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (ref) {
      PL_HIP(hipMemcpyAsync(h_pp.data(), c->hist.p + hcap + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipMemcpyAsync(h_xx.data(), c->hist.p + 2 * (size_t)hcap + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipMemcpyAsync(h_al.data(), c->hist.p + 3 * (size_t)hcap + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    PL_HIP(hipStreamSynchronize(c->stream));
    for (int j = 0; j < todo; ++j) {
      const double rr = h_hist[j];
      if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
      if (!st->converged) st->rel_residual = std::sqrt(rr / bb);
      if (rr <= thresh && !st->converged) {
        st->converged = 1;
        st->iterations = k + j + 1;
        st->info = 0.0;
        st->stop_reason = 0.0;
      }
      if (ref && !st->converged) {
        // conjugate_gradient_solver.py:102-109, in its order: the direction-norm stop, then the "tiny step" flag
        if (c->opt.mintol > 0.0 && std::sqrt(h_pp[j]) < c->opt.mintol * (std::sqrt(h_xx[j]) + 1e-12)) {
          st->converged = 1;
          st->iterations = k + j + 1;
          st->info = 0.0;
          st->stop_reason = 1.0;
        } else if (h_al[j] < 1e-6) {
          st->info = 2.0;
        }
      }
      if (st->converged && c->stop_use) break;     // (the device stopped there too: later slots of the history are not written)
    }
    k += todo;
    if (st->converged) break;
    if (adaptive) {
      const double rr_end = h_hist[todo - 1];
      next = direct && k < 16 ? 2 : chunk;
      if (rr_end < rr_prev && rr_end > thresh) {
        const double per_it = std::log(rr_end / rr_prev) / (double)(k - k_prev);      // < 0
        const double need = std::log(thresh / rr_end) / per_it;
        if (need < 2.0 * chunk) next = std::max(2, std::min(chunk, (int)std::ceil(0.75 * need)));
      }
      rr_prev = rr_end;
      k_prev = k;
    }
  }
  if (!st->converged) st->iterations = k;
  c->last_iterations = st->converged ? st->iterations : 0;
  if (c->small_use) {
    rc = small_finish(c, k);
    if (rc) return rc;
    st->short_iteration_used = 1.0;
  }
  if (c->cond_use) {   // eliminated nodes: x_c = K_cc^-1 (b_c - (K [x_v ; 0])_c)
    rc = launch_spmv(c, c->x.p, c->tmp2.p, false, nullptr, nullptr, pl::kEndsCondensed);
    if (rc) return rc;
    hipLaunchKernelGGL(pl::k_condense_backsubst<double>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                       c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const double *)c->r.p, (const double *)c->tmp2.p, c->x.p);
    PL_HIP(hipGetLastError());
  }
  if (warm && st->converged) {      // keep the solution for the next solve (before the caller's download adds ubar)
    if (c->opt.warm_start == 4 && c->xprev.p && c->xprev_valid) {     // the Galerkin start's ring takes the solution before
      c->gh_head = (c->gh_head + 1) % (pl_context::kWarmMax - 1);
      DevBuf<double> &slot = c->gh[c->gh_head];
      if (!slot.p || slot.n < (size_t)n6) PL_HIP(slot.alloc(n6));
      PL_HIP(hipMemcpyAsync(slot.p, c->xprev.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      c->gh_count = std::min(c->gh_count + 1, pl_context::kWarmMax - 1);
    } else if (c->opt.warm_start >= 2 && c->xprev.p && c->xprev_valid) {     // ... and the one(s) before it
      if (c->opt.warm_start >= 3 && c->xprev2_valid) {
        std::swap(c->xprev2.p, c->xprev3.p);
        std::swap(c->xprev2.n, c->xprev3.n);
        c->xprev3_valid = true;
      }
      std::swap(c->xprev.p, c->xprev2.p);
      std::swap(c->xprev.n, c->xprev2.n);
      c->xprev2_valid = true;
    }
    if (!c->xprev.p) PL_HIP(c->xprev.alloc(n6));
    PL_HIP(hipMemcpyAsync(c->xprev.p, c->x.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->xprev_valid = true;
  }
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// Single-reduction PCG (opts.cg_form = 1; pl_cg1.h): u -> z, w -> Ap, s -> tmp2.
// ----------------------------------------------------------------------------------------------------------
// Which solver the next pl_solve runs, decided in ONE place (pl_solve and pl_time_kernel both ask): the fp32 modes and
// the node elimination need the multi-level preconditioner on the LDS-tile kernel; the elimination is off in the
// single-reduction form and in precision = 2.  Sets c->cond_use.
inline bool mp_applies(const pl_context *c) {
  return c->opt.precision != 0 && c->opkind == 0 && c->coarse.ready && choose_kernel(c) == 3 && c->tile.ready;
}
inline void solver_plan(pl_context *c) {
  const bool mp = mp_applies(c);
  c->cond_use = c->cond_ready && c->coarse.ready && c->opkind == 0 && choose_kernel(c) == 3 && c->tile.ready &&
                (!mp || c->opt.precision == 1) && c->opt.cg_form != 1;
  // short form of the iteration (pl_small.h): the fp64 ordinary form on the LDS-resident K*p, explicit A_c^-1 at hand
  const int form = kp_form_of(c);
  c->small_use = !mp && small_wanted(c) && c->coarse.ready && !c->coarseL.ready && c->coarse.ainv_ready &&
                 (form == 1 || form == 2) && c->tile.vis_ready;
  // ... or, opt-in, the whole loop as one persistent launch (pl_persist.h): single-reduction CG without node elimination,
  // one workgroup per tile, all co-resident
  c->persist_use = c->small_use && c->opt.short_iteration == 2 && c->tile.n_tiles <= 256 &&
                   c->coarse.ncp <= pl::kPersistBlock * pl::kPersistMaxCols && c->tile.max_nodes <= pl::kPersistBlock;
  if (c->persist_use) {
    c->cond_use = false;
    c->small_use = false;
  }
}
inline bool cg1_applies(const pl_context *c) {
  return c->opt.cg_form == 1 && c->coarse.ready && !c->coarseL.ready && !c->cond_use && c->opt.precision == 0 &&
         c->opkind == 0 && c->tile.ready && choose_kernel(c) == 3;
}

int pcg_solve_cg1(pl_context *c, const double *f_dev, const double *Kubar_dev, double rtol, int max_iter,
                  pl_stats_t *st) {
  const int64_t n6 = c->N * 6;
  pl::Coarse &cs = c->coarse;
  const int ncp = cs.ncp, bs = pl::cg1_block_size(ncp);
  int rc = ensure_hist(c, max_iter + 2);
  if (rc) return rc;
  const size_t need = 2 * (size_t)bs + 2 * pl::kSlots + 4 + (size_t)ncp;
  if (!c->cg1.p || c->cg1.n < need) PL_HIP(c->cg1.alloc(need));
  double *blk[2] = {c->cg1.p, c->cg1.p + bs};
  double *gc[2] = {c->cg1.p + 2 * bs, c->cg1.p + 2 * bs + pl::kSlots};
  double *stt[2] = {c->cg1.p + 2 * bs + 2 * pl::kSlots, c->cg1.p + 2 * bs + 2 * pl::kSlots + 2};
  double *sc = c->cg1.p + 2 * bs + 2 * pl::kSlots + 4;
  double *u = c->z.p, *w = c->Ap.p, *s = c->tmp2.p;
  const double *wt = c->dist.active ? (const double *)c->dist.weight.p : (const double *)nullptr;
  const uint8_t *shared = c->dist.active ? (const uint8_t *)c->sharedbits.p : (const uint8_t *)nullptr;
  const double *Bt = cs.tile_level ? (const double *)cs.Bt_inv : (const double *)nullptr;
  const double *yt = cs.tile_level ? (const double *)cs.yt : (const double *)nullptr;
  const dim3 gt((unsigned)cs.n_tiles), blkdim(cs.vblock);
  const bool tm12 = tile_modes_now(c) == 12, cm12 = cs.cm == 12;     // (a 12-mode dense level needs the 12-mode tile level)
  auto restrict_to = [&](const double *v, double *out) {
    if (cm12)
      hipLaunchKernelGGL(pl::k_cg1_restrict<12>, gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p,
                         c->xyz.p, v, wt, out);
    else
      hipLaunchKernelGGL(pl::k_cg1_restrict<6>, gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p, cs.cen.p,
                         c->xyz.p, v, wt, out);
  };

  PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * pl::S_COUNT * pl::kSlots * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->cg1.p, 0, need * sizeof(double), c->stream));
  // r0 = P (f - K ubar), x = 0, ||b||^2 (the z / p the init kernel also writes are overwritten below)
  if (c->dist.active)
    pl::launch_pcg_init_weighted(n6, f_dev, Kubar_dev, c->fixed.p, c->dinv.p, c->dist.weight.p, c->x.p, c->r.p,
                                 c->z.p, c->p.p, c->scal.p, c->stream);
  else
    hipLaunchKernelGGL(pl::k_pcg_init, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       c->fixed.p, c->dinv.p, c->x.p, c->r.p, c->z.p, c->p.p, c->scal.p);
  PL_HIP(hipGetLastError());
  if (c->dist.active && pl::dist_sum_scalars(c->dist, c->scal.p + pl::S_BB * pl::kSlots, pl::kSlots, c->stream))
    return fail(PL_ERR_HIP, "RCCL all-reduce of the initial PCG scalars failed");
  PL_HIP(hipMemsetAsync(c->p.p, 0, n6 * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(s, 0, n6 * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(cs.rc, 0, (size_t)ncp * sizeof(double), c->stream));

  // u = M^-1 r, w = K u and the reduction block of iteration k (k = -1: the pass that prepares iteration 0)
  auto second_half = [&](int k) -> int {
    const int cur = k & 1, nxt = (k + 1) & 1;
    pl::coarse_apply(cs, cs.rc, cs.tv, cs.yc, gc[nxt], (const double *)nullptr, c->stream);
#define PL_CG1_PRE(TM)                                                                                                  \
  hipLaunchKernelGGL(pl::k_cg1_precond<TM>, gt, blkdim, 0, c->stream, c->tile.tile_start.p, (const double *)c->r.p,     \
                     cs.dinv32, c->xyz.p, cs.agg_of_tile.p, cs.cen.p, cs.yc, yt, c->fixedbits.p, shared, u,             \
                     k >= 0 ? blk[cur] : (double *)nullptr, bs, k >= 0 ? gc[cur] : (double *)nullptr, cs.cm)
    if (tm12) PL_CG1_PRE(12);
    else PL_CG1_PRE(6);
#undef PL_CG1_PRE
    int r2 = launch_spmv(c, u, w, true, blk[nxt] + ncp, nullptr, pl::kEndsAll, false);
    if (r2) return r2;
    restrict_to((const double *)w, blk[nxt]);
    if (c->dist.active && pl::dist_sum_scalars(c->dist, blk[nxt], bs, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the single-reduction PCG failed");
    PL_HIP(hipGetLastError());
    return PL_OK;
  };
  // Z^T r0 (summed over ranks once), tile level and partial sums of r0 into block 0, then u0, w0
  restrict_to((const double *)c->r.p, cs.rc);
  if (c->dist.active && pl::dist_sum_scalars(c->dist, cs.rc, ncp, c->stream))
    return fail(PL_ERR_HIP, "RCCL all-reduce of the initial coarse residual failed");
#define PL_CG1_UPD(INIT, TM, CUR, NXT, IT)                                                                                \
  hipLaunchKernelGGL((pl::k_cg1_update<INIT, TM>), gt, blkdim, 0, c->stream, c->tile.tile_start.p, cs.agg_of_tile.p,     \
                     cs.cen.p, c->xyz.p, (const double *)u, (const double *)w, cs.dinv32, wt, c->p.p, s, c->x.p, c->r.p, \
                     (const double *)blk[CUR], (const double *)gc[CUR], (const double *)stt[CUR], stt[NXT], blk[NXT], Bt, \
                     cs.yt, shared, cs.rc, sc, ncp, c->hist.p, IT)
  if (tm12) PL_CG1_UPD(true, 12, 1, 0, -1);
  else PL_CG1_UPD(true, 6, 1, 0, -1);
  rc = second_half(-1);
  if (rc) return rc;

  double h_scal[pl::kSlots];
  PL_HIP(hipMemcpyAsync(h_scal, c->scal.p + pl::S_BB * pl::kSlots, sizeof(h_scal), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  double bb = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) bb += h_scal[k];
  st->b_norm = std::sqrt(bb);
  st->iterations = 0;
  st->converged = 0;
  st->rel_residual = 0.0;
  st->info = 1.0;
  if (!(bb > 0.0)) {
    st->converged = 1;
    return std::isnan(bb) ? fail(PL_ERR_NAN, "NaN in the right-hand side") : PL_OK;
  }
  const double thresh = rtol * rtol * bb;
  const bool adaptive = c->opt.check_every <= 0;
  const int chunk = adaptive ? 32 : c->opt.check_every;
  std::vector<double> h_hist(chunk);
  int k = 0, next = chunk, k_prev = 0;
  double rr_prev = bb;
  // hist[k] = ||r_k||^2, the residual BEFORE update k (it is reduced together with that iteration's other sums)
  while (k < max_iter + 1) {
    const int todo = std::min(next, max_iter + 1 - k);
    for (int j = 0; j < todo; ++j) {
      const int it = k + j, cur = it & 1, nxt = (it + 1) & 1;
      if (tm12) PL_CG1_UPD(false, 12, cur, nxt, it);
      else PL_CG1_UPD(false, 6, cur, nxt, it);
      rc = second_half(it);
      if (rc) return rc;
    }
    PL_HIP(hipMemcpyAsync(h_hist.data(), c->hist.p + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    PL_HIP(hipStreamSynchronize(c->stream));
    for (int j = 0; j < todo; ++j) {
      const double rr = h_hist[j];
      if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
      if (!st->converged) st->rel_residual = std::sqrt(rr / bb);
      if (rr <= thresh && !st->converged) {
        st->converged = 1;
        st->iterations = k + j;      // updates applied when this residual was reached (x has had a few more since)
        st->info = 0.0;
        st->stop_reason = 0.0;
      }
    }
    k += todo;
    if (st->converged) break;
    if (adaptive) {
      const double rr_end = h_hist[todo - 1];
      next = chunk;
      if (rr_end < rr_prev && rr_end > thresh) {
        const double per_it = std::log(rr_end / rr_prev) / (double)(k - k_prev);
        const double need_it = std::log(thresh / rr_end) / per_it;
        if (need_it < 2.0 * chunk) next = std::max(2, std::min(chunk, (int)std::ceil(0.75 * need_it) + 1));
      }
      rr_prev = rr_end;
      k_prev = k;
    }
  }
  if (!st->converged) st->iterations = std::min(k, max_iter);
#undef PL_CG1_UPD
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// fp32 solver modes (opts.precision; multi-level PCG on the tile kernel only):
//   1  inner PCG on fp32-stored x, r, p, Ap; the fp64 solution accumulates the inner corrections and every restart
//      begins from the TRUE fp64 residual P(f - K(ubar + x)) (classical iterative refinement);
//   2  only the search direction p and K*p are stored in fp32, x and the residual recurrence stay fp64: no restart
//      is needed to reach fp64 accuracy, the true residual is verified once the recurrence says "converged" (the
//      fp32 rounding of K*p lets the two drift apart by ~6e-8 of the accumulated steps).
// In both modes every product and sum is evaluated in fp64 (pl_tile.h, pl_coarse.h): fp32 only halves the bytes.
// ----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(pl::kBlock) void k_mp_true_residual(int64_t n6, const double *__restrict__ f,
                                                                const double *__restrict__ Kubar,
                                                                const double *__restrict__ Kx /* may be null */,
                                                                const uint8_t *__restrict__ fixed,
                                                                const double *__restrict__ w /* may be null */,
                                                                double *__restrict__ r, double *__restrict__ rr_slots) {
  __shared__ double red[pl::kBlock / pl::kWave];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * pl::kBlock) {
    const double v = fixed[i] ? 0.0 : f[i] - Kubar[i] - (Kx ? Kx[i] : 0.0);
    r[i] = v;
    acc += (w ? w[i] : 1.0) * v * v;
  }
  const double t = pl::block_sum(acc, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(rr_slots + (blockIdx.x & (pl::kSlots - 1)), t);
}
// start of an fp32 inner solve: r32 = r, x32 = 0
__global__ __launch_bounds__(pl::kBlock) void k_mp_restart(int64_t n6, const double *__restrict__ r,
                                                          float *__restrict__ r32, float *__restrict__ x32) {
  for (int64_t i = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * pl::kBlock) {
    r32[i] = (float)r[i];
    x32[i] = 0.f;
  }
}
__global__ __launch_bounds__(pl::kBlock) void k_mp_accumulate(int64_t n6, const float *__restrict__ x32,
                                                             double *__restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * pl::kBlock)
    x[i] += (double)x32[i];
}

int read_slots(pl_context *c, const double *dev, double *sum) {
  double h[pl::kSlots];
  PL_HIP(hipMemcpyAsync(h, dev, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  *sum = 0.0;
  for (int k = 0; k < pl::kSlots; ++k) *sum += h[k];
  return PL_OK;
}

template <typename RT>
int pcg_solve_mp_t(pl_context *c, const double *f_dev, const double *Kubar_dev, double rtol, int max_iter,
                   pl_stats_t *st) {
  constexpr bool kAll32 = sizeof(RT) == 4;
  const int64_t n6 = c->N * 6;
  const int set = pl::S_COUNT * pl::kSlots;
  int rc = ensure_hist(c, max_iter + 2);
  if (rc) return rc;
  float *p32 = reinterpret_cast<float *>(c->p.p), *Ap32 = reinterpret_cast<float *>(c->Ap.p);
  // mode 1: the inner iterate / residual live in the (otherwise unused) z buffer
  RT *xi = kAll32 ? reinterpret_cast<RT *>(c->z.p) : reinterpret_cast<RT *>(c->x.p);
  RT *ri = kAll32 ? reinterpret_cast<RT *>(c->z.p) + n6 : reinterpret_cast<RT *>(c->r.p);
  const double *w = c->dist.active ? (const double *)c->dist.weight.p : (const double *)nullptr;
  double *aux = c->scal.p + pl::S_AUX * pl::kSlots;   // slots outside the two per-parity sets' live entries
  PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * set * sizeof(double), c->stream));
  PL_HIP(hipMemsetAsync(c->x.p, 0, n6 * sizeof(double), c->stream));
  hipLaunchKernelGGL(k_mp_true_residual, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                     (const double *)nullptr, c->fixed.p, w, c->r.p, aux);
  PL_HIP(hipGetLastError());
  if (c->dist.active && pl::dist_sum_scalars(c->dist, aux, pl::kSlots, c->stream))
    return fail(PL_ERR_HIP, "RCCL all-reduce of ||b||^2 failed");
  double bb = 0.0;
  rc = read_slots(c, aux, &bb);
  if (rc) return rc;
  st->b_norm = std::sqrt(bb);
  st->iterations = 0;
  st->converged = 0;
  st->rel_residual = 0.0;
  if (!(bb > 0.0)) {
    st->converged = 1;
    return std::isnan(bb) ? fail(PL_ERR_NAN, "NaN in the right-hand side") : PL_OK;
  }
  const double thresh = rtol * rtol * bb;
  // an fp32 residual recurrence is trustworthy over ~4 decades: restart from the true residual after that
  static const double drop_env = [] { const char *e = std::getenv("PL_MP_DROP"); return e ? std::atof(e) : 0.0; }();
  const double inner_drop = kAll32 ? (drop_env > 0.0 ? drop_env : 1e-8) : 0.0;      // on ||r||^2
  // (measured, tools/experiments/mp_drop_sweep.sh, iterations at 50^3 Octet / 100^3 Octet / configs[2] / configs[4]:
  //  fixed 8 decades per stage 141 / 151 / 392 / 324;  equal stages of at most 5: 136 / 146 / 352 / 293,  6: 130 / 144 / 358 / 298,
  //  7: 130 / 144 / 357 / 298,  8: 130 / 147 / 393 / 354;  fp64: 120 / 137 / 335 / -)
  static const double stage_max = [] { const char *e = std::getenv("PL_MP_STAGE"); return e ? std::atof(e) : 6.0; }();
  double rr_true = bb;
  // warm start (mode 1, as pcg_solve): the refinement begins at the previous solution of this handle, masked with the
  // CURRENT Dirichlet set - the first inner solve then works on its true residual like every later one
  const bool warm = kAll32 && c->opt.warm_start >= 1 && !c->dist.active;     // (2 / 3: the previous solution here, no extrapolation)
  if (warm && c->xprev.p && c->xprev_valid) {
    PL_HIP(hipMemcpyAsync(c->x.p, c->xprev.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_warm_mask, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N, (const uint8_t *)nullptr,
                       (const uint8_t *)c->fixed.p, c->x.p);
    rc = launch_spmv(c, c->x.p, c->tmp2.p, true, nullptr);
    if (rc) return rc;
    PL_HIP(hipMemsetAsync(aux, 0, pl::kSlots * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_mp_true_residual, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       (const double *)c->tmp2.p, c->fixed.p, w, c->r.p, aux);
    PL_HIP(hipGetLastError());
    rc = read_slots(c, aux, &rr_true);
    if (rc) return rc;
    if (std::isnan(rr_true) || std::isinf(rr_true)) return fail(PL_ERR_NAN, "NaN/Inf in the warm-start residual");
    st->rel_residual = std::sqrt(rr_true / bb);
    if (rr_true <= thresh) {       // (the previous solution already solves this system)
      st->converged = 1;
      return PL_OK;
    }
  }
  int k = 0;                      // iterations over all inner solves
  std::vector<double> h_hist(32);
  for (int outer = 0; outer < 40 && k < max_iter; ++outer) {
    // ---- (re)start: p = M^-1 r through the tail of an iteration "-1" (alpha = 0) on scalar set 1
    PL_HIP(hipMemsetAsync(c->scal.p, 0, 2 * set * sizeof(double), c->stream));
    if (kAll32)
      hipLaunchKernelGGL(k_mp_restart, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, c->r.p,
                         reinterpret_cast<float *>(ri), reinterpret_cast<float *>(xi));
    PL_HIP(hipMemsetAsync(p32, 0, n6 * sizeof(float), c->stream));
    if (kAll32 && c->cond_use) {
      // node elimination inside the inner solve (as in pcg_solve): t_c = K_cc^-1 b_c in the rows of p32, r_v -= (K t)_v
      float *r32 = reinterpret_cast<float *>(ri);
      hipLaunchKernelGGL(pl::k_condense_solve<float>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                         c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const float *)r32, p32, 1.0);
      rc = launch_spmv_f32(c, p32, Ap32, true, nullptr, c->maskC.p, pl::kEndsOthers);
      if (rc) return rc;
      hipLaunchKernelGGL(pl::k_condense_subtract<float>, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N,
                         c->cflag.p, (const float *)Ap32, r32);
      PL_HIP(hipMemsetAsync(p32, 0, n6 * sizeof(float), c->stream));
    }
    PL_HIP(hipMemsetAsync(Ap32, 0, n6 * sizeof(float), c->stream));
    rc = pcg_tail_coarse_t<float, RT>(c, c->scal.p + set, c->scal.p, max_iter + 1, p32, (const float *)Ap32, xi, ri);
    if (rc) return rc;
    // Stages of EQUAL depth: what is left to the threshold (on ||r||^2) is cut into the fewest stages of at most
    // `stage_max` decades - an fp32 residual recurrence that has dropped further has drifted from the true residual, and the
    // iterations of a stage that ends below the threshold by accident are wasted (PL_MP_DROP: a fixed drop per stage instead)
    double stop = std::max(thresh, inner_drop * rr_true);
    if (kAll32 && !(drop_env > 0.0) && rr_true > thresh) {
      const double left = std::log10(rr_true / thresh);
      const int stages = std::max(1, (int)std::ceil(left / stage_max));
      stop = stages == 1 ? thresh : std::max(thresh, rr_true * std::pow(10.0, -left / stages));
    }
    bool inner_done = false;
    int j = 0, next = 32;
    double rr_prev = rr_true;
    int j_prev = 0;
    while (!inner_done && k < max_iter) {
      const int todo = std::min(next, max_iter - k);
      for (int q = 0; q < todo; ++q) {
        double *cur = c->scal.p + ((j + q) & 1) * set, *nxt = c->scal.p + ((j + q + 1) & 1) * set;
        if (kAll32 && c->cond_use) {
          rc = launch_spmv_f32(c, p32, p32, false, nullptr, nullptr, pl::kEndsCondensedSolve);
          if (rc) return rc;
          rc = launch_spmv_f32(c, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots, c->maskC.p, pl::kEndsOthers);
        } else {
          rc = launch_spmv_f32(c, p32, Ap32, true, cur + pl::S_PAP * pl::kSlots);
        }
        if (rc) return rc;
        rc = pcg_tail_coarse_t<float, RT>(c, cur, nxt, k + q, p32, (const float *)Ap32, xi, ri);
        if (rc) return rc;
      }
      PL_HIP(hipMemcpyAsync(h_hist.data(), c->hist.p + k, todo * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipStreamSynchronize(c->stream));
      int used = todo;
      for (int q = 0; q < todo; ++q) {
        const double rr = h_hist[q];
        if (std::isnan(rr) || std::isinf(rr)) return fail(PL_ERR_NAN, "NaN/Inf in the PCG residual");
        if (rr <= stop) { inner_done = true; used = q + 1; break; }
      }
      // (the device has run the whole chunk: x holds the iterate after `todo` iterations, which is what is kept)
      const double rr_end = h_hist[todo - 1];
      j += todo;
      k += todo;
      (void)used;
      if (!inner_done) {
        next = 32;
        if (rr_end < rr_prev && rr_end > stop) {
          const double per_it = std::log(rr_end / rr_prev) / (double)(j - j_prev);
          const double need = std::log(stop / rr_end) / per_it;
          if (need < 64.0) next = std::max(2, std::min(32, (int)std::ceil(0.75 * need)));
        }
        rr_prev = rr_end;
        j_prev = j;
      }
    }
    // ---- true residual of the accumulated solution
    if (kAll32 && c->cond_use) {   // the eliminated nodes of this inner solve: x_c = K_cc^-1 (b_c - (K [x_v ; 0])_c)
      float *x32 = reinterpret_cast<float *>(xi), *r32 = reinterpret_cast<float *>(ri);
      rc = launch_spmv_f32(c, x32, Ap32, false, nullptr, nullptr, pl::kEndsCondensed);
      if (rc) return rc;
      hipLaunchKernelGGL(pl::k_condense_backsubst<float>, dim3(grid_for(c->n_cond * 6)), dim3(pl::kBlock), 0, c->stream,
                         c->n_cond, c->cnodes.p, cinv(c), ccls(c), (const float *)r32, (const float *)Ap32, x32);
    }
    if (kAll32)
      hipLaunchKernelGGL(k_mp_accumulate, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6,
                         reinterpret_cast<const float *>(xi), c->x.p);
    rc = launch_spmv(c, c->x.p, c->tmp2.p, true, nullptr);
    if (rc) return rc;
    PL_HIP(hipMemsetAsync(aux, 0, pl::kSlots * sizeof(double), c->stream));
    hipLaunchKernelGGL(k_mp_true_residual, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6, f_dev, Kubar_dev,
                       (const double *)c->tmp2.p, c->fixed.p, w, c->r.p, aux);
    PL_HIP(hipGetLastError());
    if (c->dist.active && pl::dist_sum_scalars(c->dist, aux, pl::kSlots, c->stream))
      return fail(PL_ERR_HIP, "RCCL all-reduce of the true residual failed");
    rc = read_slots(c, aux, &rr_true);
    if (rc) return rc;
    if (std::isnan(rr_true) || std::isinf(rr_true)) return fail(PL_ERR_NAN, "NaN/Inf in the true residual");
    st->rel_residual = std::sqrt(rr_true / bb);
    st->restarts = (double)(outer + 1);     // restarts (inner solves) taken
    if (rr_true <= thresh * 1.0000001) {
      st->converged = 1;
      break;
    }
  }
  st->iterations = k;
  if (warm && st->converged) {
    if (!c->xprev.p) PL_HIP(c->xprev.alloc(n6));
    PL_HIP(hipMemcpyAsync(c->xprev.p, c->x.p, n6 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->xprev_valid = true;
  }
  return PL_OK;
}

}  // namespace
