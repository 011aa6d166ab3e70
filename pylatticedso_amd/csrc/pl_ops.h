// Operator launches of libpylattice_hip: y = K x in its three kernel variants, the DDM operator, the fp32-stored
// variant, plus the multi-GPU interface sum behind them (device vectors, device numbering).
#pragma once
#include "pl_context.h"

namespace {

// ----------------------------------------------------------------------------------------------------------
// operator launches (device vectors, device numbering)
// ----------------------------------------------------------------------------------------------------------
// auto: the LDS-tile kernel when the nodes are brick-ordered (its tiles are then compact), else the per-node gather
int choose_kernel(const pl_context *c) {
  if (c->opt.spmv_kernel != 0) return c->opt.spmv_kernel;
  return (c->reordered && c->tile.ready) ? 3 : 2;
}

template <int LPN>
void launch_gather_lpn(pl_context *c, const double *x, double *y, bool masked, double *dot_dev) {
  const unsigned g = grid_for(c->n_slices, pl::kBlock / pl::kWave);   // one wave per ELL slice
#define PL_G(M, D)                                                                                               \
  hipLaunchKernelGGL((pl::k_spmv_gather<LPN, M, D>), dim3(g), dim3(pl::kBlock), 0, c->stream, c->N, c->slice_ptr.p, \
                     c->ent.p, c->rec.p, c->fixedbits.p, x, y, dot_dev)
  if (masked && dot_dev) PL_G(true, true);
  else if (masked) PL_G(true, false);
  else if (dot_dev) PL_G(false, true);
  else PL_G(false, false);
#undef PL_G
}

int dispatch_gather(pl_context *c, const double *x, double *y, bool masked, double *dot_dev) {
  switch (c->lpn) {
    case 1: launch_gather_lpn<1>(c, x, y, masked, dot_dev); break;
    case 2: launch_gather_lpn<2>(c, x, y, masked, dot_dev); break;
    case 4: launch_gather_lpn<4>(c, x, y, masked, dot_dev); break;
    case 8: launch_gather_lpn<8>(c, x, y, masked, dot_dev); break;
    case 16: launch_gather_lpn<16>(c, x, y, masked, dot_dev); break;
    default: return fail(PL_ERR_ARG, "lanes per node must be 1, 2, 4, 8 or 16");
  }
  return PL_OK;
}

// y = K x (masked -> y = P K x, x assumed zero on fixed dofs); optional dot(x, y) accumulated into *dot_dev.
// K_cc^-1 for the fused first pass of the condensed operator (kEndsCondensedSolve)
__global__ __launch_bounds__(pl::kBlock) void k_cond_ends(int64_t B, const int2 *__restrict__ conn2,
                                                         const uint8_t *__restrict__ cflag, uint8_t *__restrict__ cend) {
  const int64_t b = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (b >= B) return;
  const int2 c = conn2[b];
  cend[b] = (uint8_t)((cflag[c.x] ? 1 : 0) | (cflag[c.y] ? 2 : 0));
}
__global__ __launch_bounds__(pl::kBlock) void k_cond_base(int64_t nc, const int32_t *__restrict__ cnodes,
                                                         const uint16_t *__restrict__ cls /* may be null */,
                                                         int32_t *__restrict__ base) {
  const int64_t q = (int64_t)blockIdx.x * pl::kBlock + threadIdx.x;
  if (q < nc) base[cnodes[q]] = 36 * (cls ? (int32_t)cls[q] : (int32_t)q);
}
inline pl::CondSolve cond_solve(pl_context *c, int ends) {
  pl::CondSolve cs;
  if (ends != pl::kEndsAll) cs.cend = c->cend.p;
  if (ends == pl::kEndsCondensedSolve) {
    const int want = c->cls_ready ? 1 : 0;
    if (c->cbase_state != want) {   // (after pl_set_bc / when the class table comes or goes: once per assembly at most)
      (void)hipMemsetAsync(c->cbase.p, 0xFF, (size_t)c->N * sizeof(int32_t), c->stream);
      hipLaunchKernelGGL(k_cond_base, dim3(grid_for(c->n_cond)), dim3(pl::kBlock), 0, c->stream, c->n_cond, c->cnodes.p,
                         c->cls_ready ? (const uint16_t *)c->cls_id.p : (const uint16_t *)nullptr, c->cbase.p);
      c->cbase_state = want;
    }
    cs.inv = c->cls_ready ? (const double *)c->cls_table.p : (const double *)c->kcc_inv.p;
    cs.base = c->cbase.p;
  }
  return cs;
}

// Several GPUs: every rank holds the product of ITS struts.  The Dirichlet mask commutes with the sum over ranks, and
// x.(K x) = sum_r x_r.(K_r x_r) with the LOCAL partial products and NO multiplicity weights, so the K*x kernels run
// exactly as on one GPU; the interface rows and the slots of the partial dot then travel in one exchange.
// (reduce_dot = false: the caller sums the dot slots in a collective of its own - single-reduction PCG)
template <typename VT>
int spmv_interface_sum(pl_context *c, VT *y, double *dot_dev, bool reduce_dot = true) {
  if (!c->dist.active) return PL_OK;
  const bool with_dot = dot_dev && reduce_dot;
  int rc = pl::dist_sum_shared<VT>(c->dist, y, c->stream, with_dot ? dot_dev : nullptr, with_dot ? pl::kSlots : 0);
  if (rc) return fail(PL_ERR_HIP, "exchange of the interface forces failed (" + std::to_string(rc) + ")");
  return PL_OK;
}

// The LDS-tile K*x from whichever record source the handle has (palette ids / compact records / 64-byte records), over
// all tiles or over a list of them.
template <typename VT>
void tile_launch(pl_context *c, const uint8_t *maskbits, const VT *x, VT *y, double *dot_dev, int ends, const uint8_t *cf,
                 const pl::CondSolve &cs, const int32_t *list = nullptr, int64_t n_list = 0) {
  static const bool lds_off = [] { const char *e = std::getenv("PL_TILE_LDS"); return e && e[0] == '0'; }();
  if (c->pal_rows &&
      pl::launch_spmv_rows<VT>(c->rows, c->rword.p, c->pal_dense2.p, c->pal_entries, maskbits, x, y, dot_dev, c->stream,
                               ends, cf, cs, list, n_list))
    return;
  if (c->pal_lds &&
      pl::launch_tile_spmv_lds<VT>(c->tile, c->vword.p, c->pal_dense.p, c->pal_entries, nullptr, maskbits, x, y, dot_dev,
                                   c->stream, ends, cf, cs, list, n_list))
    return;
  if (!c->pal_ready && c->rec5.p && c->vword_dir.p && c->vword_dir_fresh && !lds_off &&
      pl::launch_tile_spmv_lds<VT>(c->tile, c->vword_dir.p, c->tile.dir_table.p, c->tile.n_dir,
                                   reinterpret_cast<const pl::Rec5 *>(c->rec5.p), maskbits, x, y, dot_dev, c->stream, ends,
                                   cf, cs, list, n_list))
    return;
  if (c->pal_ready)
    pl::launch_tile_spmv<VT>(c->tile, c->conn.p, c->palette.p, c->pal_id.p, maskbits, x, y, dot_dev, c->stream,
                             (const double *)nullptr, ends, cf, cs, list, n_list);
  else if (c->rec5.p)
    pl::launch_tile_spmv<VT>(c->tile, c->conn.p, reinterpret_cast<const pl::Record *>(c->rec5.p), nullptr, maskbits, x, y,
                             dot_dev, c->stream, c->xyz.p, ends, cf, cs, list, n_list);
  else
    pl::launch_tile_spmv<VT>(c->tile, c->conn.p, c->rec.p, nullptr, maskbits, x, y, dot_dev, c->stream,
                             (const double *)nullptr, ends, cf, cs, list, n_list);
}

// Which K*x kernel the handle's operator runs (pl_stats_t.kp_form; same order of tests as tile_launch / launch_spmv)
int kp_form_of(const pl_context *c) {
  static const bool lds_off = [] { const char *e = std::getenv("PL_TILE_LDS"); return e && e[0] == '0'; }();
  if (c->opkind == 1) return 7;
  const int kind = choose_kernel(c);
  if (kind == 1) return 6;
  if (!(kind == 3 && c->tile.ready)) return 5;
  if (c->pal_rows) return 4;
  if (c->pal_lds) return 1;
  if (!c->pal_ready && c->rec5.p && c->vword_dir.p && c->vword_dir_fresh && !lds_off) return 2;
  return 3;
}

// Tile K*x of a handle, with the interface sum of a multi-GPU handle behind it.
// Several GPUs: an eliminated node is never shared with another rank, so all its struts are this rank's and the passes that
// accumulate ONLY the strut ends at eliminated nodes (kEndsCondensed / kEndsCondensedSolve) are complete locally; the pass
// over the other ends is an ordinary partial product and takes the interface exchange.
// Overlap (neighbour exchange only): the tiles that own interface rows run first; their rows are packed, exchanged and
// added on the communication stream while the interior tiles run on the main stream; the scalar slots of the fused dot
// product (to which both launches add) take their small all-reduce afterwards.
template <typename VT>
int tile_spmv(pl_context *c, const uint8_t *maskbits, const VT *x, VT *y, double *dot_dev, int ends, bool reduce_dot) {
  const uint8_t *cf = ends != pl::kEndsAll ? (const uint8_t *)c->cflag.p : (const uint8_t *)nullptr;
  const pl::CondSolve cs = cond_solve(c, ends);
  const bool exchange = c->dist.active && (ends == pl::kEndsAll || ends == pl::kEndsOthers);
  if (!(exchange && c->ov_ready && c->dist.p2p)) {
    tile_launch<VT>(c, maskbits, x, y, dot_dev, ends, cf, cs);
    PL_HIP(hipGetLastError());
    return exchange ? spmv_interface_sum<VT>(c, y, dot_dev, reduce_dot) : PL_OK;
  }
  tile_launch<VT>(c, maskbits, x, y, dot_dev, ends, cf, cs, c->ov_iface.p, c->n_ov_iface);
  PL_HIP(hipEventRecord(c->ev_ov_a, c->stream));
  PL_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_ov_a, 0));
  if (pl::dist_exchange_p2p<VT>(c->dist, y, c->comm_stream)) return fail(PL_ERR_HIP, "exchange of the interface forces failed");
  PL_HIP(hipEventRecord(c->ev_ov_x, c->comm_stream));
  tile_launch<VT>(c, maskbits, x, y, dot_dev, ends, cf, cs, c->ov_inner.p, c->n_ov_inner);
  PL_HIP(hipStreamWaitEvent(c->stream, c->ev_ov_x, 0));
  PL_HIP(hipGetLastError());
  if (dot_dev && reduce_dot && pl::dist_sum_scalars(c->dist, dot_dev, pl::kSlots, c->stream))
    return fail(PL_ERR_HIP, "all-reduce of the K*p dot-product slots failed");
  return PL_OK;
}

int launch_spmv(pl_context *c, const double *x, double *y, bool masked, double *dot_dev,
                const uint8_t *maskbits = nullptr, int ends = pl::kEndsAll, bool reduce_dot = true) {
  const int kind = choose_kernel(c);
  if ((maskbits || ends != pl::kEndsAll) && kind == 3 && c->tile.ready && c->opkind == 0)
    // tile kernel with a caller-chosen row mask and / or only one kind of strut ends (node elimination)
    return tile_spmv<double>(c, maskbits, x, y, dot_dev, ends, reduce_dot);
  const int64_t n6 = c->N * 6;
  if (c->opkind == 1) {
    const int m = 6 * c->ddm_nb;
    const size_t lds = ((size_t)m * m + (size_t)(pl::kBlock / pl::kWave) * m) * sizeof(double);
    const unsigned gw = grid_for((c->ddm_cells + pl::kDdmWaveChunk - 1) / pl::kDdmWaveChunk, pl::kBlock / pl::kWave);
    // (read at every launch, so that one process can compare the matrix-pipe kernels with the generic ones: A/B runs, tests)
    const char *mfma_env = std::getenv("PL_DDM_MFMA");
    const bool mfma_off = mfma_env && mfma_env[0] == '0';
    const int64_t tile_waves = (c->ddm_n_tiles + pl::kDdmTilesPerWave - 1) / pl::kDdmTilesPerWave;
    // (KS, NBW): K steps and 16-column blocks per wave; a tile's ceil(m / 16) column blocks are dealt over `slices` waves
#define PL_DM(KS, NBW)                                                                                                   \
  do {                                                                                                                   \
    const int slices = ((m + 15) / 16 + (NBW) - 1) / (NBW);                                                              \
    hipLaunchKernelGGL((pl::k_ddm_cell_product_mfma<KS, NBW>), dim3(grid_for(tile_waves * slices, pl::kBlock / pl::kWave)), \
                       dim3(pl::kBlock), 0, c->stream, c->ddm_n_tiles, c->ddm_nb, c->ddm_tiles.p, c->ddm_tile_S.p,        \
                       c->ddm_tile_gidx.p, c->ddm_St.p, x, c->ddm_stage.p, slices);                                       \
  } while (0)
    const bool mfma_ok = !mfma_off && c->ddm_n_tiles > 0 && c->ddm_tile_gidx.p;
    if (mfma_ok && m <= 32)
      PL_DM(8, 2);
    else if (mfma_ok && m <= 48)
      PL_DM(12, 3);
    else if (mfma_ok && m <= 72)       // Hybrid1 (12 boundary nodes): two slices of 3 column blocks
      PL_DM(18, 3);
    else if (mfma_ok && m <= 96)
      PL_DM(24, 2);
    else if (mfma_ok && m <= 120)
      PL_DM(30, 2);
    else if (mfma_ok && m <= 156)      // the reference's BCC + Hybrid1 (+ Hybrid4) hybrids: 26 boundary nodes, five slices
      PL_DM(39, 2);
    else if (mfma_ok && m <= 192)
      PL_DM(48, 1);
    else if (m <= 48)
      hipLaunchKernelGGL(pl::k_ddm_cell_product_reg<48>, dim3(gw), dim3(pl::kBlock), 0, c->stream, c->ddm_cells,
                         c->ddm_nb, c->ddm_order.p, c->ddm_cell_nodes.p, c->ddm_cell_S.p, c->ddm_St.p, x, c->ddm_stage.p);
    else if (m <= 64)
      hipLaunchKernelGGL(pl::k_ddm_cell_product_reg<64>, dim3(gw), dim3(pl::kBlock), 0, c->stream, c->ddm_cells,
                         c->ddm_nb, c->ddm_order.p, c->ddm_cell_nodes.p, c->ddm_cell_S.p, c->ddm_St.p, x, c->ddm_stage.p);
    else if (lds <= 64 * 1024)
      hipLaunchKernelGGL(pl::k_ddm_cell_product_lds, dim3(grid_for(c->ddm_cells, pl::kDdmChunk)), dim3(pl::kBlock), lds,
                         c->stream, c->ddm_cells, c->ddm_nb, c->ddm_order.p, c->ddm_cell_nodes.p, c->ddm_cell_S.p,
                         c->ddm_St.p, x, c->ddm_stage.p);
    else
      hipLaunchKernelGGL(pl::k_ddm_cell_product, dim3(grid_for(c->ddm_cells, pl::kBlock / pl::kWave)), dim3(pl::kBlock),
                         0, c->stream, c->ddm_cells, c->ddm_nb, c->ddm_cell_nodes.p, c->ddm_cell_S.p, c->ddm_St.p, x,
                         c->ddm_stage.p);
#undef PL_DM
    hipLaunchKernelGGL(pl::k_ddm_node_gather, dim3(grid_for(n6)), dim3(pl::kBlock), 0, c->stream, c->N,
                       c->ddm_node_ptr.p, c->ddm_node_ent.p, (const double *)c->ddm_stage.p, y,
                       masked ? (const uint8_t *)c->fixed.p : (const uint8_t *)nullptr, x, dot_dev);
    PL_HIP(hipGetLastError());
    return PL_OK;
  }
  if (kind == 1) {
    PL_HIP(hipMemsetAsync(y, 0, n6 * sizeof(double), c->stream));
    hipLaunchKernelGGL(pl::k_spmv_atomic, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B, c->conn.p,
                       c->rec.p, x, y);
    if (masked || dot_dev)
      hipLaunchKernelGGL(pl::k_mask_dot, dim3(grid_stream(n6)), dim3(pl::kBlock), 0, c->stream, n6,
                         masked ? c->fixed.p : (const uint8_t *)nullptr, x, y, dot_dev);
  } else if (kind == 3 && c->tile.ready) {
    return tile_spmv<double>(c, masked ? (const uint8_t *)c->fixedbits.p : (const uint8_t *)nullptr, x, y, dot_dev,
                             pl::kEndsAll, reduce_dot);
  } else {
    int rc = dispatch_gather(c, x, y, masked, dot_dev);
    if (rc) return rc;
  }
  PL_HIP(hipGetLastError());
  return spmv_interface_sum(c, y, dot_dev, reduce_dot);
}

// K*p of the fp32 solver modes: tile kernel only, fp32-stored x / y, fp64 arithmetic (pl_tile.h)
int launch_spmv_f32(pl_context *c, const float *x, float *y, bool masked, double *dot_dev,
                    const uint8_t *maskbits = nullptr, int ends = pl::kEndsAll) {
  const uint8_t *mk = maskbits ? maskbits : (masked ? (const uint8_t *)c->fixedbits.p : (const uint8_t *)nullptr);
  return tile_spmv<float>(c, mk, x, y, dot_dev, ends, true);
}

// residual history + (reference-CG mode) direction norm, solution norm and step length of every iteration
int ensure_hist(pl_context *c, int cap) {
  if (cap <= c->hist_cap) return PL_OK;
  PL_HIP(c->hist.alloc((size_t)cap * 4));
  c->hist_cap = cap;
  return PL_OK;
}
// the reference's CG extras are on when the caller asked for any of them (pl_opts_t.mintol / restart_every)
inline bool ref_cg(const pl_context *c) { return c->opt.mintol > 0.0 || c->opt.restart_every > 0; }

}  // namespace
