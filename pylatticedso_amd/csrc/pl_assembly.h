// Assembly stages of libpylattice_hip: strut records, record palette, Jacobi diagonal, tile blocks, the dense coarse
// level(s) and their factorisation, node elimination blocks, BSR fill; host-side incidence / BSR pattern.
#pragma once
#include "pl_ops.h"

namespace {

// Short form of the PCG iteration (pl_small.h): what can be decided from the options and the sizes alone (the assembly
// builds the explicit inverse of the dense level for it).  Every tile reads cm rows of A_c^-1 per iteration: automatic only
// while that stays below ~16 MB per iteration (BASELINE configs[3]: 128 tiles x 12 x 1 536 x 4 B = 9.4 MB).
inline bool small_wanted(const pl_context *c) {
  const pl::Coarse &cs = c->coarse;
  if (c->opt.short_iteration < 0 || c->opkind != 0 || c->dist.active || c->coarseL.enabled || c->opt.precision != 0 ||
      c->opt.cg_form == 1 || !cs.enabled || !c->tile.ready || choose_kernel(c) != 3 || c->opt.mintol > 0.0 ||
      c->opt.restart_every > 0)
    return false;
  if (c->opt.short_iteration == 1) return true;
  static const double auto_mb = [] { const char *e = std::getenv("PL_SHORT_ITERATION_MB"); return e ? std::atof(e) : 16.0; }();
  return (double)cs.n_tiles * cs.cm * cs.ncp * 4.0 <= auto_mb * 1048576.0;
}


int launch_records(pl_context *c) {
  hipLaunchKernelGGL(pl::k_build_records, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B, c->xyz.p,
                     c->conn.p, c->radius.p, c->seg_len.p, c->seg_nsub.p, c->mult.p, c->mat, c->rec.p, c->rec5.p);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// Try to replace the per-strut records by palette ids (periodic lattices); leaves pal_ready = false otherwise.
// Visit words of the LDS-resident K*p (pl_tile.h): static local rows | dense palette id | condensed-end bits.  Rebuilt
// behind every palette build and whenever the set of eliminated nodes changes (needs the palette's dense ids: a no-op
// before the first assembly, whose own palette build then picks the condensed-end bits up).
void refresh_visit_words(pl_context *c, hipStream_t st) {
  if (c->rword.p && c->pal_dense_of_slot.p)      // (the row words carry no condensed-end bits: rows are selected by node)
    hipLaunchKernelGGL(pl::k_row_words, dim3(grid_for(c->rows.n_words)), dim3(pl::kBlock), 0, st, c->rows.n_words,
                       c->rows.rloc.p, c->rows.rstrut.p, c->pal_id.p, c->pal_dense_of_slot.p, c->rword.p);
  if (!c->vword.p || !c->pal_dense_of_slot.p) return;
  hipLaunchKernelGGL(pl::k_visit_words, dim3(grid_for(c->tile.n_visits)), dim3(pl::kBlock), 0, st, c->tile.n_visits,
                     c->tile.vloc.p, c->tile.vstrut.p, c->pal_id.p, c->pal_dense_of_slot.p,
                     c->cend.p ? (const uint8_t *)c->cend.p : (const uint8_t *)nullptr, c->vword.p);
}
// The streaming form's words (direction-palette entry instead of the record-palette id): static but for the condensed-end
// bits, so rebuilt only when the set of eliminated nodes changes (and once at the first assembly).
int refresh_visit_words_dir(pl_context *c, hipStream_t st) {
  if (!c->tile.vis_ready || c->tile.n_dir <= 0 || !c->rec5.p) return PL_OK;
  if (!c->vword_dir.p) PL_HIP(c->vword_dir.alloc((size_t)c->tile.n_visits));
  hipLaunchKernelGGL(pl::k_visit_words, dim3(grid_for(c->tile.n_visits)), dim3(pl::kBlock), 0, st, c->tile.n_visits,
                     c->tile.vloc.p, c->tile.vstrut.p, (const uint16_t *)nullptr, (const int *)nullptr,
                     c->cend.p ? (const uint8_t *)c->cend.p : (const uint8_t *)nullptr, c->vword_dir.p);
  c->vword_dir_fresh = true;
  return PL_OK;
}

// launch_palette queues the kernels and the flag read-back on `st`; finish_palette (after a sync) reads the verdict.
int launch_palette(pl_context *c, hipStream_t st) {
  c->pal_ready = false;
  c->pal_lds = false;
  c->pal_rows = false;
  c->pal_host_flags[0] = 1;
  c->pal_host_flags[1] = 0;
  if (!c->vword_dir_fresh) {      // (first assembly of a handle without node elimination)
    int rcw = refresh_visit_words_dir(c, st);
    if (rcw) return rcw;
  }
  if (!c->opt.palette) return PL_OK;
  // A design loop on a graded lattice fails this attempt at every assembly (48 k distinct radii at configs[3]: insert + verify
  // = 87 us of a 355-us assembly).  After a failure the next 1, 3, 7, 15 assemblies go without one; the first success resets.
  c->pal_skipped = false;
  if (c->pal_skip_left > 0) {
    c->pal_skip_left--;
    c->pal_skipped = true;
    return PL_OK;
  }
  if (!c->pal_keys.p) {
    PL_HIP(c->pal_keys.alloc(pl::kPalSize));
    PL_HIP(c->pal_owner.alloc(pl::kPalSize));
    PL_HIP(c->pal_flags.alloc(2));
    PL_HIP(c->pal_id.alloc(c->B));
    PL_HIP(c->palette.alloc(pl::kPalSize));
    PL_HIP(c->pal_dense_of_slot.alloc(pl::kPalSize));
    PL_HIP(c->pal_dense.alloc(pl::kPalDenseMax));
    if (c->tile.vis_ready) PL_HIP(c->vword.alloc((size_t)c->tile.n_visits));
    if (c->rows.ready) {
      PL_HIP(c->rword.alloc((size_t)c->rows.n_words));
      PL_HIP(c->pal_dense2.alloc((size_t)2 * pl::kPalDenseMax));
    }
    void *pinned = nullptr;
    PL_HIP(hipHostMalloc(&pinned, 2 * sizeof(int), hipHostMallocDefault));
    c->pal_host_flags = static_cast<int *>(pinned);
    c->pal_host_flags[0] = 1;
    c->pal_host_flags[1] = 0;
  }
  PL_HIP(hipMemsetAsync(c->pal_keys.p, 0xFF, pl::kPalSize * sizeof(unsigned long long), st));
  PL_HIP(hipMemsetAsync(c->pal_owner.p, 0x7F, pl::kPalSize * sizeof(int), st));
  PL_HIP(hipMemsetAsync(c->pal_flags.p, 0, 2 * sizeof(int), st));
  PL_HIP(hipMemsetAsync(c->palette.p, 0, pl::kPalSize * sizeof(pl::Record), st));
  const dim3 g(grid_for(c->B)), blk(pl::kBlock);
  hipLaunchKernelGGL(pl::k_pal_insert, g, blk, 0, st, c->B, c->rec.p, c->pal_keys.p, c->pal_owner.p, c->pal_id.p,
                     c->pal_flags.p);
  hipLaunchKernelGGL(pl::k_pal_publish, g, blk, 0, st, c->B, c->rec.p, c->pal_owner.p, c->pal_id.p, c->palette.p);
  hipLaunchKernelGGL(pl::k_pal_verify, g, blk, 0, st, c->B, c->rec.p, c->pal_id.p, c->palette.p, c->pal_owner.p,
                     c->pal_flags.p, c->pal_dense_of_slot.p, c->pal_dense.p, pl::kPalDenseMax);
  refresh_visit_words(c, st);
  if (c->pal_dense2.p)
    hipLaunchKernelGGL(pl::k_pal_orient, dim3(1), dim3(pl::kPalDenseMax), 0, st, pl::kPalDenseMax, c->pal_dense.p,
                       c->pal_dense2.p);
  PL_HIP(hipGetLastError());
  PL_HIP(hipMemcpyAsync(c->pal_host_flags, c->pal_flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  return PL_OK;
}
void finish_palette(pl_context *c) {
  if (!c->opt.palette) return;
  if (c->pal_skipped) {          // (no attempt was made: the streaming form, as after a failed one)
    c->pal_entries = 0;
    c->pal_ready = false;
    c->pal_lds = false;
    c->pal_rows = false;
    return;
  }
  c->pal_entries = c->pal_host_flags[1];
  c->pal_ready = (c->pal_host_flags[0] == 0);
  // the LDS-resident K*p (pl_tile.h) when the whole palette fits its LDS table; PL_TILE_LDS=0 keeps the gather kernel (A/B)
  static const bool lds_off = [] { const char *e = std::getenv("PL_TILE_LDS"); return e && e[0] == '0'; }();
  c->pal_lds = c->pal_ready && c->vword.p && c->pal_entries > 0 && c->pal_entries <= pl::kPalDenseMax && !lds_off;
  // A palette too large for the LDS table would serve the gather kernel (2-byte ids: the faster form on LARGE lattices).  On a
  // small lattice the launches count, not the bytes: keep the LDS-resident streaming form, which the short iteration
  // (pl_small.h) needs - the design loops of LatticeOpti are exactly this case (a few hundred distinct radii).
  if (c->pal_ready && !c->pal_lds && !lds_off && small_wanted(c) && c->rec5.p && c->vword_dir.p && c->tile.n_dir > 0 &&
      c->tile.n_dir <= pl::kPalDenseMax)
    c->pal_ready = false;
  if (c->pal_ready) {            // (an attempt whose palette is not USED counts as failed too)
    c->pal_fail_streak = 0;
  } else {
    c->pal_fail_streak = std::min(c->pal_fail_streak + 1, 4);
    c->pal_skip_left = (1 << c->pal_fail_streak) - 1;
  }
  // the row kernel (pl_rows.h): opt-in with PL_ROWS=1 (read at every assembly, so that one process can compare both) -
  // measured slower than the tile kernel at 50^3 Octet (51 against 36 us: see the header of pl_rows.h)
  const char *re = std::getenv("PL_ROWS");
  c->pal_rows = c->pal_lds && c->rword.p && c->pal_dense2.p && re && re[0] == '1';
}
int build_palette(pl_context *c) {
  int rc = launch_palette(c, c->stream);
  if (rc) return rc;
  PL_HIP(hipStreamSynchronize(c->stream));
  finish_palette(c);
  return PL_OK;
}

int launch_diag(pl_context *c, hipStream_t st) {
  const unsigned g = grid_for(c->n_slices, pl::kBlock / pl::kWave);
  const uint8_t *fb = c->have_bc ? c->fixedbits.p : (const uint8_t *)nullptr;
#define PL_D(L)                                                                                                  \
  hipLaunchKernelGGL((pl::k_diag_gather<L>), dim3(g), dim3(pl::kBlock), 0, st, c->N, c->slice_ptr.p, c->ent.p, \
                     c->rec.p, fb, c->diag.p, c->dinv.p)
  switch (c->lpn) { case 1: PL_D(1); break; case 2: PL_D(2); break; case 4: PL_D(4); break; case 8: PL_D(8); break;
                    default: PL_D(16); }
#undef PL_D
  PL_HIP(hipGetLastError());
  return PL_OK;
}
// multi-GPU: the diagonal of shared nodes is the sum over ranks; then invert again (main stream: RCCL)
int finish_diag_dist(pl_context *c) {
  if (!c->dist.active) return PL_OK;
  int rc = pl::dist_sum_shared(c->dist, c->diag.p, c->stream);
  if (rc) return fail(PL_ERR_HIP, "RCCL all-reduce of the Jacobi diagonal failed");
  pl::launch_invert_diag(c->N * 6, c->diag.p, c->have_bc ? c->fixed.p : nullptr, c->dinv.p, c->stream);
  return PL_OK;
}

// Dirichlet mask OR "shared with another rank" (all six dofs): the mask of the rank-local levels
__global__ void k_local_mask(int64_t N, const uint8_t *__restrict__ fixedbits, const uint8_t *__restrict__ shared,
                             uint8_t *__restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) mask[i] = (uint8_t)(fixedbits[i] | (shared[i] ? 0x3f : 0));
}

// fp32 copy of the Jacobi inverse for the multi-level PCG kernels (main stream, after the diagonal is final)
int launch_dinv32(pl_context *c) {
  if (!c->coarse.enabled) return PL_OK;
  hipLaunchKernelGGL(pl::k_to_float, dim3(grid_for(c->N * 6)), dim3(pl::kBlock), 0, c->stream, c->N * 6, c->dinv.p,
                     c->coarse.dinv32);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// The rank-local levels (tile blocks on several GPUs, local dense level) see shared nodes as constrained.
int launch_local_mask(pl_context *c) {
  if (!c->coarse.enabled || !(c->dist.active || c->coarseL.enabled)) return PL_OK;
  hipLaunchKernelGGL(k_local_mask, dim3(grid_for(c->N)), dim3(pl::kBlock), 0, c->stream, c->N, c->fixedbits.p,
                     c->sharedbits.p, c->maskL.p);
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// modes of the tile level in use (not with the rank-local level of precond = 4); decided the same way when the blocks are
// built and when they are applied.  With a communicator the strain modes, like the rigid ones, leave out the nodes shared
// with other ranks.
inline int tile_modes_now(const pl_context *c) {
  return (c->coarse.tile_level && c->coarse.tile_modes == 12 && !c->coarseL.enabled) ? 12 : 6;
}

int launch_tile_blocks(pl_context *c, hipStream_t st) {
  pl::Coarse &cs = c->coarse;
  if (!cs.enabled || !c->have_bc || !cs.tile_level) return PL_OK;
  if (cs.cm == 12) return PL_OK;        // built inside build_coarse_level, which needs them first
  const uint8_t *fb = c->dist.active ? c->maskL.p : c->fixedbits.p;
  const bool twelve = tile_modes_now(c) == 12;
  if (!twelve)
    hipLaunchKernelGGL(pl::k_tile_blocks, dim3((unsigned)cs.n_tiles), dim3(pl::kBlock), 0, st, c->tile.tile_start.p,
                       c->tile.home_ptr.p, c->tile.foreign_ptr.p, c->tile.foreign_idx.p,
                       reinterpret_cast<const int2 *>(c->conn.p), c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p,
                       fb, cs.Bt_inv, (double *)nullptr);
  if (twelve) {
    hipLaunchKernelGGL(pl::k_tile_blocks_strain<true>, dim3((unsigned)cs.n_tiles), dim3(pl::kBlock), 0, st,
                       c->tile.tile_start.p, c->tile.home_ptr.p, c->tile.foreign_ptr.p, c->tile.foreign_idx.p,
                       reinterpret_cast<const int2 *>(c->conn.p), c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p, fb,
                       cs.Bt_raw);
    hipLaunchKernelGGL(pl::k_tile_blocks_strain<false>, dim3((unsigned)cs.n_tiles), dim3(pl::kBlock), 0, st,
                       c->tile.tile_start.p, c->tile.home_ptr.p, c->tile.foreign_ptr.p, c->tile.foreign_idx.p,
                       reinterpret_cast<const int2 *>(c->conn.p), c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p, fb,
                       cs.Bt_raw);
    hipLaunchKernelGGL(pl::k_tile_invert12, dim3((unsigned)cs.n_tiles), dim3(pl::kInv12Block), 0, st, cs.n_tiles,
                       (const double *)cs.Bt_raw, cs.Bt_inv);
  }
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// A_c = Z^T P K P Z on the device, then its Cholesky factor and W = L^-1 (pl_dense.h).
int build_coarse_level(pl_context *c, pl::Coarse &cs, const uint8_t *mask, bool reduce,
                       const std::function<void(int)> &after_chol = nullptr) {
  cs.ready = false;
  if (!cs.enabled || !c->have_bc) {
    if (after_chol) after_chol(1);
    return PL_OK;
  }
  bool tile_invert_pending = false;
  const int n = cs.ncp;
  if (!cs.ac_clean) PL_HIP(hipMemsetAsync(cs.Ac, 0, (size_t)n * n * sizeof(double), c->stream));
  cs.ac_clean = false;
  PL_HIP(hipMemsetAsync(cs.info, 0, 2 * sizeof(int), c->stream));
  if (cs.cm == 12) {
    // 12 modes per aggregate: A_c from the 12 x 12 tile blocks (built HERE, ahead of the factorisation that needs them,
    // instead of beside it on the side stream) plus the cross-tile struts
    const dim3 gt((unsigned)cs.n_tiles), blk(pl::kBlock);
    const int2 *conn2 = reinterpret_cast<const int2 *>(c->conn.p);
    // (the rigid x rigid part on a second stream beside the strain rows and the cross-tile struts; the inversion of the
    // tile blocks, which only the solve needs, beside the factorisation)
    // Several GPUs: A_c is the sum over ranks of what each rank's struts give on ALL nodes (Dirichlet mask only), while the
    // tile LEVEL lives on this rank's own nodes (mask | shared): two sets of tile blocks, the second one for the level.
    const bool two_sets = c->dist.active;
    if (two_sets) {
      if (!cs.Bt_rawA && hipMalloc((void **)&cs.Bt_rawA, (size_t)cs.n_tiles * 144 * sizeof(double)) != hipSuccess)
        return fail(PL_ERR_HIP, "pl_assemble: out of device memory for the tile blocks");
      hipLaunchKernelGGL(pl::k_tile_blocks_strain<true>, gt, blk, 0, c->stream, c->tile.tile_start.p, c->tile.home_ptr.p,
                         c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p,
                         c->xyz.p, (const uint8_t *)c->maskL.p, cs.Bt_raw);
      hipLaunchKernelGGL(pl::k_tile_blocks_strain<false>, gt, blk, 0, c->stream, c->tile.tile_start.p, c->tile.home_ptr.p,
                         c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p,
                         c->xyz.p, (const uint8_t *)c->maskL.p, cs.Bt_raw);
    }
    double *rawA = two_sets ? cs.Bt_rawA : cs.Bt_raw;
    // (rigid x rigid part and the cross-tile struts on a second stream beside the strain rows)
    PL_HIP(hipEventRecord(c->ev_t0, c->stream));
    PL_HIP(hipStreamWaitEvent(c->side2, c->ev_t0, 0));
    hipLaunchKernelGGL(pl::k_tile_blocks_strain<true>, gt, blk, 0, c->side2, c->tile.tile_start.p, c->tile.home_ptr.p,
                       c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p,
                       mask, rawA);
    if (cs.n_cross > 0)
      hipLaunchKernelGGL(pl::k_coarse_cross12, dim3(grid_for(cs.n_cross)), blk, 0, c->side2, cs.n_cross,
                         cs.cross_idx.p, c->conn.p, c->rec.p, cs.agg_of_node.p, cs.cen.p, c->xyz.p, mask, n, cs.Ac);
    PL_HIP(hipEventRecord(c->ev_t1, c->side2));
    hipLaunchKernelGGL(pl::k_tile_blocks_strain<false>, gt, blk, 0, c->stream, c->tile.tile_start.p, c->tile.home_ptr.p,
                       c->tile.foreign_ptr.p, c->tile.foreign_idx.p, conn2, c->rec.p, cs.agg_of_tile.p, cs.cen.p, c->xyz.p,
                       mask, rawA);
    PL_HIP(hipStreamWaitEvent(c->stream, c->ev_t1, 0));
    hipLaunchKernelGGL(pl::k_agg_add_tiles, dim3(grid_for(cs.n_tiles * 144)), blk, 0, c->stream, cs.n_tiles,
                       cs.agg_of_tile.p, (const double *)rawA, n, cs.Ac);
    PL_HIP(hipEventRecord(c->ev_t0, c->stream));
    PL_HIP(hipStreamWaitEvent(c->side2, c->ev_t0, 0));
    hipLaunchKernelGGL(pl::k_tile_invert12, dim3((unsigned)cs.n_tiles), dim3(pl::kInv12Block), 0,
                       c->side2, cs.n_tiles, (const double *)cs.Bt_raw, cs.Bt_inv);
    PL_HIP(hipEventRecord(c->ev_t1, c->side2));
    tile_invert_pending = true;
  } else {
  if (cs.n_fix < 0) {   // Dirichlet set changed: list the in-aggregate struts that touch it, grouped by aggregate
    if (!cs.fix_count) PL_HIP(hipMalloc((void **)&cs.fix_count, sizeof(int)));
    int cnt = 0;
    DevBuf<int64_t> keys;
    for (int pass = 0; pass < 2; ++pass) {
      PL_HIP(hipMemsetAsync(cs.fix_count, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(pl::k_list_fixed_struts, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B,
                         c->conn.p, cs.agg_of_node.p, mask, pass ? keys.p : (int64_t *)nullptr, cs.fix_count);
      if (pass == 0) {
        PL_HIP(hipMemcpyAsync(&cnt, cs.fix_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PL_HIP(hipStreamSynchronize(c->stream));
        if (cnt == 0) break;
        PL_HIP(keys.alloc((size_t)cnt));
      }
    }
    std::vector<int32_t> list;
    if (cnt > 0) {
      std::vector<int64_t> hk((size_t)cnt);
      PL_HIP(hipMemcpyAsync(hk.data(), keys.p, hk.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
      PL_HIP(hipStreamSynchronize(c->stream));
      std::sort(hk.begin(), hk.end());
      list.reserve(hk.size() + hk.size() / 4 + pl::kWave);
      for (size_t q = 0; q < hk.size(); ++q) {
        if (q > 0 && (hk[q] >> 32) != (hk[q - 1] >> 32))
          while (list.size() % pl::kWave) list.push_back(-1);
        list.push_back((int32_t)(hk[q] & 0xffffffffLL));
      }
      while (list.size() % pl::kWave) list.push_back(-1);
      PL_HIP(cs.fix_list.upload(list));
    }
    cs.n_fix = (int64_t)list.size();
  }
  if (cs.n_fix > 0)
    hipLaunchKernelGGL(pl::k_coarse_assemble, dim3(grid_for(cs.n_fix)), dim3(pl::kBlock), 0, c->stream, cs.n_fix,
                       cs.fix_list.p, c->conn.p, c->rec.p, cs.agg_of_node.p, cs.cen.p, c->xyz.p, mask, n,
                       cs.Ac);
  if (cs.n_cross > 0)
    hipLaunchKernelGGL(pl::k_coarse_assemble_cross, dim3(grid_for(cs.n_cross)), dim3(pl::kBlock), 0, c->stream,
                       cs.n_cross, cs.cross_idx.p, c->conn.p, c->rec.p, cs.agg_of_node.p, cs.cen.p, c->xyz.p,
                       mask, n, cs.Ac);
  }
  if (reduce && c->dist.active) {   // every rank holds the contribution of ITS struts; all ranks then factor the same matrix
    const int nb = n / pl::kNB;
    if (cs.bw_blocks > 0 && cs.bw_blocks + 1 < nb) {   // only the block band of the lower triangle travels (L_f is free)
      const int64_t cnt = (int64_t)n * (cs.bw_blocks + 1) * pl::kNB;
      hipLaunchKernelGGL(pl::k_band_pack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, n, n,
                         cs.bw_blocks, cs.Ac, cs.Lf);
      if (pl::dist_sum_scalars(c->dist, cs.Lf, (int)cnt, c->stream))
        return fail(PL_ERR_HIP, "RCCL all-reduce of the coarse operator failed");
      hipLaunchKernelGGL(pl::k_band_unpack, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, n, n,
                         cs.bw_blocks, cs.Lf, cs.Ac);
    } else if (pl::dist_sum_scalars(c->dist, cs.Ac, n * n, c->stream)) {
      return fail(PL_ERR_HIP, "RCCL all-reduce of the coarse operator failed");
    }
  }
  hipLaunchKernelGGL(pl::k_coarse_regularize, dim3((n + 255) / 256), dim3(256), 0, c->stream, n, cs.Ac);
  // the inverse factor in row ranges behind the chain (second stream), so that only the last range runs after it
  pl::TrtriPhases ph;
  {
    const char *e = std::getenv("PL_TRTRI_ROWS");
    ph.rows = e ? std::atoi(e) : 3;     // (measured: 50^3 Octet 1.81 / 1.77 / 1.75 ms with 8 / 4 / 2 block rows per range;
                                         // configs[2], 96 blocks: 7.82 / 7.52 / 7.47 / 7.38 / 7.34 ms with 12 / 6 / 4 / 3 / 2)
    ph.stream = c->side2;     // (on a CU-masked stream beside the masked BSR fill: 1.75 -> 2.45 ms, configs[2] 7.7 -> 10.5)
    ph.ev_go = c->ev_p0;
    ph.ev_done = c->ev_p1;
  }
  pl::coarse_factor(cs, n, c->stream, after_chol, c->opt.chol_persistent ? cs.bar : (unsigned *)nullptr, ph);
  cs.ainv_ready = false;
  // explicit A_c^-1: for the short form of the iteration (pl_small.h), and for every global level small enough to be applied
  // in one GEMV launch (pl_coarse.h coarse_apply; 2 n^3 / 3 flops on the matrix pipe: ~60 us at 1 536 dofs)
  if (&cs == &c->coarse && (small_wanted(c) || (n <= pl::kOneGemvMaxDofs && c->opkind == 0))) {
    if (!cs.Ainv && hipMalloc((void **)&cs.Ainv, (size_t)n * n * sizeof(float)) != hipSuccess)
      return fail(PL_ERR_HIP, "pl_assemble: out of device memory for the explicit inverse of the dense level");
    const long tiles = (long)(n / 32) * (n / 32 + 1) / 2;
    if (cs.w16)
      hipLaunchKernelGGL(pl::k_dense_explicit_inverse<pl::bf16_t>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, c->stream, n,
                         reinterpret_cast<const pl::bf16_t *>(cs.W), n, cs.Ainv);
    else
      hipLaunchKernelGGL(pl::k_dense_explicit_inverse<float>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, c->stream, n,
                         (const float *)cs.W, n, cs.Ainv);
    cs.ainv_ready = true;
  }
  if (tile_invert_pending) PL_HIP(hipStreamWaitEvent(c->stream, c->ev_t1, 0));
  PL_HIP(hipGetLastError());
  int info[2] = {0, 0};
  PL_HIP(hipMemcpyAsync(info, cs.info, sizeof(info), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  if (&cs == &c->coarse) c->coarse_info = info[0];
  cs.ready = (info[0] == 0);   // not SPD -> fall back to Jacobi
  return PL_OK;
}

int build_coarse(pl_context *c, const std::function<void(int)> &after_chol = nullptr) {
  int rc = build_coarse_level(c, c->coarse, c->fixedbits.p, true, after_chol);
  if (rc || !c->coarseL.enabled) return rc;
  // (mask = Dirichlet | shared, launch_local_mask: its modes live on this rank's own nodes only)
  return build_coarse_level(c, c->coarseL, c->maskL.p, false);
}

inline const double *cinv(const pl_context *c) { return c->cls_ready ? c->cls_table.p : c->kcc_inv.p; }
inline const uint16_t *ccls(const pl_context *c) { return c->cls_ready ? c->cls_id.p : (const uint16_t *)nullptr; }

// The nodes eliminated in this solve: the candidates of pl_create (an independent set of the node graph, so K_cc is block
// diagonal) that carry no Dirichlet dof and are not shared with another rank.
int select_condensed(pl_context *c, const std::vector<uint8_t> &bits) {
  c->cond_ready = false;
  c->n_cond = 0;
  c->cond_agree = -1;
  // (several GPUs: nodes shared with another rank stay unknowns - an eliminated node then has all its struts on this
  // rank, its 6 x 6 block and the first pass of the condensed operator are complete locally, pl_ops.h)
  const bool wanted = c->opt.condense >= 0 && c->opkind == 0 && c->coarse.enabled && c->opt.precision != 2 &&
                      !c->h_cand.empty();
  if (!wanted) return PL_OK;
  const int64_t N = c->N;
  std::vector<int32_t> picked;
  for (int64_t i = 0; i < N; ++i)
    if (c->h_cand[i] && !bits[i] && (c->h_shared.empty() || !c->h_shared[i])) picked.push_back((int32_t)i);
  if (picked.empty()) return PL_OK;
  std::vector<uint8_t> flag((size_t)N, 0), mask(bits);
  for (int32_t i : picked) {
    flag[i] = 1;
    mask[i] = 0x3f;
  }
  PL_HIP(c->cnodes.alloc(picked.size()));
  PL_HIP(c->kcc_inv.alloc(picked.size() * 36));
  if (!c->maskC.p) {
    PL_HIP(c->maskC.alloc(N));
    PL_HIP(c->cflag.alloc(N));
  }
  PL_HIP(hipMemcpy(c->cnodes.p, picked.data(), picked.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  {
    std::vector<int32_t> keep;
    keep.reserve((size_t)N - picked.size());
    for (int64_t i = 0; i < N; ++i)
      if (!flag[i]) keep.push_back((int32_t)i);
    PL_HIP(c->ckeep.alloc(std::max<size_t>(1, keep.size())));
    if (!keep.empty()) PL_HIP(hipMemcpy(c->ckeep.p, keep.data(), keep.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (!c->cbase.p) PL_HIP(c->cbase.alloc(N));
  c->cbase_state = -1;
  PL_HIP(hipMemcpy(c->maskC.p, mask.data(), N, hipMemcpyHostToDevice));
  PL_HIP(hipMemcpy(c->cflag.p, flag.data(), N, hipMemcpyHostToDevice));
  // per strut: which ends are condensed (read by every pass of the condensed operator)
  if (!c->cend.p) PL_HIP(c->cend.alloc(c->B));
  hipLaunchKernelGGL(k_cond_ends, dim3(grid_for(c->B)), dim3(pl::kBlock), 0, c->stream, c->B,
                     reinterpret_cast<const int2 *>(c->conn.p), (const uint8_t *)c->cflag.p, c->cend.p);
  refresh_visit_words(c, c->stream);   // (the same bits ride in the visit words of the LDS-resident K*p)
  if (refresh_visit_words_dir(c, c->stream)) return PL_ERR_HIP;
  PL_HIP(hipGetLastError());
  c->n_cond = (int64_t)picked.size();
  return PL_OK;
}
// Several GPUs: a solve with node elimination has one exchange more (its prologue) than a solve without, so either every
// rank eliminates or none does.  One 1-double all-reduce after a pl_set_bc (main stream; collective - every rank of a
// handle with these options calls it at the same point of pl_assemble / pl_set_bc).
int agree_condensed(pl_context *c) {
  if (!c->dist.active || c->cond_agree != -1) return PL_OK;
  const bool asked = c->opt.condense >= 0 && c->opkind == 0 && c->coarse.enabled && c->opt.precision != 2;
  if (!asked) {   // (options are the same on every rank: nobody has candidates, nobody calls the collective)
    c->cond_agree = 0;
    return PL_OK;
  }
  double *slot = c->scal.p + pl::S_AUX * pl::kSlots;
  double v = c->n_cond > 0 ? 0.0 : 1.0;
  PL_HIP(hipMemcpyAsync(slot, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (pl::dist_sum_scalars(c->dist, slot, 1, c->stream)) return fail(PL_ERR_HIP, "all-reduce of the node-elimination vote failed");
  PL_HIP(hipMemcpyAsync(&v, slot, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  PL_HIP(hipStreamSynchronize(c->stream));
  c->cond_agree = (v == 0.0) ? 1 : 0;
  return PL_OK;
}
// K_cc^-1 of the condensed nodes from the current records (after launch_records, same stream)
int launch_condensed_blocks(pl_context *c, hipStream_t st) {
  c->cond_ready = false;
  if (c->n_cond <= 0 || !c->have_bc) return PL_OK;
  if (c->dist.active && c->cond_agree != 1) return PL_OK;
  hipLaunchKernelGGL(pl::k_node_block_inverse, dim3(grid_for(c->n_cond)), dim3(pl::kBlock), 0, st, c->n_cond,
                     c->cnodes.p, pl::kWave / c->lpn, c->slice_ptr.p, c->ent.p, c->rec.p, c->kcc_inv.p);
  PL_HIP(hipGetLastError());
  c->cond_ready = true;
  c->cls_ready = false;
  c->cbase_state = -1;       // class ids are assigned anew below
  if (c->opt.palette && c->pal_id.p) {
    // Classes by the record-palette ids of the incident struts (queued behind launch_palette on the same stream; if the
    // record palette turns out not to hold, finish_condensed_classes drops the classes as well)
    if (!c->cls_table.p) {
      PL_HIP(c->cls_keys.alloc(65536));
      PL_HIP(c->cls_owner.alloc(65536));
      PL_HIP(c->cls_flags.alloc(1));
      PL_HIP(c->cls_table.alloc((size_t)65536 * 36));
      void *pinned = nullptr;
      PL_HIP(hipHostMalloc(&pinned, sizeof(int), hipHostMallocDefault));
      c->cls_host_flag = static_cast<int *>(pinned);
    }
    if (c->cls_key.n < (size_t)c->n_cond) {
      PL_HIP(c->cls_key.alloc((size_t)c->n_cond));
      PL_HIP(c->cls_id.alloc((size_t)c->n_cond));
    }
    *c->cls_host_flag = 1;
    PL_HIP(hipMemsetAsync(c->cls_keys.p, 0xFF, 65536 * sizeof(unsigned long long), st));
    PL_HIP(hipMemsetAsync(c->cls_owner.p, 0x7F, 65536 * sizeof(int), st));
    PL_HIP(hipMemsetAsync(c->cls_flags.p, 0, sizeof(int), st));
    const dim3 g(grid_for(c->n_cond)), blk(pl::kBlock);
    hipLaunchKernelGGL(pl::k_cls_hash, g, blk, 0, st, c->n_cond, c->cnodes.p, pl::kWave / c->lpn, c->slice_ptr.p,
                       c->ent.p, c->pal_id.p, c->cls_key.p);
    hipLaunchKernelGGL(pl::k_cls_insert, g, blk, 0, st, c->n_cond, c->cls_key.p, c->cls_keys.p, c->cls_owner.p,
                       c->cls_id.p, c->cls_flags.p);
    hipLaunchKernelGGL(pl::k_cls_publish, dim3(grid_for(c->n_cond * 36)), blk, 0, st, c->n_cond, c->kcc_inv.p,
                       c->cls_owner.p, c->cls_id.p, c->cls_table.p);
    hipLaunchKernelGGL(pl::k_cls_verify, g, blk, 0, st, c->n_cond, c->kcc_inv.p, c->cls_id.p, c->cls_table.p,
                       c->cls_flags.p);
    PL_HIP(hipGetLastError());
    PL_HIP(hipMemcpyAsync(c->cls_host_flag, c->cls_flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  return PL_OK;
}
// after the stream has drained and finish_palette() has run
void finish_condensed_classes(pl_context *c) {
  c->cls_ready = c->cond_ready && c->cls_host_flag && *c->cls_host_flag == 0 && c->pal_ready && c->opt.palette;
  if (c->cls_host_flag) *c->cls_host_flag = 1;
}

// slices [slice0, slice1) of the sliced-ELL incidence (slice1 < 0: to the end)
int launch_bsr_fill(pl_context *c, int with_bc, hipStream_t st, int64_t slice0 = 0, int64_t slice1 = -1) {
  const uint8_t *fb = c->have_bc ? c->fixedbits.p : (const uint8_t *)nullptr;
  const size_t lds = (size_t)(pl::kBsrBlock / pl::kWave) * 64 * pl::kBsrPitch * sizeof(double);   // 38 KB
  if (slice1 < 0 || slice1 > c->n_slices) slice1 = c->n_slices;
  if (slice0 >= slice1) return PL_OK;
  const unsigned gb = grid_for(slice1 - slice0, pl::kBsrBlock / pl::kWave);
#define PL_B(L)                                                                                                  \
  hipLaunchKernelGGL((pl::k_bsr_fill<L>), dim3(gb), dim3(pl::kBsrBlock), lds, st, c->N, c->slice_ptr.p, c->ent.p,  \
                     c->rec.p, c->bsr_rowptr.p, c->ent_slot.p, c->diag_slot.p, fb, with_bc, c->bsr_vals.p, slice0, slice1)
  switch (c->lpn) { case 1: PL_B(1); break; case 2: PL_B(2); break; case 4: PL_B(4); break; case 8: PL_B(8); break;
                    default: PL_B(16); }
#undef PL_B
  PL_HIP(hipGetLastError());
  return PL_OK;
}

// ----------------------------------------------------------------------------------------------------------
// incidence (sliced ELL) + BSR pattern, on the host
// ----------------------------------------------------------------------------------------------------------
int build_incidence(pl_context *c, const std::vector<int32_t> &conn) {
  const int64_t N = c->N, B = c->B;
  std::vector<int32_t> deg(N, 0);
  pl::parallel_for(2 * B, [&](int64_t k0, int64_t k1, unsigned) {
    for (int64_t k = k0; k < k1; ++k) __atomic_fetch_add(&deg[conn[k]], 1, __ATOMIC_RELAXED);
  }, 1 << 16);
  std::vector<int64_t> ptr(N + 1, 0);
  for (int64_t i = 0; i < N; ++i) ptr[i + 1] = ptr[i] + deg[i];
  struct E {
    int32_t other, code;
  };
  std::vector<E> adj((size_t)ptr[N]);
  std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
  pl::parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {       // any order inside a row: the rows are sorted next
    for (int64_t b = b0; b < b1; ++b) {
      const int32_t a = conn[2 * b], d = conn[2 * b + 1];
      adj[__atomic_fetch_add(&fill[d], (int64_t)1, __ATOMIC_RELAXED)] = {a, (int32_t)b};   // d is the strut's tip (point2)
      adj[__atomic_fetch_add(&fill[a], (int64_t)1, __ATOMIC_RELAXED)] =                    // a is point1 -> reversed record
          {d, (int32_t)((uint32_t)b | 0x80000000u)};
    }
  }, 1 << 16);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i)
      std::sort(adj.begin() + ptr[i], adj.begin() + ptr[i + 1],
                [](const E &l, const E &r) { return l.other < r.other || (l.other == r.other && l.code < r.code); });
  });

  // sliced ELL: kSliceNodes (16) nodes per slice, width padded to a multiple of kLPN (4) so that one slice is a
  // whole number of 64-entry wave trips
  const int SN = pl::kWave / c->lpn;
  const int64_t S = (N + SN - 1) / SN;
  std::vector<int64_t> sp(S + 1, 0);
  for (int64_t s = 0; s < S; ++s) {
    int w = 0;
    for (int64_t i = s * SN; i < std::min<int64_t>(N, s * SN + SN); ++i) w = std::max(w, deg[i]);
    w = (w + c->lpn - 1) / c->lpn * c->lpn;
    sp[s + 1] = sp[s] + (int64_t)w * SN;
  }
  std::vector<int2> ent((size_t)sp[S], int2{-1, 0});
  // BSR pattern: per row the diagonal block + one block per entry, columns ascending.  Parallel struts between the
  // same pair of nodes (possible in hybrid cells) get separate blocks with equal column index.
  c->h_rowptr.assign(N + 1, 0);
  for (int64_t i = 0; i < N; ++i) c->h_rowptr[i + 1] = c->h_rowptr[i] + deg[i] + 1;
  c->nblk = c->h_rowptr[N];
  c->h_col.assign((size_t)c->nblk, 0);
  std::vector<int32_t> ent_slot((size_t)sp[S], 0), diag_slot(N, 0);
  pl::parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
  for (int64_t i = i0; i < i1; ++i) {
    const int64_t s = i / SN, lane = i % SN;
    int slot = 0;
    bool diag_done = false;
    for (int j = 0; j < deg[i]; ++j) {
      const E &e = adj[ptr[i] + j];
      if (!diag_done && e.other > i) {
        diag_slot[i] = slot;
        c->h_col[c->h_rowptr[i] + slot++] = (int32_t)i;
        diag_done = true;
      }
      const int64_t pos = sp[s] + (int64_t)j * SN + lane;
      ent[pos] = int2{e.other, e.code};
      ent_slot[pos] = slot;
      c->h_col[c->h_rowptr[i] + slot++] = e.other;
    }
    if (!diag_done) {
      diag_slot[i] = slot;
      c->h_col[c->h_rowptr[i] + slot++] = (int32_t)i;
    }
  }
  });
  c->n_slices = S;
  c->n_ent = sp[S];
  PL_HIP(c->slice_ptr.alloc(S + 1));
  PL_HIP(c->ent.alloc(std::max<size_t>(1, ent.size())));
  PL_HIP(c->ent_slot.alloc(std::max<size_t>(1, ent_slot.size())));
  PL_HIP(c->diag_slot.alloc(N));
  PL_HIP(hipMemcpy(c->slice_ptr.p, sp.data(), (S + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  if (!ent.empty()) {
    PL_HIP(hipMemcpy(c->ent.p, ent.data(), ent.size() * sizeof(int2), hipMemcpyHostToDevice));
    PL_HIP(hipMemcpy(c->ent_slot.p, ent_slot.data(), ent_slot.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  PL_HIP(hipMemcpy(c->diag_slot.p, diag_slot.data(), N * sizeof(int32_t), hipMemcpyHostToDevice));
  return PL_OK;
}

bool valid(pl_handle h) { return h != nullptr; }

// PL_TIMING=1 in the environment: wall clock of the host-side stages of pl_create on stderr
struct StageTimer {
  const char *what;
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit StageTimer(const char *w) : what(w), on(std::getenv("PL_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char *stage) {
    if (!on) return;
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[%s] %-28s %8.1f ms\n", what, stage, std::chrono::duration<double, std::milli>(t1 - t0).count());
    t0 = t1;
  }
};

}  // namespace
