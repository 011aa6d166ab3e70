// Single-reduction form of the multi-level PCG (pl_opts_t.cg_form = 1; Chronopoulos & Gear 1989 with the coarse residual
// carried by recurrence).  The ordinary form needs three exchanges per iteration on several GPUs - interface rows of
// K*p, p.Ap, then [Z^T r | r.r | r.z] - because alpha must be known before r, and r before M^-1 r.  Here
//     u = M^-1 r,  w = K u,  gamma = r.u,  delta = u.w
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)
//     p = u + beta p,  s = w + beta s (= K p),  x += alpha p,  r -= alpha s
// and the dense level's right-hand side follows r without being restricted again:
//     Z^T s = Z^T w + beta Z^T s,   Z^T r -= alpha Z^T s
// so that u = M^-1 r needs nothing from the other ranks, and ONE all-reduce per iteration carries
//     [ Z^T w : ncp | u.w : kSlots | r.(D^-1 + tile level) r : kSlots | r.r : kSlots ]
// (all four are sums over ranks of local partial sums).  The price is three more stored vectors (u, w, s): about 13.5
// vector passes per iteration instead of 9.  ||r_k|| is known one iteration late (it rides in the
// same all-reduce), so a solve reports one iteration more than the ordinary form.
//
// Per iteration k, all on one stream:
//   k_cg1_update    scalars from the reduced block; p, s, x, r; tile-level restriction of the new r, its share of r.u,
//                   r.r; workgroup 0: Z^T s, Z^T r, ||r_k||^2 -> history, (gamma, alpha) for the next iteration
//   k_tri_gemv x2   y_c = A_c^-1 Z^T r, r_c.y_c (replicated on every rank, kept out of the all-reduce)
//   k_cg1_precond   u = D^-1 r + P Z (y_c + y_t); clears the block the next-but-one iteration accumulates into
//   k_spmv_tile     w = P K u, u.w          (+ interface exchange on several GPUs)
//   k_cg1_restrict  Z^T w
//   all-reduce      (several GPUs only)
#pragma once
#include "pl_coarse.h"

namespace pl {

// offsets inside one reduction block
__host__ __device__ inline int cg1_block_size(int ncp) { return ncp + 3 * kSlots; }

// Z^T (weight o v) of every aggregate, one workgroup per tile, atomics into out[CM * agg + k] (~8 tiles per aggregate).
// CM = modes per aggregate of the dense level: 6 rigid, or 12 = rigid + uniform strains (pl_coarse.h: strain_disp).
template <int CM>
__global__ __launch_bounds__(kBlock) void k_cg1_restrict(const int32_t *__restrict__ tile_start,
                                                         const int32_t *__restrict__ agg_of_tile,
                                                         const double *__restrict__ cen, const double *__restrict__ xyz,
                                                         const double *__restrict__ v,
                                                         const double *__restrict__ wt /* may be null */,
                                                         double *__restrict__ out) {
  __shared__ double red[CM][kBlock / kWave];
  const int t = blockIdx.x;
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double acc[CM];
#pragma unroll
  for (int k = 0; k < CM; ++k) acc[k] = 0.0;
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    double vv[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double2 q = load_pair(v, 3 * (int64_t)i + k);
      vv[2 * k] = q.x;
      vv[2 * k + 1] = q.y;
    }
    if (wt) {
#pragma unroll
      for (int k = 0; k < 6; ++k) vv[k] *= wt[6 * (int64_t)i + k];
    }
    const double rx = xyz[3 * (int64_t)i] - c0, ry = xyz[3 * (int64_t)i + 1] - c1, rz = xyz[3 * (int64_t)i + 2] - c2;
    acc[0] += vv[0];
    acc[1] += vv[1];
    acc[2] += vv[2];
    acc[3] += vv[3] + (ry * vv[2] - rz * vv[1]);
    acc[4] += vv[4] + (rz * vv[0] - rx * vv[2]);
    acc[5] += vv[5] + (rx * vv[1] - ry * vv[0]);
    if constexpr (CM == 12) {
      acc[6] += rx * vv[0];
      acc[7] += ry * vv[1];
      acc[8] += rz * vv[2];
      acc[9] += 0.5 * (ry * vv[0] + rx * vv[1]);
      acc[10] += 0.5 * (rz * vv[1] + ry * vv[2]);
      acc[11] += 0.5 * (rz * vv[0] + rx * vv[2]);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < CM; ++k) {
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[k][wv] = s;
  }
  __syncthreads();
  if (threadIdx.x < CM) {
    double s = 0.0;
    for (int q = 0; q < nw; ++q) s += red[threadIdx.x][q];
    unsafeAtomicAdd(out + CM * a + threadIdx.x, s);
  }
}

// INIT: the pass before iteration 0 - no vector update (alpha = beta = 0), only the tile level and the partial sums of
// r0 (Z^T r0 itself comes from k_cg1_restrict).
// TM = modes of the tile level (6, or 12 = rigid + uniform strains about the aggregate's reference point, B_t^-1 12 x 12).
template <bool INIT, int TM>
__global__ __launch_bounds__(kBlock) void k_cg1_update(const int32_t *__restrict__ tile_start,
                                                       const int32_t *__restrict__ agg_of_tile,
                                                       const double *__restrict__ cen, const double *__restrict__ xyz,
                                                       const double *__restrict__ u, const double *__restrict__ w,
                                                       const float *__restrict__ dinv32,
                                                       const double *__restrict__ wt /* may be null */,
                                                       double *__restrict__ p, double *__restrict__ s,
                                                       double *__restrict__ x, double *__restrict__ r,
                                                       const double *__restrict__ blk /* reduced block of this iteration */,
                                                       const double *__restrict__ gc /* r_c.y_c slots */,
                                                       const double *__restrict__ st_cur, double *__restrict__ st_nxt,
                                                       double *__restrict__ blk_nxt,
                                                       const double *__restrict__ Bt_inv /* may be null */,
                                                       double *__restrict__ yt,
                                                       const uint8_t *__restrict__ shared /* may be null */,
                                                       double *__restrict__ rc, double *__restrict__ sc, int ncp,
                                                       double *__restrict__ hist, int k) {
  __shared__ double red[14][kBlock / kWave];
  const int t = blockIdx.x;
  double alpha = 0.0, beta = 0.0;
  if (!INIT) {
    const double delta = scalar_read(blk + ncp, 0);
    const double gamma = scalar_read(blk + ncp, 1) + scalar_read(gc, 0);
    const double g_old = st_cur[0], a_old = st_cur[1];
    double den = delta;
    if (k > 0 && g_old != 0.0 && a_old != 0.0) {
      beta = gamma / g_old;
      den = delta - beta * gamma / a_old;
    }
    alpha = (den != 0.0) ? gamma / den : 0.0;
    if (t == 0) {   // dense level by recurrence (replicated on every rank), history, state of the next iteration
      for (int e = threadIdx.x; e < ncp; e += blockDim.x) {
        const double sv = blk[e] + beta * sc[e];
        sc[e] = sv;
        rc[e] -= alpha * sv;
      }
      if (threadIdx.x < kWave) {
        const double rr = scalar_read(blk + ncp, 2);
        if (threadIdx.x == 0) {
          hist[k] = rr;
          st_nxt[0] = gamma;
          st_nxt[1] = alpha;
        }
      }
    }
  }
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double bi[TM];                        // row threadIdx.x of B_t^-1, fetched early: it is needed at the very end
#pragma unroll
  for (int j = 0; j < TM; ++j) bi[j] = 0.0;
  if (Bt_inv && threadIdx.x < TM) {
#pragma unroll
    for (int j = 0; j < TM; ++j) bi[j] = Bt_inv[(size_t)t * (TM * TM) + TM * threadIdx.x + j];
  }
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // 0-5 tile restriction (nodes of this rank alone), 6 r.r, 7 r.D^-1 r
  double accS[6] = {0, 0, 0, 0, 0, 0};        // TM = 12: the tile's strain restrictions (same nodes)
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    double rv[6], dv[6];
    const float2 *d2 = reinterpret_cast<const float2 *>(dinv32 + 6 * (int64_t)i);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int64_t j = 3 * (int64_t)i + q;
      double2 rr = load_pair(r, j);
      if (!INIT) {
        const double2 uu = load_pair(u, j), ww = load_pair(w, j);
        double2 pp = load_pair(p, j), ss = load_pair(s, j), xx = load_pair(x, j);
        pp.x = uu.x + beta * pp.x;
        pp.y = uu.y + beta * pp.y;
        ss.x = ww.x + beta * ss.x;
        ss.y = ww.y + beta * ss.y;
        xx.x += alpha * pp.x;
        xx.y += alpha * pp.y;
        rr.x -= alpha * ss.x;
        rr.y -= alpha * ss.y;
        store_pair(p, j, pp);
        store_pair(s, j, ss);
        store_pair(x, j, xx);
        store_pair(r, j, rr);
      }
      const float2 dd = d2[q];
      rv[2 * q] = rr.x;
      rv[2 * q + 1] = rr.y;
      dv[2 * q] = dd.x;
      dv[2 * q + 1] = dd.y;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double wq = wt ? wt[6 * (int64_t)i + q] : 1.0;
      acc[6] += wq * rv[q] * rv[q];
      acc[7] += wq * dv[q] * rv[q] * rv[q];
    }
    if (!(shared && shared[i])) {
      const double rx = xyz[3 * (int64_t)i] - c0, ry = xyz[3 * (int64_t)i + 1] - c1, rz = xyz[3 * (int64_t)i + 2] - c2;
      acc[0] += rv[0];
      acc[1] += rv[1];
      acc[2] += rv[2];
      acc[3] += rv[3] + (ry * rv[2] - rz * rv[1]);
      acc[4] += rv[4] + (rz * rv[0] - rx * rv[2]);
      acc[5] += rv[5] + (rx * rv[1] - ry * rv[0]);
      if constexpr (TM == 12) {
        accS[0] += rx * rv[0];
        accS[1] += ry * rv[1];
        accS[2] += rz * rv[2];
        accS[3] += 0.5 * (ry * rv[0] + rx * rv[1]);
        accS[4] += 0.5 * (rz * rv[1] + ry * rv[2]);
        accS[5] += 0.5 * (rz * rv[0] + rx * rv[2]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double v = wave_sum(acc[q]);
    if (lane == 0) red[q][wv] = v;
  }
  if constexpr (TM == 12) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const double v = wave_sum(accS[q]);
      if (lane == 0) red[8 + q][wv] = v;
    }
  }
  __syncthreads();
  // first wave: lanes 0..5 rigid restriction, 6 r.r, 7 r.D^-1 r, 8..13 strain restriction; then the tile level
  // y_t = B_t^-1 (Z_t^T r), whose share r_t . y_t of r.u joins r.D^-1 r.  The 16 lanes belong to one wave: the sums reach every
  // lane by v_readlane (compile-time lane numbers), the final sum by a DPP row reduction - no LDS exchange without a barrier
  // (round-4 advisor finding), and no fence that would wait for the atomics issued on the way.
  if (threadIdx.x < 16) {
    double *gam_slot = blk_nxt + ncp + kSlots + (blockIdx.x & (kSlots - 1)), *rr_slot = gam_slot + kSlots;
    double v = 0.0;
    if (threadIdx.x < (TM == 12 ? 14 : 8))
      for (int q = 0; q < nw; ++q) v += red[threadIdx.x][q];
    if (threadIdx.x == 6) unsafeAtomicAdd(rr_slot, v);
    if (!Bt_inv) {
      if (threadIdx.x == 7) unsafeAtomicAdd(gam_slot, v);
    } else {
      // mode m of the tile level sits in lane m (m < 6) or lane m + 2 (strains)
      double y = 0.0, stv = 0.0;
#pragma unroll
      for (int j = 0; j < TM; ++j) y += bi[j] * lane_value(v, j < 6 ? j : j + 2);
      // (fetched by all 16 lanes: a permute must not read a lane that sits out of a branch)
      stv = lane_value_dyn(v, threadIdx.x < 6 ? (int)threadIdx.x : min((int)threadIdx.x + 2, 15));
      if (threadIdx.x < TM) yt[TM * (size_t)t + threadIdx.x] = y;
      const double g = row_sums((threadIdx.x < TM ? y * stv : 0.0) + (threadIdx.x == 7 ? v : 0.0));      // lane 15: the total
      if (threadIdx.x == 15) unsafeAtomicAdd(gam_slot, g);
    }
  }
}

// u = D^-1 r + P Z (y_c + y_t), stored; `clear` [n_clear] and `clear2` [kSlots] are zeroed for their next use.
// TM / cm: modes of the tile level / per aggregate of the dense level (12 = rigid + uniform strains; cm = 12 needs TM = 12).
template <int TM>
__global__ __launch_bounds__(kBlock) void k_cg1_precond(const int32_t *__restrict__ tile_start,
                                                        const double *__restrict__ r, const float *__restrict__ dinv32,
                                                        const double *__restrict__ xyz,
                                                        const int32_t *__restrict__ agg_of_tile,
                                                        const double *__restrict__ cen, const double *__restrict__ yc,
                                                        const double *__restrict__ yt /* may be null */,
                                                        const uint8_t *__restrict__ fixedbits,
                                                        const uint8_t *__restrict__ shared /* may be null */,
                                                        double *__restrict__ u, double *__restrict__ clear, int n_clear,
                                                        double *__restrict__ clear2, int cm) {
  if (clear && (blockIdx.x == 1 || gridDim.x == 1))
    for (int e = threadIdx.x; e < n_clear; e += blockDim.x) clear[e] = 0.0;
  if (clear2 && blockIdx.x == 0)
    for (int e = threadIdx.x; e < kSlots; e += blockDim.x) clear2[e] = 0.0;
  const int t = blockIdx.x;
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double *y = yc + cm * a;
  double U0 = y[0], U1 = y[1], U2 = y[2], W0 = y[3], W1 = y[4], W2 = y[5];
  double T[TM];
#pragma unroll
  for (int k = 0; k < TM; ++k) T[k] = 0.0;
  if (yt) {
    const double *q = yt + TM * (size_t)t;
#pragma unroll
    for (int k = 0; k < TM; ++k) T[k] = q[k];
    if (!shared) {   // one GPU: tile and aggregate use the same reference point, the two rigid motions just add
      U0 += T[0]; U1 += T[1]; U2 += T[2]; W0 += T[3]; W1 += T[4]; W2 += T[5];
    }
  }
  double ED[6] = {0, 0, 0, 0, 0, 0};   // the aggregate's uniform strains (12-mode dense level)
  if constexpr (TM == 12) {
    if (cm == 12) {
#pragma unroll
      for (int k = 0; k < 6; ++k) ED[k] = y[6 + k];
      if (!shared) {   // one GPU: same reference point, same nodes - they add to the tile's
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          T[6 + k] += ED[k];
          ED[k] = 0.0;
        }
      }
    }
  }
  const bool own_t = yt && shared;   // several GPUs: nodes shared with other ranks are left out of the tile level
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  for (int64_t i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    const double rx = xyz[3 * i] - c0, ry = xyz[3 * i + 1] - c1, rz = xyz[3 * i + 2] - c2;
    double zc[6] = {U0 + (W1 * rz - W2 * ry), U1 + (W2 * rx - W0 * rz), U2 + (W0 * ry - W1 * rx), W0, W1, W2};
    if constexpr (TM == 12) {   // uniform strains: u += eps r (several GPUs: the aggregate's everywhere, the tile's on own nodes)
      if (shared && cm == 12) {
        zc[0] += ED[0] * rx + 0.5 * (ED[3] * ry + ED[5] * rz);
        zc[1] += ED[1] * ry + 0.5 * (ED[3] * rx + ED[4] * rz);
        zc[2] += ED[2] * rz + 0.5 * (ED[4] * ry + ED[5] * rx);
      }
      if (!(shared && shared[i])) {
        zc[0] += T[6] * rx + 0.5 * (T[9] * ry + T[11] * rz);
        zc[1] += T[7] * ry + 0.5 * (T[9] * rx + T[10] * rz);
        zc[2] += T[8] * rz + 0.5 * (T[10] * ry + T[11] * rx);
      }
    }
    if (own_t && !shared[i]) {
      zc[0] += T[0] + (T[4] * rz - T[5] * ry);
      zc[1] += T[1] + (T[5] * rx - T[3] * rz);
      zc[2] += T[2] + (T[3] * ry - T[4] * rx);
      zc[3] += T[3];
      zc[4] += T[4];
      zc[5] += T[5];
    }
    const unsigned fb = fixedbits[i];
    const float2 *d2 = reinterpret_cast<const float2 *>(dinv32 + 6 * i);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double2 rr = load_pair(r, 3 * i + q);
      const float2 dd = d2[q];
      double2 uu;
      uu.x = dd.x * rr.x + (((fb >> (2 * q)) & 1u) ? 0.0 : zc[2 * q]);
      uu.y = dd.y * rr.y + (((fb >> (2 * q + 1)) & 1u) ? 0.0 : zc[2 * q + 1]);
      store_pair(u, 3 * i + q, uu);
    }
  }
}

}  // namespace pl
