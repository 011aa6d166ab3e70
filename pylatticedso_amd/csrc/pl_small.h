// Short form of the multi-level PCG iteration for small lattices (pl_opts_t.short_iteration).
//
// On a lattice of a few hundred K*p tiles (the design loops of LatticeOpti: BASELINE configs[3] has 128 tiles) a launch lasts
// 3 - 6 us whatever it computes, and the ordinary iteration is a chain of five dependent launches - six under node
// elimination: K*p (two passes), update + restriction + tile solve, two triangular GEMVs of the dense level, direction.
// The dependencies of a PCG iteration with a coarse level need only three (four) of those boundaries:
//     p.Kp  ->  alpha  ->  r, Z^T r  ->  z = M^-1 r, r.z  ->  beta  ->  p
// Short form:
//   * the dense level's solve is y_c = A_c^-1 r_c with the EXPLICIT inverse (k_dense_explicit_inverse: W^T W on the fp64
//     matrix pipe, once per assembly); a tile needs only the cm rows of its own aggregate, so the z-kernel computes them
//     itself (cm x ncp multiply-adds per workgroup) - no GEMV launches;
//   * k_small_z writes z = D^-1 r + P Z (y_c + y_t) and the slots of r.z in ONE launch; beta is known only after it, so
//   * the next K*p launch forms p = z + beta p_old while it stages its rows (pl_tile.h, DEFER) and makes the iterate's update
//     x += alpha p_old on the way.
// Per iteration: K*p (1 or 2 launches), k_pcg_update_tile (unchanged), k_small_z.  Same preconditioner, same recurrences:
// the iterates equal the ordinary form's up to the rounding of A_c^-1 (stored in fp32 like W).
// Scalars: a ring of four slotted sets R_k (S_RZ_OLD = r_k.z_k, S_PAP = p_k.K p_k), the restriction r_c double-buffered by
// iteration parity - nothing is zeroed by a launch that may still be read by its predecessor.
// Applies when n_tiles x cm x ncp is small (every tile reads its rows of A_c^-1 once per iteration): see small_applies().
#pragma once
#include "pl_coarse.h"

namespace pl {

typedef double v4f64_small __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double w_entry(const float *W, size_t i) { return (double)W[i]; }
__device__ __forceinline__ double w_entry(const bf16_t *W, size_t i) {
  return (double)__uint_as_float((unsigned)reinterpret_cast<const uint16_t *>(W)[i] << 16);
}

// Ainv = W^T W (W = L^-1 lower triangular, row-major n x ld in fp32 or bfloat16): Ainv[a][b] = sum_{k >= max(a, b)} W[k][a] W[k][b],
// accumulated in fp64 on v_mfma_f64_16x16x4_f64, stored in fp32, exactly symmetric (lower tiles computed, mirrored).
// One wave per 32 x 32 tile of the lower triangle; operand lane (i = lane & 15, k = lane >> 4) reads W[k0 + k][a0 + i]: four
// rows of 16 consecutive entries per instruction.  n is a multiple of 64.
template <typename WT>
__global__ __launch_bounds__(256) void k_dense_explicit_inverse(int n, const WT *__restrict__ W, int ld,
                                                                float *__restrict__ Ainv) {
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int nt = n / 32;
  const long q = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= (long)nt * (nt + 1) / 2) return;
  int ti = (int)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
  while ((long)(ti + 1) * (ti + 2) / 2 <= q) ++ti;
  while ((long)ti * (ti + 1) / 2 > q) --ti;
  const int tj = (int)(q - (long)ti * (ti + 1) / 2);          // tj <= ti
  const int a0 = 32 * ti, b0 = 32 * tj;
  v4f64_small acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = v4f64_small{0.0, 0.0, 0.0, 0.0};
  for (int k0 = a0; k0 < n; k0 += 8) {                        // rows below a0 are zero in columns a0.. (n % 8 == 0)
    double A[2][2], B[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const size_t row = (size_t)(k0 + 4 * h + kq) * ld;
      A[h][0] = w_entry(W, row + a0 + i);
      A[h][1] = w_entry(W, row + a0 + 16 + i);
      B[h][0] = w_entry(W, row + b0 + i);
      B[h][1] = w_entry(W, row + b0 + 16 + i);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[h][u], B[h][v], acc[u][v], 0, 0, 0);
  }
  // D[row = kq + 4 r][col = i] of block (u, v) = Ainv[a0 + 16 u + row][b0 + 16 v + col]
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + 16 * u + kq + 4 * r, b = b0 + 16 * v + i;
        const float val = (float)acc[u][v][r];
        if (ti != tj) {
          Ainv[(size_t)a * ld + b] = val;
          Ainv[(size_t)b * ld + a] = val;
        } else if (b <= a) {          // diagonal tile: its lower half decides, the upper half is its mirror image
          Ainv[(size_t)a * ld + b] = val;
          Ainv[(size_t)b * ld + a] = val;
        }
      }
}

// z = D^-1 r + P Z (y_c + y_t) for the rows of one tile, with y_c = the tile's aggregate's rows of A_c^-1 r_c computed here.
// One workgroup per tile.  Also: r.z into the slots of the NEXT iteration's scalar set, ||r||^2 of this iteration into the
// residual history, and the clearing of what the iteration after next accumulates into.
//   rc_cur   [ncp + 2 kSlots]  restriction of the current residual + the r.r slots (written by k_pcg_update_tile)
//   rc_clear [ncp + 2 kSlots]  the other parity's buffer, zeroed here
//   sc_next  slotted scalar set that receives r.z (S_RZ_OLD);  sc_clear: the set after it, zeroed here
template <int TM>
__global__ __launch_bounds__(kBlock) void k_small_z(const int32_t *__restrict__ tile_start,
                                                    const int32_t *__restrict__ agg_of_tile, const double *__restrict__ cen,
                                                    const double *__restrict__ xyz, const double *__restrict__ r,
                                                    const float *__restrict__ dinv32, const uint8_t *__restrict__ fixedbits,
                                                    const uint8_t *__restrict__ skip_rows /* may be null */,
                                                    const float *__restrict__ Ainv, int ncp, int cm,
                                                    const double *__restrict__ rc_cur, double *__restrict__ rc_clear,
                                                    const double *__restrict__ yt /* may be null */,
                                                    double *__restrict__ z, double *__restrict__ sc_next,
                                                    double *__restrict__ sc_clear, double *__restrict__ hist, int hist_slot) {
  __shared__ double ysh[16];
  __shared__ double red[kBlock / kWave];
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int a = agg_of_tile[t];
  // rows cm a .. cm a + cm - 1 of A_c^-1 times r_c: wave wv takes rows wv, wv + nw, ...; 4 columns per lane and trip
  const double2 *rc2 = reinterpret_cast<const double2 *>(rc_cur);
  for (int m = wv; m < cm; m += nw) {
    const float4 *row4 = reinterpret_cast<const float4 *>(Ainv + (size_t)(cm * a + m) * ncp);
    double s = 0.0;
#pragma unroll 4
    for (int j = lane; j < (ncp >> 2); j += 64) {
      const float4 w = row4[j];
      const double2 x0 = rc2[2 * j], x1 = rc2[2 * j + 1];
      s += (double)w.x * x0.x + (double)w.y * x0.y + (double)w.z * x1.x + (double)w.w * x1.y;
    }
    s = wave_sum(s);
    if (lane == 0) ysh[m] = s;
  }
  if (blockIdx.x == 0 && wv == 0) {            // history, scalar set of the iteration after next
    double rr = 0.0;
    for (int q = lane; q < kSlots; q += kWave) rr += rc_cur[ncp + q];
    rr = wave_sum(rr);
    if (lane == 0) hist[hist_slot] = rr;
    for (int q = lane; q < kSlots; q += kWave) {
      sc_clear[S_RZ_OLD * kSlots + q] = 0.0;
      sc_clear[S_PAP * kSlots + q] = 0.0;
    }
  }
  if (blockIdx.x == gridDim.x - 1)
    for (int e = threadIdx.x; e < ncp + 2 * kSlots; e += blockDim.x) rc_clear[e] = 0.0;
  __syncthreads();
  double C[12];
#pragma unroll
  for (int m = 0; m < 12; ++m) C[m] = 0.0;
#pragma unroll
  for (int m = 0; m < 12; ++m)
    if (m < cm) C[m] = ysh[m];
  if (yt) {
    const double *w = yt + TM * (size_t)t;
#pragma unroll
    for (int m = 0; m < TM; ++m) C[m] += w[m];
  }
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  double acc = 0.0;
  for (int64_t i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    if (skip_rows && skip_rows[i]) continue;          // eliminated node: not an unknown of this CG
    const double rx = xyz[3 * i] - c0, ry = xyz[3 * i + 1] - c1, rz = xyz[3 * i + 2] - c2;
    double zc[6] = {C[0] + (C[4] * rz - C[5] * ry), C[1] + (C[5] * rx - C[3] * rz), C[2] + (C[3] * ry - C[4] * rx),
                    C[3], C[4], C[5]};
    if constexpr (TM == 12) {
      zc[0] += C[6] * rx + 0.5 * (C[9] * ry + C[11] * rz);
      zc[1] += C[7] * ry + 0.5 * (C[9] * rx + C[10] * rz);
      zc[2] += C[8] * rz + 0.5 * (C[10] * ry + C[11] * rx);
    }
    const unsigned fb = fixedbits[i];
    const float2 *d2 = reinterpret_cast<const float2 *>(dinv32 + 6 * i);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double2 rr = load_pair(r, 3 * i + q);
      const float2 dd = d2[q];
      double2 zz;
      zz.x = dd.x * rr.x + (((fb >> (2 * q)) & 1u) ? 0.0 : zc[2 * q]);
      zz.y = dd.y * rr.y + (((fb >> (2 * q + 1)) & 1u) ? 0.0 : zc[2 * q + 1]);
      store_pair(z, 3 * i + q, zz);
      acc += rr.x * zz.x + rr.y * zz.y;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int q = 0; q < nw; ++q) s += red[q];
    unsafeAtomicAdd(sc_next + S_RZ_OLD * kSlots + (blockIdx.x & (kSlots - 1)), s);
  }
}

// after the last iteration: x += alpha p of that iteration (the K*p launch of the next one would have made it)
// (rows of eliminated nodes keep their zeros: the back-substitution after the loop fills them)
__global__ __launch_bounds__(kBlock) void k_small_final_x(int64_t N, const uint8_t *__restrict__ skip_rows /* may be null */,
                                                          const double *__restrict__ p, const double *__restrict__ sc,
                                                          double *__restrict__ x) {
  const double rz = scalar_read(sc, S_RZ_OLD), pap = scalar_read(sc, S_PAP);
  const double alpha = (pap != 0.0) ? rz / pap : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < 6 * N; i += (int64_t)gridDim.x * kBlock) {
    if (skip_rows && skip_rows[i / 6]) continue;
    x[i] += alpha * p[i];
  }
}

}  // namespace pl
