// Small dense SPD solver on the device for the coarse operator of the two-level preconditioner (n <= a few thousand).
//
// Blocked right-looking Cholesky A = L L^T (64 x 64 blocks, lower triangle of a row-major matrix), then the explicit
// inverse factor W = L^-1 (blocked, all block columns in one launch), so that applying A^-1 is two triangular GEMVs
//     t = W r ,  y = W^T t ,  r.A^-1 r = t.t
// with the same traffic as one full GEMV (W^T is stored explicitly so both are row-per-wave, coalesced).  Hand-written because vendor BLAS/LAPACK libraries loaded into a process
// that already holds PyTorch's bundled ROCm libraries resolve against the wrong versions (observed: minutes of
// start-up or a crash); it is ~1 % of a solve, so simple LDS-tiled fp64 VALU kernels are enough.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pl_kernels.h"

namespace pl {

constexpr int kNB = 64;          // block size
constexpr int kLdT = kNB + 2;    // LDS row pitch (doubles) of the staged tiles

// acc[4][4] += sum_k At[k][4*ty + a] * Bt[k][4*tx + b]   (both tiles staged k-major in LDS)
__device__ __forceinline__ void tile_fma(const double *At, const double *Bt, int tx, int ty, double acc[4][4]) {
#pragma unroll 8
  for (int k = 0; k < kNB; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = At[k * kLdT + 4 * ty + q];
      b[q] = Bt[k * kLdT + 4 * tx + q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
  }
}

// stage a 64 x 64 global tile into LDS as T[k][row] = G[row][k]  (i.e. k-major: the GEMM's inner index first)
__device__ __forceinline__ void stage_rows_as_k_minor(const double *G, int ld, double *T) {
  for (int e = threadIdx.x; e < kNB * kNB; e += kBlock) {
    const int row = e / kNB, k = e % kNB;
    T[k * kLdT + row] = G[(size_t)row * ld + k];
  }
}
// stage as T[k][col] = G[k][col]
__device__ __forceinline__ void stage_rows_as_k_major(const double *G, int ld, double *T) {
  for (int e = threadIdx.x; e < kNB * kNB; e += kBlock) {
    const int k = e / kNB, col = e % kNB;
    T[k * kLdT + col] = G[(size_t)k * ld + col];
  }
}

// Diagonal block: L_kk and its inverse by ONE workgroup.
// Cholesky: right-looking over tile columns of width 4, every thread keeps a 4 x 4 register tile of the block; per
// step the diagonal tile is factored and inverted by its owner, the tiles below it are solved by theirs and travel
// through LDS (two barriers per step, 32 in all), the rank-4 update runs in registers.
// Inverse: X = L^-1 by halves - the two 32 x 32 diagonal sub-blocks are inverted by one wave each (column per lane,
// L broadcast from LDS), then X21 = -X22 (L21 X11) as two small LDS GEMMs over all 256 threads.
// info[0] != 0 if a pivot is not positive.
constexpr int kLdD = kNB + 1;    // LDS pitch of the diagonal block
constexpr int kHalf = kNB / 2;

__global__ __launch_bounds__(kBlock) void k_chol_diag(double *__restrict__ A, int ld, int kb,
                                                     double *__restrict__ Dinv, int *__restrict__ info) {
  __shared__ double Ls[kNB * kLdD];
  __shared__ double T[kHalf * (kHalf + 1)];
  __shared__ double colt[kNB * 4];
  __shared__ double dm[16];
  __shared__ int bad_s;
  static_assert(kNB == 64 && kBlock == 256, "k_chol_diag is written for 64 x 64 blocks and 256 threads");
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;    // register tile: rows 4 ty + i, columns 4 tx + c
  const bool act = ty >= tx;                                 // tiles of the lower triangle
  double *Akk = A + ((size_t)kb * kNB) * ld + (size_t)kb * kNB;
  double a[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[i][c] = act ? Akk[(size_t)(4 * ty + i) * ld + 4 * tx + c] : 0.0;
  if (threadIdx.x == 0) bad_s = 0;
  for (int jt = 0; jt < kNB / 4; ++jt) {                     // four columns (one tile column) per step
    if (tx == jt && ty == jt) {                              // diagonal tile: 4 x 4 Cholesky + inverse in registers
      double l[4][4], m[4][4];
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double d = a[c][c];
#pragma unroll
        for (int q = 0; q < c; ++q) d -= l[c][q] * l[c][q];
        ok = ok && (d > 0.0);
        const double ri = rsqrt(d > 0.0 ? d : 1.0);
        l[c][c] = (d > 0.0 ? d : 1.0) * ri;
        m[c][c] = ri;
#pragma unroll
        for (int i = c + 1; i < 4; ++i) {
          double v = a[i][c];
#pragma unroll
          for (int q = 0; q < c; ++q) v -= l[i][q] * l[c][q];
          l[i][c] = v * ri;
        }
      }
      if (!ok) bad_s = 1;
#pragma unroll
      for (int c = 0; c < 4; ++c)                            // M = L^-1 (lower), column by column
#pragma unroll
        for (int i = c + 1; i < 4; ++i) {
          double v = 0.0;
#pragma unroll
          for (int q = c; q < i; ++q) v -= l[i][q] * m[q][c];
          m[i][c] = v * m[i][i];
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a[i][c] = c <= i ? l[i][c] : 0.0;
          dm[4 * i + c] = c <= i ? m[i][c] : 0.0;
        }
    }
    __syncthreads();
    if (tx == jt && ty > jt) {                               // tiles below it: L_ik = A_ik M^T, published row-wise
      double mm[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) mm[e] = dm[e];
      double l[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double v = 0.0;
#pragma unroll
          for (int q = 0; q <= c; ++q) v += a[i][q] * mm[4 * c + q];
          l[i][c] = v;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a[i][c] = l[i][c];
          colt[(4 * ty + i) * 4 + c] = l[i][c];
        }
    }
    __syncthreads();
    if (act && tx > jt) {                                    // rank-4 update of everything right of the tile column
      double rv[4][4], cv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          rv[i][q] = colt[(4 * ty + i) * 4 + q];
          cv[i][q] = colt[(4 * tx + i) * 4 + q];
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) a[i][c] -= rv[i][q] * cv[c][q];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int row = 4 * ty + i, cx = 4 * tx + c;
      const double v = (act && cx <= row) ? a[i][c] : 0.0;
      Ls[row * kLdD + cx] = v;
      Akk[(size_t)row * ld + cx] = v;                        // L_kk (upper part zero)
    }
  __syncthreads();
  if (bad_s && threadIdx.x == 0) info[0] = kb * kNB + 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv < 2 && lane < kHalf) {                              // X11 (wave 0) and X22 (wave 1), in place
    const int o = kHalf * wv;
    double x[kHalf];
#pragma unroll
    for (int i = 0; i < kHalf; ++i) {
      double sacc = (lane == i) ? 1.0 : 0.0;
#pragma unroll
      for (int m = 0; m < i; ++m) sacc -= Ls[(o + i) * kLdD + o + m] * x[m];
      x[i] = sacc / Ls[(o + i) * kLdD + o + i];
    }
#pragma unroll
    for (int i = 0; i < kHalf; ++i) Ls[(o + i) * kLdD + o + lane] = x[i];
  }
  __syncthreads();
  const int r = threadIdx.x & (kHalf - 1), cg = threadIdx.x / kHalf;    // row, group of 4 columns
  {
    double t4[4] = {0, 0, 0, 0};                             // T = L21 X11
#pragma unroll 8
    for (int m = 0; m < kHalf; ++m) {
      const double l = Ls[(kHalf + r) * kLdD + m];
#pragma unroll
      for (int q = 0; q < 4; ++q) t4[q] += l * Ls[m * kLdD + 4 * cg + q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) T[r * (kHalf + 1) + 4 * cg + q] = t4[q];
  }
  __syncthreads();
  {
    double o4[4] = {0, 0, 0, 0};                             // X21 = -X22 T
#pragma unroll 8
    for (int m = 0; m < kHalf; ++m) {
      const double xv = Ls[(kHalf + r) * kLdD + kHalf + m];
#pragma unroll
      for (int q = 0; q < 4; ++q) o4[q] += xv * T[m * (kHalf + 1) + 4 * cg + q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Ls[(kHalf + r) * kLdD + 4 * cg + q] = -o4[q];   // nobody reads L21 any more
  }
  __syncthreads();
  double *Dk = Dinv + (size_t)kb * kNB * kNB;
  for (int e = threadIdx.x; e < kNB * kNB; e += kBlock) Dk[e] = Ls[(e / kNB) * kLdD + e % kNB];
}

// Panel: A_ik <- A_ik * L_kk^-T = A_ik * Dinv_k^T for block rows i > kb.  grid.x = nb - kb - 1.
__global__ __launch_bounds__(kBlock) void k_chol_panel(double *__restrict__ A, int ld, int kb,
                                                       const double *__restrict__ Dinv) {
  __shared__ double At[kNB * kLdT], Bt[kNB * kLdT];
  const int ib = kb + 1 + blockIdx.x;
  double *Aik = A + ((size_t)ib * kNB) * ld + (size_t)kb * kNB;
  stage_rows_as_k_minor(Aik, ld, At);                                   // At[k][row] = A_ik[row][k]
  stage_rows_as_k_minor(Dinv + (size_t)kb * kNB * kNB, kNB, Bt);        // Bt[k][col] = Dinv[col][k]  (C = A * Dinv^T)
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4] = {};
  tile_fma(At, Bt, tx, ty, acc);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Aik[(size_t)(4 * ty + i) * ld + 4 * tx + j] = acc[i][j];
}

// Trailing update: A_ij -= A_ik * A_jk^T for kb < j <= i.  grid = (nb-kb-1, nb-kb-1), blocks with j > i exit.
__global__ __launch_bounds__(kBlock) void k_chol_trail(double *__restrict__ A, int ld, int kb) {
  const int ib = kb + 1 + blockIdx.x, jb = kb + 1 + blockIdx.y;
  if (jb > ib) return;
  __shared__ double At[kNB * kLdT], Bt[kNB * kLdT];
  stage_rows_as_k_minor(A + ((size_t)ib * kNB) * ld + (size_t)kb * kNB, ld, At);
  stage_rows_as_k_minor(A + ((size_t)jb * kNB) * ld + (size_t)kb * kNB, ld, Bt);
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double acc[4][4] = {};
  tile_fma(At, Bt, tx, ty, acc);
  double *Aij = A + ((size_t)ib * kNB) * ld + (size_t)jb * kNB;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Aij[(size_t)(4 * ty + i) * ld + 4 * tx + j] -= acc[i][j];
}

// W = L^-1 column by column: block column k of W depends on nothing but L, so ONE launch computes all of W with
// no synchronisation between workgroups.  A workgroup owns kCw columns of block column k and walks down the block
// rows,  W_kk = Dinv_k,  W_ik = -Dinv_i * sum_{j = max(k, i - bw)}^{i-1} L_ij W_jk   (bw = block bandwidth of L:
// the coarse operator couples only neighbouring aggregates, so L is banded and the sums are short),
// re-reading its own earlier W_jk slices from global memory (same workgroup: visible after the barrier).
// W^T is written alongside (the second triangular GEMV wants rows).
constexpr int kCw = 8;

__global__ __launch_bounds__(kBlock) void k_trtri_cols(const double *__restrict__ L, int ld, int nb, int bw,
                                                       const double *__restrict__ Dinv, double *__restrict__ W,
                                                       double *__restrict__ Wt) {
  __shared__ double At[kNB * kLdT];
  __shared__ double Ws[kNB * kCw], S[kNB * kCw];
  constexpr int kSl = kNB / kCw;
  const int kb = blockIdx.x / kSl, c0 = (blockIdx.x % kSl) * kCw;
  const int r = threadIdx.x & (kNB - 1), g = threadIdx.x / kNB;      // row, column pair (2g, 2g+1) of the slice
  for (int e = threadIdx.x; e < kNB * kCw; e += kBlock) {
    const int row = e / kCw, c = e % kCw;
    const double v = Dinv[(size_t)kb * kNB * kNB + (size_t)row * kNB + c0 + c];
    W[((size_t)kb * kNB + row) * ld + (size_t)kb * kNB + c0 + c] = v;
    Wt[((size_t)kb * kNB + c0 + c) * ld + (size_t)kb * kNB + row] = v;
  }
  for (int ib = kb + 1; ib < nb; ++ib) {
    double acc0 = 0.0, acc1 = 0.0;
    const int jlo = max(kb, ib - bw);
    for (int jb = jlo; jb < ib; ++jb) {
      __syncthreads();                                                 // previous tile consumed, earlier W rows written
      stage_rows_as_k_minor(L + ((size_t)ib * kNB) * ld + (size_t)jb * kNB, ld, At);    // At[m][row] = L_ij[row][m]
      for (int e = threadIdx.x; e < kNB * kCw; e += kBlock) {
        const int m = e / kCw, c = e % kCw;
        Ws[e] = W[((size_t)jb * kNB + m) * ld + (size_t)kb * kNB + c0 + c];
      }
      __syncthreads();
#pragma unroll 8
      for (int m = 0; m < kNB; ++m) {
        const double a = At[m * kLdT + r];
        acc0 += a * Ws[m * kCw + 2 * g];
        acc1 += a * Ws[m * kCw + 2 * g + 1];
      }
    }
    __syncthreads();
    S[r * kCw + 2 * g] = acc0;
    S[r * kCw + 2 * g + 1] = acc1;
    stage_rows_as_k_minor(Dinv + (size_t)ib * kNB * kNB, kNB, At);                      // At[m][row] = Dinv_i[row][m]
    __syncthreads();
    double o0 = 0.0, o1 = 0.0;
#pragma unroll 8
    for (int m = 0; m < kNB; ++m) {
      const double a = At[m * kLdT + r];
      o0 += a * S[m * kCw + 2 * g];
      o1 += a * S[m * kCw + 2 * g + 1];
    }
    double *Wik = W + ((size_t)ib * kNB + r) * ld + (size_t)kb * kNB + c0 + 2 * g;
    Wik[0] = -o0;
    Wik[1] = -o1;
    double *Wtki = Wt + ((size_t)kb * kNB + c0 + 2 * g) * ld + (size_t)ib * kNB + r;
    Wtki[0] = -o0;
    Wtki[ld] = -o1;
  }
}

// t = W r (W lower triangular, one wave per row); dot_out[slot] += t.t
__global__ __launch_bounds__(kBlock) void k_tri_gemv(int n, const double *__restrict__ W, int ld,
                                                     const double *__restrict__ r, double *__restrict__ t,
                                                     double *__restrict__ dot_out, const double *__restrict__ add0) {
  __shared__ double red[kBlock / kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * (kBlock / kWave) + wv;
  double sq = 0.0;
  if (row < n) {
    const double2 *Wr = reinterpret_cast<const double2 *>(W + (size_t)row * ld);
    const double2 *r2 = reinterpret_cast<const double2 *>(r);
    double s = 0.0;
    const int n2 = (row >> 1) + 1;                 // double2 pairs covering columns 0..row (W is zero above the diagonal)
#pragma unroll 4
    for (int j = lane; j < n2; j += 64) {
      const double2 w = Wr[j], v = r2[j];
      s += w.x * v.x + w.y * v.y;
    }
    s = wave_sum(s);
    if (lane == 0) {
      t[row] = s;
      sq = s * s;
    }
  }
  if (dot_out) {
    if (lane == 0) red[wv] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int q = 0; q < kBlock / kWave; ++q) s += red[q];
      if (blockIdx.x == 0 && add0)
        for (int q = 0; q < kSlots; ++q) s += add0[q];     // add0 = a slotted scalar (e.g. r.D^-1 r)
      unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
    }
  }
}

// y = W^T t with the explicitly stored transpose (upper triangular rows, one wave per row).
__global__ __launch_bounds__(kBlock) void k_tri_gemv_upper(int n, const double *__restrict__ Wt, int ld,
                                                           const double *__restrict__ t, double *__restrict__ y) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * (kBlock / kWave) + wv;
  if (row >= n) return;
  const double2 *Wr = reinterpret_cast<const double2 *>(Wt + (size_t)row * ld);
  const double2 *t2 = reinterpret_cast<const double2 *>(t);
  double s = 0.0;
  const int n2 = n >> 1;                           // n is a multiple of 64; W^T is zero below the diagonal
#pragma unroll 4
  for (int j = (row >> 1) + lane; j < n2; j += 64) {
    const double2 w = Wr[j], v = t2[j];
    s += w.x * v.x + w.y * v.y;
  }
  s = wave_sum(s);
  if (lane == 0) y[row] = s;
}

// Host driver: factor A (n x n, ld, n multiple of kNB; lower triangle used, overwritten by L) and build W = L^-1.
// bw: block bandwidth of A (blocks (i, j) with i - j > bw are zero), nb for a full matrix.
inline void dense_factor_inverse(double *A, double *W, double *Wt, double *Dinv, int n, int ld, int *info, int bw,
                                 hipStream_t s) {
  const int nb = n / kNB;
  if (bw <= 0 || bw > nb) bw = nb;
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(kBlock), 0, s, A, ld, k, Dinv, info);
    const int rest = std::min(nb - k - 1, bw);                         // the fill stays inside the band
    if (rest > 0) {
      hipLaunchKernelGGL(k_chol_panel, dim3(rest), dim3(kBlock), 0, s, A, ld, k, Dinv);
      hipLaunchKernelGGL(k_chol_trail, dim3(rest, rest), dim3(kBlock), 0, s, A, ld, k);
    }
  }
  hipLaunchKernelGGL(k_trtri_cols, dim3(nb * (kNB / kCw)), dim3(kBlock), 0, s, A, ld, nb, bw, Dinv, W, Wt);
}

// y = A^-1 r through W; dot_out[kSlots] += r.A^-1 r (+ *add0 once)
inline void dense_apply(const double *W, const double *Wt, int n, int ld, const double *r, double *t, double *y,
                        double *dot_out, const double *add0, hipStream_t s) {
  hipLaunchKernelGGL(k_tri_gemv, dim3((n + 3) / 4), dim3(kBlock), 0, s, n, W, ld, r, t, dot_out, add0);
  hipLaunchKernelGGL(k_tri_gemv_upper, dim3((n + 3) / 4), dim3(kBlock), 0, s, n, Wt, ld, t, y);
}

}  // namespace pl
