// Small dense SPD solver on the device for the coarse operator of the two-level preconditioner (n <= a few thousand).
//
// Blocked right-looking Cholesky A = L L^T (64 x 64 blocks, lower triangle of a row-major matrix; one kernel launch
// per block column: panel solve, trailing update and the next diagonal block fused), then the explicit
// inverse factor W = L^-1 (blocked, all block columns in one launch), so that applying A^-1 is two triangular GEMVs
//     t = W r ,  y = W^T t ,  r.A^-1 r = t.t
// with the same traffic as one full GEMV (W^T is stored explicitly so both are row-per-wave, coalesced).  Hand-written because vendor BLAS/LAPACK libraries loaded into a process
// that already holds PyTorch's bundled ROCm libraries resolve against the wrong versions (observed: minutes of
// start-up or a crash); it is ~1 % of a solve, so simple LDS-tiled fp64 VALU kernels are enough.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

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

// A 64 x 64 tile through registers: all 16 loads of a thread are in flight at once (a plain load -> LDS-store loop is
// compiled as sixteen dependent round trips: global_load, s_waitcnt vmcnt(0), ds_write per element).
__device__ __forceinline__ void tile_fetch(const double *__restrict__ G, int ld, double regs[kNB * kNB / kBlock]) {
#pragma unroll
  for (int q = 0; q < kNB * kNB / kBlock; ++q) {
    const int e = threadIdx.x + q * kBlock;
    regs[q] = G[(size_t)(e / kNB) * ld + e % kNB];
  }
}
__device__ __forceinline__ void tile_store_k_minor(const double regs[kNB * kNB / kBlock], double *T) {
#pragma unroll
  for (int q = 0; q < kNB * kNB / kBlock; ++q) {
    const int e = threadIdx.x + q * kBlock;
    T[(e % kNB) * kLdT + e / kNB] = regs[q];                            // T[m][row] = G[row][m]
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

constexpr int kDiagLds = kNB * kLdD + kHalf * (kHalf + 1) + kNB * 4 + 16 + 2;   // doubles of LDS scratch

// `a` = this thread's 4 x 4 tile of the block (rows 4 ty + i, columns 4 tx + c); lds = kDiagLds doubles.
__device__ __forceinline__ void chol_diag_body(double a[4][4], double *__restrict__ Akk, int ld, int kb,
                                               double *__restrict__ Dinv, int *__restrict__ info, double *lds) {
  double *Ls = lds, *T = Ls + kNB * kLdD, *colt = T + kHalf * (kHalf + 1), *dm = colt + kNB * 4;
  int &bad_s = *reinterpret_cast<int *>(dm + 16);
  static_assert(kNB == 64 && kBlock == 256, "chol_diag_body is written for 64 x 64 blocks and 256 threads");
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;    // register tile: rows 4 ty + i, columns 4 tx + c
  const bool act = ty >= tx;                                 // tiles of the lower triangle
  if (threadIdx.x == 0) bad_s = 0;
  __syncthreads();
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

__global__ __launch_bounds__(kBlock) void k_chol_diag(double *__restrict__ A, int ld, int kb,
                                                     double *__restrict__ Dinv, int *__restrict__ info) {
  __shared__ double lds[kDiagLds];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double *Akk = A + ((size_t)kb * kNB) * ld + (size_t)kb * kNB;
  double a[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) a[i][c] = ty >= tx ? Akk[(size_t)(4 * ty + i) * ld + 4 * tx + c] : 0.0;
  chol_diag_body(a, Akk, ld, kb, Dinv, info, lds);
}

// One launch per block column k (a dependent chain of tiny kernels costs ~13 us per link on this GPU, whatever
// the work): workgroup (i, j), k < j <= i <= k + bw, forms L_ik = A_ik L_kk^-T and L_jk itself, applies
// A_ij -= L_ik L_jk^T, and - for (k+1, k+1) - goes straight on to factor and invert that diagonal block, so the
// next link can start.  L_ik goes to the separate matrix Lf (written by workgroup (i, i)); column k of A stays
// readable for everybody.
__device__ __forceinline__ void chol_step_body(double *__restrict__ A, double *__restrict__ Lf, int ld, int kb, int ib,
                                               int jb, double *__restrict__ Dinv, int *__restrict__ info, double *S) {
  double *X = S, *Y = S + kNB * kLdT, *D = S + 2 * kNB * kLdT;
  static_assert(kDiagLds <= 2 * kNB * kLdT, "diagonal-block scratch must fit into X|Y");
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double *Aij = A + ((size_t)ib * kNB) * ld + (size_t)jb * kNB;
  // every global operand of this workgroup is requested up front: one memory round trip per link of the chain
  double rd[kNB * kNB / kBlock], rj[kNB * kNB / kBlock], ri[kNB * kNB / kBlock], a[4][4];
  tile_fetch(Dinv + (size_t)kb * kNB * kNB, kNB, rd);
  tile_fetch(A + ((size_t)jb * kNB) * ld + (size_t)kb * kNB, ld, rj);
  if (ib != jb) tile_fetch(A + ((size_t)ib * kNB) * ld + (size_t)kb * kNB, ld, ri);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[i][j] = Aij[(size_t)(4 * ty + i) * ld + 4 * tx + j];
  tile_store_k_minor(rd, D);                                                             // D[m][col] = Dinv_k[col][m]
  tile_store_k_minor(rj, X);                                                             // X[m][row] = A_jk[row][m]
  __syncthreads();
  {
    double l[4][4] = {};
    tile_fma(X, D, tx, ty, l);                                                           // L_jk = A_jk Dinv_k^T
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Y[(4 * tx + j) * kLdT + 4 * ty + i] = l[i][j];         // Y[m][row] = L_jk[row][m]
    if (ib == jb) {
      double *Lg = Lf + ((size_t)ib * kNB) * ld + (size_t)kb * kNB;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Lg[(size_t)(4 * ty + i) * ld + 4 * tx + j] = l[i][j];
    }
  }
  __syncthreads();
  const double *Li = Y;
  if (ib != jb) {
    tile_store_k_minor(ri, X);                                                           // X[m][row] = A_ik[row][m]
    __syncthreads();
    double l[4][4] = {};
    tile_fma(X, D, tx, ty, l);                                                           // L_ik
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) X[(4 * tx + j) * kLdT + 4 * ty + i] = l[i][j];         // X[m][row] = L_ik[row][m]
    __syncthreads();
    Li = X;
  }
  double acc[4][4] = {};
  tile_fma(Li, Y, tx, ty, acc);                                                          // L_ik L_jk^T
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[i][j] -= acc[i][j];
  if (ib == kb + 1 && jb == kb + 1) {        // look-ahead: this block is final now - factor it here
    __syncthreads();                         // X, Y are scratch from here on
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ty < tx) a[i][j] = 0.0;
    chol_diag_body(a, Aij, ld, kb + 1, Dinv, info, X);
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) Aij[(size_t)(4 * ty + i) * ld + 4 * tx + j] = a[i][j];
}

__global__ __launch_bounds__(kBlock) void k_chol_step(double *__restrict__ A, double *__restrict__ Lf, int ld, int kb,
                                                      double *__restrict__ Dinv, int *__restrict__ info) {
  __shared__ double S[3 * kNB * kLdT];                       // 101 KB of the CU's 160 KB
  const int ib = kb + 1 + blockIdx.x, jb = kb + 1 + blockIdx.y;
  if (jb > ib) return;
  chol_step_body(A, Lf, ld, kb, ib, jb, Dinv, info, S);
}

// The whole chain of block columns in ONE launch: the grid is the band (bw x bw workgroups, at most kChainMaxGrid so that
// they are all resident - one per CU at 101 KB of LDS), every workgroup walks k = 0 .. nb - 2 and a grid barrier
// (one atomic counter, release / acquire fences at agent scope) separates the steps.  What it saves is the ~13 us a
// dependent kernel launch costs per link.  A barrier that does not complete within ~2 s sets info[0] = -7 and every
// workgroup leaves (a hung kernel would take the GPU with it).
constexpr int kChainMaxGrid = 196;
__device__ __forceinline__ bool chain_barrier(unsigned *bar, unsigned target, int *info) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __threadfence();                                             // release: this workgroup's blocks are visible
    atomicAdd(bar, 1u);
    int good = 1;
    unsigned spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 26) || __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == -7) {
        atomicExch(info, -7);
        good = 0;
        break;
      }
    }
    __threadfence();                                             // acquire: drop stale lines of the vector L1
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
__global__ __launch_bounds__(kBlock) void k_chol_chain(double *__restrict__ A, double *__restrict__ Lf, int ld, int nb,
                                                       int bw, double *__restrict__ Dinv, int *__restrict__ info,
                                                       unsigned *__restrict__ bar) {
  __shared__ double S[3 * kNB * kLdT];
  const unsigned nwg = gridDim.x * gridDim.y;
  for (int kb = 0; kb + 1 < nb; ++kb) {
    const int rest = min(nb - kb - 1, bw);
    const int ib = kb + 1 + (int)blockIdx.x, jb = kb + 1 + (int)blockIdx.y;
    if ((int)blockIdx.x < rest && (int)blockIdx.y < rest && jb <= ib) chol_step_body(A, Lf, ld, kb, ib, jb, Dinv, info, S);
    if (kb + 2 < nb && !chain_barrier(bar, (unsigned)(kb + 1) * nwg, info)) return;
  }
}

// W = L^-1 column by column: block column k of W depends on nothing but L, so ONE launch computes all of W with
// no synchronisation between workgroups.  A workgroup owns kCw columns of block column k and walks down the block
// rows,  W_kk = Dinv_k,  W_ik = -Dinv_i * sum_{j = max(k, i - bw)}^{i-1} L_ij W_jk   (bw = block bandwidth of L:
// the coarse operator couples only neighbouring aggregates, so L is banded and the sums are short).
// The walk is a chain of ~bw * nb small tile products per workgroup, so it is latency that counts: the next L (or
// Dinv) tile is fetched into registers while the current one is multiplied, and (RING) the last kRing W_jk slices
// stay in LDS; without RING (bw too large) they are re-read from global memory (same workgroup wrote them).
// W^T is written alongside (the second triangular GEMV wants rows).
constexpr int kCw = 8;
constexpr int kRing = 12;

// PHASES: the walk can be cut into row ranges [row0, row1) launched one after the other, each as soon as the factorisation
// chain has finished block row row1 - 1 (dense_factor_inverse), so that only the last range runs after the chain.  A later
// range finds the last bw slices of its columns where the earlier one left them: in fp64, transposed, in the STRICTLY UPPER
// blocks (kb, ib) of `scratch` = the matrix being factored, which the factorisation never touches (it works on the lower
// triangle), so the result does not depend on where the walk was cut.
template <typename WT, bool RING>
__global__ __launch_bounds__(kBlock) void k_trtri_cols(const double *__restrict__ L, int ld, int nb, int bw,
                                                       const double *__restrict__ Dinv, WT *__restrict__ W,
                                                       WT *__restrict__ Wt, int row0 = 0, int row1 = 1 << 30,
                                                       double *__restrict__ scratch = nullptr) {
  __shared__ double At[kNB * kLdT];
  __shared__ double Ws[(RING ? kRing : 1) * kNB * kCw], S[kNB * kCw];
  constexpr int kSl = kNB / kCw;
  const int kb = blockIdx.x / kSl, c0 = (blockIdx.x % kSl) * kCw;
  row1 = min(row1, nb);
  if (kb >= row1) return;                                            // this column starts in a later range
  const int r = threadIdx.x & (kNB - 1), g = threadIdx.x / kNB;      // row, column pair (2g, 2g+1) of the slice
  const bool first = kb >= row0;                                     // the range that holds the diagonal block
  const int ib0 = first ? kb + 1 : row0;
  if (first || (RING && kb >= ib0 - bw))                             // (a later range: only if the band still reaches it)
    for (int e = threadIdx.x; e < kNB * kCw; e += kBlock) {
      const int row = e / kCw, c = e % kCw;
      const double v = Dinv[(size_t)kb * kNB * kNB + (size_t)row * kNB + c0 + c];
      if (first) {
        W[((size_t)kb * kNB + row) * ld + (size_t)kb * kNB + c0 + c] = (WT)v;
        Wt[((size_t)kb * kNB + c0 + c) * ld + (size_t)kb * kNB + row] = (WT)v;
      }
      if (RING) Ws[(kb % kRing) * kNB * kCw + e] = v;
    }
  if (RING && !first)                                                // slices an earlier range computed (fp64 copies)
    for (int jb = max(kb + 1, ib0 - bw); jb < ib0; ++jb)
      for (int e = threadIdx.x; e < kNB * kCw; e += kBlock) {
        const int c = e / kNB, row = e % kNB;
        Ws[(jb % kRing) * kNB * kCw + row * kCw + c] = scratch[((size_t)kb * kNB + c0 + c) * ld + (size_t)jb * kNB + row];
      }
  double regs[kNB * kNB / kBlock];
  if (ib0 < row1) tile_fetch(L + ((size_t)ib0 * kNB) * ld + (size_t)max(kb, ib0 - bw) * kNB, ld, regs);
  for (int ib = ib0; ib < row1; ++ib) {
    double acc0 = 0.0, acc1 = 0.0;
    const int jlo = max(kb, ib - bw);
    for (int jb = jlo; jb < ib; ++jb) {
      __syncthreads();                                                 // previous tile consumed, earlier W rows written
      tile_store_k_minor(regs, At);                                    // At[m][row] = L_ij[row][m]
      if (!RING)
        for (int e = threadIdx.x; e < kNB * kCw; e += kBlock) {
          const int m = e / kCw, c = e % kCw;
          Ws[e] = (double)W[((size_t)jb * kNB + m) * ld + (size_t)kb * kNB + c0 + c];
        }
      __syncthreads();
      if (jb + 1 < ib) tile_fetch(L + ((size_t)ib * kNB) * ld + (size_t)(jb + 1) * kNB, ld, regs);
      else tile_fetch(Dinv + (size_t)ib * kNB * kNB, kNB, regs);
      const double *w = Ws + (RING ? (jb % kRing) * kNB * kCw : 0);
#pragma unroll 8
      for (int m = 0; m < kNB; ++m) {
        const double a = At[m * kLdT + r];
        acc0 += a * w[m * kCw + 2 * g];
        acc1 += a * w[m * kCw + 2 * g + 1];
      }
    }
    __syncthreads();
    S[r * kCw + 2 * g] = acc0;
    S[r * kCw + 2 * g + 1] = acc1;
    tile_store_k_minor(regs, At);                                      // At[m][row] = Dinv_i[row][m]
    __syncthreads();
    if (ib + 1 < row1) tile_fetch(L + ((size_t)(ib + 1) * kNB) * ld + (size_t)max(kb, ib + 1 - bw) * kNB, ld, regs);
    double o0 = 0.0, o1 = 0.0;
#pragma unroll 8
    for (int m = 0; m < kNB; ++m) {
      const double a = At[m * kLdT + r];
      o0 += a * S[m * kCw + 2 * g];
      o1 += a * S[m * kCw + 2 * g + 1];
    }
    WT *Wik = W + ((size_t)ib * kNB + r) * ld + (size_t)kb * kNB + c0 + 2 * g;
    Wik[0] = (WT)(-o0);
    Wik[1] = (WT)(-o1);
    WT *Wtki = Wt + ((size_t)kb * kNB + c0 + 2 * g) * ld + (size_t)ib * kNB + r;
    Wtki[0] = (WT)(-o0);
    Wtki[ld] = (WT)(-o1);
    if (RING && scratch && row1 < nb && ib + bw >= row1) {             // a later range will want this slice
      double *sc = scratch + ((size_t)kb * kNB + c0 + 2 * g) * ld + (size_t)ib * kNB + r;
      sc[0] = -o0;
      sc[ld] = -o1;
    }
    if (RING) {
      double *slot = Ws + (ib % kRing) * kNB * kCw;                    // free: row ib + 1 needs slices ib+1-bw .. ib
      slot[r * kCw + 2 * g] = -o0;
      slot[r * kCw + 2 * g + 1] = -o1;
    }
  }
}

// The same walk on the matrix pipe for LARGE levels (band too wide for the LDS ring: the 6 144-dof level of configs[2] /
// configs[4] and of 8-GPU runs, 77 GFLOP).  A workgroup owns 16 columns; wave w computes rows 16 w .. 16 w + 15 of every
// 64 x 16 product with v_mfma_f64_16x16x4_f64: per 64 x 64 tile 16 instructions and 32 LDS reads per lane, where the vector
// form issues 128 multiply-adds and 192 LDS reads per lane - the walk was LDS-bound (1.14 us per tile product).  The fp64
// matrix rate equals the vector rate on gfx950; what the MFMA saves is operand traffic.
// Operand / result layout of v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 reg][col = lane & 15].
constexpr int kCwM = 16;
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <typename WT>
__global__ __launch_bounds__(kBlock) void k_trtri_cols_mfma(const double *__restrict__ L, int ld, int nb, int bw,
                                                            const double *__restrict__ Dinv, WT *__restrict__ W,
                                                            WT *__restrict__ Wt, int row0 = 0, int row1 = 1 << 30) {
  __shared__ double At[kNB * kLdT];
  __shared__ double Ws[kNB * kCwM], S[kNB * kCwM];
  constexpr int kSl = kNB / kCwM;
  constexpr int kWr = kNB * kCwM / kBlock;                             // entries of a 64 x 16 slice per thread
  const int kb = blockIdx.x / kSl, c0 = (blockIdx.x % kSl) * kCwM;
  row1 = min(row1, nb);
  if (kb >= row1) return;                                              // (row ranges: see k_trtri_cols; the slices are
  const bool first = kb >= row0;                                       // re-read as stored here, so nothing else is kept)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (first)
    for (int e = threadIdx.x; e < kNB * kCwM; e += kBlock) {
      const int row = e / kCwM, c = e % kCwM;
      const double v = Dinv[(size_t)kb * kNB * kNB + (size_t)row * kNB + c0 + c];
      W[((size_t)kb * kNB + row) * ld + (size_t)kb * kNB + c0 + c] = (WT)v;
      Wt[((size_t)kb * kNB + c0 + c) * ld + (size_t)kb * kNB + row] = (WT)v;
    }
  const int ib0 = first ? kb + 1 : row0;
  // slice jb of this workgroup's columns, as stored (the same workgroup wrote it, at least one barrier ago): requested one
  // product ahead, like the L tiles - a load inside the walk is a memory round trip per tile product
  auto slice_fetch = [&](int jb, WT wr[kWr]) {
#pragma unroll
    for (int q = 0; q < kWr; ++q) {
      const int e = threadIdx.x + q * kBlock;
      wr[q] = W[((size_t)jb * kNB + e / kCwM) * ld + (size_t)kb * kNB + c0 + e % kCwM];
    }
  };
  double regs[kNB * kNB / kBlock];
  WT wregs[kWr];
  __syncthreads();                                                     // (the diagonal slice above is written)
  if (ib0 < row1) {
    const int j0 = max(kb, ib0 - bw);
    tile_fetch(L + ((size_t)ib0 * kNB) * ld + (size_t)j0 * kNB, ld, regs);
    slice_fetch(j0, wregs);
  }
  // one 64 x 64 (staged k-minor in At) times 64 x 16 (row-major in B) product into this wave's 16 x 16 block
  auto product = [&](const double *B, v4f64 acc) -> v4f64 {
#pragma unroll
    for (int s4 = 0; s4 < kNB / 4; ++s4) {
      const double a = At[(4 * s4 + lk) * kLdT + 16 * wv + li];
      const double b = B[(4 * s4 + lk) * kCwM + li];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
  };
  for (int ib = ib0; ib < row1; ++ib) {
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const int jlo = max(kb, ib - bw);
    for (int jb = jlo; jb < ib; ++jb) {
      __syncthreads();                                                 // previous tile consumed, earlier W rows written
      tile_store_k_minor(regs, At);                                    // At[m][row] = L_ij[row][m]
#pragma unroll
      for (int q = 0; q < kWr; ++q) Ws[threadIdx.x + q * kBlock] = (double)wregs[q];
      __syncthreads();
      if (jb + 1 < ib) {
        tile_fetch(L + ((size_t)ib * kNB) * ld + (size_t)(jb + 1) * kNB, ld, regs);
        // (slice ib - 1 was written in the previous round of the outer loop, before at least one barrier)
        slice_fetch(jb + 1, wregs);
      } else {
        tile_fetch(Dinv + (size_t)ib * kNB * kNB, kNB, regs);
      }
      acc = product(Ws, acc);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) S[(16 * wv + lk + 4 * q) * kCwM + li] = acc[q];
    tile_store_k_minor(regs, At);                                      // At[m][row] = Dinv_i[row][m]
    __syncthreads();
    const int jn = max(kb, ib + 1 - bw);                               // first slice of the next block row
    if (ib + 1 < row1) {
      tile_fetch(L + ((size_t)(ib + 1) * kNB) * ld + (size_t)jn * kNB, ld, regs);
      if (jn < ib) slice_fetch(jn, wregs);                             // (slice ib itself is only being computed)
    }
    v4f64 o = {0.0, 0.0, 0.0, 0.0};
    o = product(S, o);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * wv + lk + 4 * q;
      W[((size_t)ib * kNB + row) * ld + (size_t)kb * kNB + c0 + li] = (WT)(-o[q]);
      Wt[((size_t)kb * kNB + c0 + li) * ld + (size_t)ib * kNB + row] = (WT)(-o[q]);
    }
    if (ib + 1 < row1 && jn == ib) {                                   // band of one block: the slice just written
      __syncthreads();
      slice_fetch(ib, wregs);
    }
  }
}

// bfloat16 storage of the inverse factor (large dense levels: at 6 144 dofs the two triangular GEMVs read 2 x 75 MB in
// fp32 - 44 of the 354 us of a 100^3 BCC iteration; bf16 halves that).  fp32's exponent range (entries of W span many
// decades between translational and rotational modes), 8 bits of mantissa: W16^T W16 is still exactly symmetric positive
// definite and the same operator in every iteration, which is all PCG asks of a preconditioner.
struct bf16_t {
  uint16_t v;
  bf16_t() = default;
  __device__ explicit bf16_t(double d) {
    const uint32_t u = __float_as_uint((float)d);
    v = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);                     // round to nearest even
  }
  __device__ explicit operator double() const { return (double)__uint_as_float((uint32_t)v << 16); }
};

// Row dot products of the triangular GEMVs: columns [c_lo, c_hi) of one row, strided over the wave.
// W may be stored in fp32 (the preconditioner is then W32^T W32 - still exactly symmetric positive definite - and
// the GEMVs move half the bytes); the vectors and the accumulation stay fp64.
__device__ __forceinline__ double row_dot(const double *__restrict__ Wr, const double *__restrict__ v, int c_lo,
                                          int c_hi, int lane) {
  const double2 *W2 = reinterpret_cast<const double2 *>(Wr);
  const double2 *v2 = reinterpret_cast<const double2 *>(v);
  double s = 0.0;
#pragma unroll 4
  for (int j = (c_lo >> 1) + lane; j < ((c_hi + 1) >> 1); j += 64) {
    const double2 w = W2[j], x = v2[j];
    s += w.x * x.x + w.y * x.y;
  }
  return s;
}
__device__ __forceinline__ double row_dot(const float *__restrict__ Wr, const double *__restrict__ v, int c_lo,
                                          int c_hi, int lane) {
  const float4 *W4 = reinterpret_cast<const float4 *>(Wr);
  const double2 *v2 = reinterpret_cast<const double2 *>(v);
  double s = 0.0;
  // (tried: all loads of up to 8 trips of a row in flight at once, clamped or predicated - a row of the 1 536-dof operator in
  // ONE memory round trip instead of the 2 + 4 this loop compiles to: the PCG iteration got 1 us SLOWER both ways)
#ifndef PL_F32_UNROLL
#define PL_F32_UNROLL 4
#endif
#pragma unroll PL_F32_UNROLL
  for (int j = (c_lo >> 2) + lane; j < ((c_hi + 3) >> 2); j += 64) {
    const float4 w = W4[j];
    const double2 x0 = v2[2 * j], x1 = v2[2 * j + 1];
    s += (double)w.x * x0.x + (double)w.y * x0.y + (double)w.z * x1.x + (double)w.w * x1.y;
  }
  return s;
}

__device__ __forceinline__ double row_dot(const bf16_t *__restrict__ Wr, const double *__restrict__ v, int c_lo,
                                          int c_hi, int lane) {
  const uint4 *W8 = reinterpret_cast<const uint4 *>(Wr);                     // eight entries per 16-byte load
  const double2 *v2 = reinterpret_cast<const double2 *>(v);
  double s = 0.0;
#ifndef PL_BF16_UNROLL
#define PL_BF16_UNROLL 1       // (100^3 BCC iteration with 1 / 2 / 4 / 8 trips in flight: 281.4 / 283.4 / 286.9 / 288.2 us)
#endif
#pragma unroll PL_BF16_UNROLL
  for (int j = (c_lo >> 3) + lane; j < ((c_hi + 7) >> 3); j += 64) {
    const uint4 w = W8[j];
    const double2 x0 = v2[4 * j], x1 = v2[4 * j + 1], x2 = v2[4 * j + 2], x3 = v2[4 * j + 3];
    s += (double)__uint_as_float(w.x << 16) * x0.x + (double)__uint_as_float(w.x & 0xFFFF0000u) * x0.y +
         (double)__uint_as_float(w.y << 16) * x1.x + (double)__uint_as_float(w.y & 0xFFFF0000u) * x1.y +
         (double)__uint_as_float(w.z << 16) * x2.x + (double)__uint_as_float(w.z & 0xFFFF0000u) * x2.y +
         (double)__uint_as_float(w.w << 16) * x3.x + (double)__uint_as_float(w.w & 0xFFFF0000u) * x3.y;
  }
  return s;
}

// (Round 3, tried and dropped: FOUR rows per wave for the bfloat16 W of a 6 144-dof level, the fp64 vector entries - 8 bytes
// against 2 of W - loaded once for the four rows: 100^3 BCC iteration 285 -> 296 us; a quarter of the waves, each with a four
// times longer dependent loop, loses more than the vector L1 traffic it saves.)
// t = W r (W lower triangular, one wave per row); dot_out[slot] += t.t
// (rows are zero right of the diagonal and n is a multiple of 64, so reading a few columns past it is harmless)
template <typename WT>
__global__ __launch_bounds__(kBlock) void k_tri_gemv(int n, const WT *__restrict__ W, int ld,
                                                     const double *__restrict__ r, double *__restrict__ t,
                                                     double *__restrict__ dot_out, const double *__restrict__ add0) {
  __shared__ double red[kBlock / kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // longest rows first (the last workgroups to start then hold the shortest rows: at 6 144 dofs the kernel's tail was 2 us)
  const int row = n - 1 - (blockIdx.x * (kBlock / kWave) + wv);
  double sq = 0.0;
  if (row >= 0) {
    const double s = wave_sum(row_dot(W + (size_t)row * ld, r, 0, row + 1, lane));
    if (lane == 0) {
      t[row] = s;
      sq = s * s;
    }
  }
  if (dot_out) {
    double extra = 0.0;
    if (blockIdx.x == 0 && add0 && wv == 0) {              // add0 = a slotted scalar (e.g. r.D^-1 r), summed by wave 0
      for (int q = lane; q < kSlots; q += kWave) extra += add0[q];
      extra = wave_sum(extra);
    }
    if (lane == 0) red[wv] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = extra;
      for (int q = 0; q < kBlock / kWave; ++q) s += red[q];
      unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
    }
  }
}

// y = W^T t with the explicitly stored transpose (upper triangular rows, one wave per row).
template <typename WT>
__global__ __launch_bounds__(kBlock) void k_tri_gemv_upper(int n, const WT *__restrict__ Wt, int ld,
                                                           const double *__restrict__ t, double *__restrict__ y) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * (kBlock / kWave) + wv;
  if (row >= n) return;
  const double s = wave_sum(row_dot(Wt + (size_t)row * ld, t, row, n, lane));   // zero left of the diagonal
  if (lane == 0) y[row] = s;
}

// Host driver: factor A (n x n, ld, n multiple of kNB; lower triangle used; its diagonal blocks end up holding
// L_kk, the off-diagonal blocks of L go to Lf) and build W = L^-1.
// bw: block bandwidth of A (blocks (i, j) with i - j > bw are zero), nb for a full matrix.
// Row ranges of the inverse factor launched on a second stream while the chain is still running (rows = 0: one launch
// after the chain, as until round 4).
struct TrtriPhases {
  hipStream_t stream = nullptr;
  hipEvent_t ev_go = nullptr, ev_done = nullptr;
  int rows = 0;
};
template <typename WT>
inline void dense_factor_inverse(double *A, double *Lf, WT *W, WT *Wt, double *Dinv, int n, int ld, int *info,
                                 int bw, hipStream_t s, const std::function<void(int)> &after_chol = nullptr,
                                 unsigned *bar = nullptr, TrtriPhases ph = TrtriPhases()) {
  const int nb = n / kNB;
  if (bw <= 0 || bw > nb) bw = nb;
  if (ph.rows <= 0 || !ph.ev_go || !ph.ev_done) ph.stream = nullptr;
  int done = 0;                                                          // block rows of W already launched
  auto trtri = [&](hipStream_t st, int row0, int row1) {
    if (bw + 1 <= kRing)
      hipLaunchKernelGGL((k_trtri_cols<WT, true>), dim3(std::min(nb, row1) * (kNB / kCw)), dim3(kBlock), 0, st, Lf, ld, nb, bw,
                         Dinv, W, Wt, row0, row1, A);
    else
      hipLaunchKernelGGL((k_trtri_cols_mfma<WT>), dim3(std::min(nb, row1) * (kNB / kCwM)), dim3(kBlock), 0, st, Lf, ld, nb, bw,
                         Dinv, W, Wt, row0, row1);
  };
  if (after_chol) after_chol(0);       // (head of the chain: work that may run beside it on reserved CUs)
  hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(kBlock), 0, s, A, ld, 0, Dinv, info);
  const int g = std::min(nb - 1, bw);
  if (bar && nb > 2 && g * g <= kChainMaxGrid) {   // the band fits the chip: one persistent launch for the whole chain
    (void)hipMemsetAsync(bar, 0, sizeof(unsigned), s);
    hipLaunchKernelGGL(k_chol_chain, dim3(g, g), dim3(kBlock), 0, s, A, Lf, ld, nb, bw, Dinv, info, bar);
  } else {
    for (int k = 0; k + 1 < nb; ++k) {
      const int rest = std::min(nb - k - 1, bw);                         // the fill stays inside the band
      hipLaunchKernelGGL(k_chol_step, dim3(rest, rest), dim3(kBlock), 0, s, A, Lf, ld, k, Dinv, info);
      // the inverse factor walks BEHIND the chain: block rows [done, k + 2) of L and their diagonal inverses are final now
      if (ph.stream && k + 2 < nb && (k + 2 - done) >= ph.rows && (nb - (k + 2)) >= ph.rows / 2) {
        (void)hipEventRecord(ph.ev_go, s);
        (void)hipStreamWaitEvent(ph.stream, ph.ev_go, 0);
        trtri(ph.stream, done, k + 2);
        (void)hipEventRecord(ph.ev_done, ph.stream);
        done = k + 2;
      }
    }
  }
  // after_chol: bulk work of the caller that should NOT run beside this latency-bound chain of dependent launches (every
  // link is slower next to a bandwidth-heavy kernel: measured 44 -> 57 us, also when released at 55 % of the chain) but
  // beside the single-launch inverse factor that follows
  if (after_chol) after_chol(1);
  if (done > 0) (void)hipStreamWaitEvent(s, ph.ev_done, 0);
  trtri(s, done, nb);
}

// y = G r with the explicitly stored symmetric G = A^-1 (fp32, pl_small.h k_dense_explicit_inverse), one wave per row;
// dot_out[slot] += r.y (+ add0 once).  ONE launch instead of the two triangular GEMVs: at <= 2 048 dofs those are two latency
// chains of ~4.9 us each, whatever they read.
__global__ __launch_bounds__(kBlock) void k_full_gemv(int n, const float *__restrict__ G, int ld, const double *__restrict__ r,
                                                      double *__restrict__ y, double *__restrict__ dot_out,
                                                      const double *__restrict__ add0) {
  __shared__ double red[kBlock / kWave];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int row = blockIdx.x * (kBlock / kWave) + wv;
  double part = 0.0;
  if (row < n) {
    const double s = wave_sum(row_dot(G + (size_t)row * ld, r, 0, n, lane));
    if (lane == 0) {
      y[row] = s;
      part = s * r[row];
    }
  }
  if (dot_out) {
    double extra = 0.0;
    if (blockIdx.x == 0 && add0 && wv == 0) {
      for (int q = lane; q < kSlots; q += kWave) extra += add0[q];
      extra = wave_sum(extra);
    }
    if (lane == 0) red[wv] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = extra;
      for (int q = 0; q < kBlock / kWave; ++q) s += red[q];
      unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
    }
  }
}

// y = A^-1 r through W; dot_out[kSlots] += r.A^-1 r (+ *add0 once)
template <typename WT>
inline void dense_apply(const WT *W, const WT *Wt, int n, int ld, const double *r, double *t, double *y,
                        double *dot_out, const double *add0, hipStream_t s) {
  hipLaunchKernelGGL(k_tri_gemv<WT>, dim3((n + 3) / 4), dim3(kBlock), 0, s, n, W, ld, r, t, dot_out, add0);
  hipLaunchKernelGGL(k_tri_gemv_upper<WT>, dim3((n + 3) / 4), dim3(kBlock), 0, s, n, Wt, ld, t, y);
}

}  // namespace pl
