// Domain-decomposition (Schur) operator of pyLatticeSim on the device:  y = sum_cells B_c^T S_c B_c x.
// Replaces LatticeSim.calculate_reaction_force_global / update_reaction_force_each_cell / solve_sub_problem
// (lattice_sim.py:1180-1252) + Cell.set_displacement_at_boundary_nodes / set_reaction_force_on_nodes
// (cell.py:684-750): per cell gather the 6 n_b boundary values, multiply by the dense cell Schur complement, scatter-add.
// Cells with the same (geometry, radii) share one matrix (the reference groups them the same way,
// lattice_sim.py:846-919), so S is a small palette; it is stored TRANSPOSED so that lane i, which owns row i, reads
// consecutive addresses.  One wave per cell; rows beyond 64 (n_b > 10) are looped.
#pragma once
#include <hip/hip_runtime.h>

#include "pl_kernels.h"
#include "pl_coarse.h"

namespace pl {

constexpr int kDdmMaxM = 6 * 27;   // up to 27 boundary nodes per cell

__global__ __launch_bounds__(kBlock) void k_ddm_apply(int64_t C, int nb, const int32_t *__restrict__ cell_nodes,
                                                      const int32_t *__restrict__ cell_S,
                                                      const double *__restrict__ St, const double *__restrict__ x,
                                                      double *__restrict__ y) {
  __shared__ double ucell[kBlock / kWave][kDdmMaxM];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * (kBlock / kWave) + wv;
  if (c >= C) return;                      // whole wave exits together
  const int m = 6 * nb;
  const int32_t *nodes = cell_nodes + c * nb;
  double part = 0.0;
  for (int i = lane; i < m; i += 64) {
    const double v = x[6 * (int64_t)nodes[i / 6] + i % 6];
    ucell[wv][i] = v;
    part += v;
  }
  // the reference skips the product when np.sum(displacement_cell) == 0 (lattice_sim.py:1239): same test here
  const double tot = wave_sum(part);
  if (tot == 0.0) return;
  const double *S = St + (size_t)cell_S[c] * m * m;
  for (int i = lane; i < m; i += 64) {
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc += S[(size_t)j * m + i] * ucell[wv][j];     // St[j][i] = S[i][j]
    unsafeAtomicAdd(y + 6 * (int64_t)nodes[i / 6] + i % 6, acc);
  }
}

// Atomic-free form of the same product (the one the solver uses): every cell writes its m local results to a staging
// row (coalesced), then every node sums the entries of the cells it belongs to through a node -> (cell, slot) list built
// once on the host.  No memset of y, no f64 global atomics (MI355X_MICROARCH.md prices scattered ones at ~1/17 of the
// streaming rate), and the result is bitwise reproducible.
__global__ __launch_bounds__(kBlock) void k_ddm_cell_product(int64_t C, int nb, const int32_t *__restrict__ cell_nodes,
                                                             const int32_t *__restrict__ cell_S,
                                                             const double *__restrict__ St,
                                                             const double *__restrict__ x,
                                                             double *__restrict__ stage) {
  __shared__ double ucell[kBlock / kWave][kDdmMaxM];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * (kBlock / kWave) + wv;
  if (c >= C) return;
  const int m = 6 * nb;
  const int32_t *nodes = cell_nodes + c * nb;
  double part = 0.0;
  for (int i = lane; i < m; i += 64) {
    const double v = x[6 * (int64_t)nodes[i / 6] + i % 6];
    ucell[wv][i] = v;
    part += v;
  }
  const double tot = wave_sum(part);
  const bool skip = tot == 0.0;            // lattice_sim.py:1239: np.sum(displacement_cell) == 0 -> zero reactions
  const double *S = St + (size_t)cell_S[c] * m * m;
  for (int i = lane; i < m; i += 64) {
    double acc = 0.0;
    if (!skip)
      for (int j = 0; j < m; ++j) acc += S[(size_t)j * m + i] * ucell[wv][j];
    stage[c * m + i] = acc;
  }
}
// Same product with the cell matrix staged in LDS: cells are visited in the order of their matrix id (`order`, sorted
// on the host), a workgroup takes kDdmChunk consecutive ones and (re)loads S^T into LDS only when the id changes - on a
// lattice with few distinct cells that is once per workgroup instead of once per cell from L2 (18 KB per BCC cell:
// the L2 read of S, not the arithmetic, bounded k_ddm_cell_product).  Needs 8 m^2 bytes of LDS (m <= 84).
constexpr int kDdmChunk = 32;
__global__ __launch_bounds__(kBlock) void k_ddm_cell_product_lds(int64_t C, int nb, const int32_t *__restrict__ order,
                                                                 const int32_t *__restrict__ cell_nodes,
                                                                 const int32_t *__restrict__ cell_S,
                                                                 const double *__restrict__ St,
                                                                 const double *__restrict__ x,
                                                                 double *__restrict__ stage) {
  extern __shared__ double lds[];                 // [m*m] S^T, then [waves][m] cell vectors
  const int m = 6 * nb;
  double *Ss = lds, *ucell = lds + m * m;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t q0 = (int64_t)blockIdx.x * kDdmChunk;
  int loaded = -1;
  for (int round = 0; round < kDdmChunk / (kBlock / kWave); ++round) {
    const int64_t q = q0 + round * (kBlock / kWave) + wv;
    const int64_t c = q < C ? order[q] : -1;
    // all waves of the round must agree on the matrix in LDS: take the first cell's id, cells with another id in the
    // same round wait for a reload (they are contiguous in `order`, so this happens at most once per id boundary)
    const int64_t qf = q0 + round * (kBlock / kWave);
    if (qf >= C) break;
    const int want = cell_S[order[qf]];
    if (want != loaded) {
      __syncthreads();
      const double *S = St + (size_t)want * m * m;
      for (int e = threadIdx.x; e < m * m; e += kBlock) Ss[e] = S[e];
      loaded = want;
      __syncthreads();
    }
    if (c < 0) continue;
    const int32_t *nodes = cell_nodes + c * nb;
    double *u = ucell + wv * m;
    double part = 0.0;
    for (int i = lane; i < m; i += 64) {
      const double v = x[6 * (int64_t)nodes[i / 6] + i % 6];
      u[i] = v;
      part += v;
    }
    const double tot = wave_sum(part);
    const bool skip = tot == 0.0;
    const bool own = cell_S[c] == loaded;          // (else: the rare cell whose matrix is not the staged one)
    const double *Sg = St + (size_t)cell_S[c] * m * m;
    for (int i = lane; i < m; i += 64) {
      double acc = 0.0;
      if (!skip) {
        if (own)
          for (int j = 0; j < m; ++j) acc += Ss[j * m + i] * u[j];
        else
          for (int j = 0; j < m; ++j) acc += Sg[(size_t)j * m + i] * u[j];
      }
      stage[c * m + i] = acc;
    }
  }
}

// Register-resident form for m <= 64 (BCC: 48, Hybrid4: 36): lane i keeps row i of S (column i of S^T) in registers,
// the cell's vector lives one value per lane and is broadcast with v_readlane (scalar operand of the FMA): no LDS and
// no L2 traffic per cell at all.  A wave walks kDdmWaveChunk consecutive cells of the id-sorted order and reloads its
// row only when the matrix id changes.
constexpr int kDdmWaveChunk = 8;
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
template <int MT>
__global__ __launch_bounds__(kBlock) void k_ddm_cell_product_reg(int64_t C, int nb, const int32_t *__restrict__ order,
                                                                 const int32_t *__restrict__ cell_nodes,
                                                                 const int32_t *__restrict__ cell_S,
                                                                 const double *__restrict__ St,
                                                                 const double *__restrict__ x,
                                                                 double *__restrict__ stage) {
  const int m = 6 * nb;
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t q0 = w * kDdmWaveChunk;
  if (q0 >= C) return;
  double Srow[MT];
  int loaded = -1;
  const int il = lane < m ? lane : 0;
  // all gathers of the chunk first (independent loads in flight together), then the products
  int64_t cell[kDdmWaveChunk];
  double uu[kDdmWaveChunk];
#pragma unroll
  for (int k = 0; k < kDdmWaveChunk; ++k) {
    const int64_t q = q0 + k;
    cell[k] = q < C ? (int64_t)order[q] : -1;
  }
#pragma unroll
  for (int k = 0; k < kDdmWaveChunk; ++k)
    uu[k] = (cell[k] >= 0 && lane < m) ? x[6 * (int64_t)cell_nodes[cell[k] * nb + lane / 6] + lane % 6] : 0.0;
#pragma unroll
  for (int k = 0; k < kDdmWaveChunk; ++k) {
    const int64_t c = cell[k];
    if (c < 0) break;
    const int id = cell_S[c];
    if (id != loaded) {
      const double *S = St + (size_t)id * m * m;
#pragma unroll
      for (int j = 0; j < MT; ++j) Srow[j] = j < m ? S[(size_t)j * m + il] : 0.0;      // St[j][i] = S[i][j]
      loaded = id;
    }
    const double u = uu[k];
    const double tot = wave_sum(u);
    double acc = 0.0;
    if (tot != 0.0) {                       // lattice_sim.py:1239
#pragma unroll
      for (int j = 0; j < MT; ++j) acc += Srow[j] * readlane_f64(u, j);
    }
    if (lane < m) stage[c * m + lane] = acc;
  }
}

// The cell product on the fp64 matrix pipe (round 4; SURVEY.md 8(f1): "the one place a small MFMA / batched GEMV helps").
// Cells of one matrix class form tiles of 16 (host-built list, -1 = padding): Y[16 cells x m] = U[16 x m] S^T[m x m] with
// v_mfma_f64_16x16x4_f64 - per tile ceil(m / 4) x ceil(m / 16) instructions (36 for the BCC cell, m = 48).  S^T lives in
// registers as the B operands (lane (k, j): rows k + 4 kk, columns j + 16 blk; 36 doubles per lane at m = 48), loaded
// once per wave and kept while the class does not change; the A operand of a lane is its cell's gathered displacement
// (cell i = lane & 15, components (lane >> 4) + 4 kk); D comes back as [cell (lane >> 4) + 4 r][column lane & 15], so a store
// instruction writes 16 consecutive doubles of four staging rows.  The register-GEMV form (k_ddm_cell_product_reg) issued
// two v_readlane per multiply-add and reloaded a lane's row of S every eight cells: 28.7 us at 32^3 cells against 12.8 us
// here (profiles/r04_g_ddm32_*; one tile per wave: eight per wave left the chip with 256 waves and took 44.9 us).
// Operand / result layout as in pl_dense.h.  The reference's skip rule (sum of a cell's displacements == 0 -> zero
// reactions, lattice_sim.py:1239) is kept: the four lanes of a cell add their partial sums.
typedef double v4f64_ddm __attribute__((ext_vector_type(4)));
// K steps (of 4) of the matrix-pipe cell product for a cell with m dofs: one of the instantiated sizes (pl_ops.h dispatch)
inline int ddm_mfma_ks(int m) { return m <= 32 ? 8 : m <= 48 ? 12 : m <= 72 ? 18 : m <= 96 ? 24 : m <= 120 ? 30 : m <= 156 ? 39 : 48; }
#ifndef PL_DDM_TPW
#define PL_DDM_TPW 1
#endif
constexpr int kDdmTilesPerWave = PL_DDM_TPW;
// Larger cells (round 5: Hybrid1, 12 boundary nodes, m = 72; the reference's BCC + Hybrid1 hybrids, 26 nodes, m = 156): S^T no
// longer fits one wave's registers, so a tile's OUTPUT columns are cut into `slices` groups of NBW 16-column blocks and every
// slice is one wave (it holds S^T[all k][its columns]: KS x NBW doubles, gathers the tile's displacements itself - the other
// slices' gathers of the same entries hit the caches).  The K loop runs in chunks of KC instructions so that only KC gathered
// operands are live at a time; the skip rule needs the sum over ALL of a cell's displacements, which is complete only at the
// end - the product is formed regardless and dropped there.
template <int KS, int NBW, int KC = (KS < 12 ? KS : (KS % 13 == 0 ? 13 : (KS % 12 == 0 ? 12 : (KS % 6 == 0 ? 6 : KS))))>
__global__ __launch_bounds__(kBlock) void k_ddm_cell_product_mfma(int64_t n_tiles, int nb, const int32_t *__restrict__ tiles,
                                                                  const int32_t *__restrict__ tile_S,
                                                                  const int32_t *__restrict__ gidx /* [n_tiles][KS][64] */,
                                                                  const double *__restrict__ St,
                                                                  const double *__restrict__ x,
                                                                  double *__restrict__ stage, int slices = 1) {
  static_assert(KS % KC == 0, "the K loop runs in whole chunks");
  const int m = 6 * nb;
  const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
  const int64_t w = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t wt = w / slices;
  const int cb0 = (int)(w - wt * slices) * NBW;               // first 16-column block of this wave's slice
  const int64_t t0 = wt * kDdmTilesPerWave;
  if (t0 >= n_tiles) return;
  double Bs[KS][NBW];
  int loaded = -1;
  for (int q = 0; q < kDdmTilesPerWave; ++q) {
    const int64_t t = t0 + q;
    if (t >= n_tiles) break;                                   // (wave-uniform)
    const int id = tile_S[t];
    if (id != loaded) {
      const double *S = St + (size_t)id * m * m;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
          const int k = 4 * kk + kq, j = 16 * (cb0 + b) + i;
          Bs[kk][b] = (k < m && j < m) ? S[(size_t)k * m + j] : 0.0;        // St[k][j] = S[j][k]
        }
      // (tried: S through LDS once per workgroup, B operands from there: 15.0 against 12.8 us - the barrier and the
      // extra hop cost more than the L2 reads they save)
      loaded = id;
    }
    const int32_t cell = tiles[16 * t + i];
    double part = 0.0;
    v4f64_ddm acc[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) acc[b] = v4f64_ddm{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kc = 0; kc < KS; kc += KC) {
      // (the position in x of every operand entry was resolved on the host: one hop, the index loads coalesced)
      int32_t gi[KC];
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) gi[kk] = gidx[((int64_t)t * KS + kc + kk) * 64 + lane];
      double Au[KC];
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) {
        Au[kk] = gi[kk] >= 0 ? x[gi[kk]] : 0.0;
        part += Au[kk];
      }
#pragma unroll
      for (int kk = 0; kk < KC; ++kk)
#pragma unroll
        for (int b = 0; b < NBW; ++b)
          acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(Au[kk], Bs[kc + kk][b], acc[b], 0, 0, 0);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const int skip = (part == 0.0) ? 1 : 0;                     // of cell i, in all four lanes that hold it
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = kq + 4 * r;                               // D row = cell `row` of the tile
      const int32_t crow = __shfl(cell, row);
      const int srow = __shfl(skip, row);
      if (crow < 0) continue;
#pragma unroll
      for (int b = 0; b < NBW; ++b) {
        const int j = 16 * (cb0 + b) + i;
        if (j < m) stage[(int64_t)crow * m + j] = srow ? 0.0 : acc[b][r];
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_ddm_node_gather(int64_t N, const int64_t *__restrict__ node_ptr,
                                                            const int32_t *__restrict__ node_ent,
                                                            const double *__restrict__ stage,
                                                            double *__restrict__ y) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= 6 * N) return;
  const int64_t n = t / 6;
  const int k = (int)(t - 6 * n);
  double acc = 0.0;
  for (int64_t q = node_ptr[n]; q < node_ptr[n + 1]; ++q) acc += stage[6 * (int64_t)node_ent[q] + k];
  y[t] = acc;
}

// diag(sum_c B_c^T S_c B_c): the Jacobi preconditioner offered in place of the reference's SuperLU factorisation of
// the assembled Schur matrix (lattice_sim.py:1351-1415) when a preset enables the preconditioner.
__global__ __launch_bounds__(kBlock) void k_ddm_diag(int64_t C, int nb, const int32_t *__restrict__ cell_nodes,
                                                     const int32_t *__restrict__ cell_S,
                                                     const double *__restrict__ St, double *__restrict__ diag) {
  const int m = 6 * nb;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= C * m) return;
  const int64_t c = e / m;
  const int i = (int)(e - c * m);
  const double *S = St + (size_t)cell_S[c] * m * m;
  unsafeAtomicAdd(diag + 6 * (int64_t)cell_nodes[c * nb + i / 6] + i % 6, S[(size_t)i * m + i]);
}

// Node-block Jacobi (opts.precond = 3 on a DDM handle; what LatticeSim.solve_DDM asks for above PL_DDM_DENSE_MAX dofs, where
// the assembled matrix is not factorised): the 6 x 6 diagonal blocks of G = sum_c B_c^T Shat_c B_c, inverted per node.
// Constrained dofs are taken out of the block before the inversion and get zero rows / columns in the inverse.  Measured on the
// host (BCC cantilevers, r = 0.05, 1e-8): 20^3 cells 296 iterations against 384 with the diagonal alone, 12^3 175 against 254.
__global__ __launch_bounds__(kBlock) void k_ddm_node_blocks(int64_t C, int nb, const int32_t *__restrict__ cell_nodes,
                                                            const int32_t *__restrict__ cell_S,
                                                            const double *__restrict__ St, double *__restrict__ B) {
  const int m = 6 * nb;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= C * nb * 36) return;
  const int64_t c = e / (nb * 36);
  const int q = (int)(e - c * nb * 36), a = q / 36, ij = q - 36 * a, i = ij / 6, j = ij - 6 * i;
  const double *S = St + (size_t)cell_S[c] * m * m;        // (stored transposed; the diagonal blocks are symmetric)
  unsafeAtomicAdd(B + 36 * (int64_t)cell_nodes[c * nb + a] + ij, S[(size_t)(6 * a + i) * m + 6 * a + j]);
}
__global__ __launch_bounds__(kBlock) void k_ddm_node_blocks_invert(int64_t N, const uint8_t *__restrict__ fixed /* may be null */,
                                                                   double *__restrict__ B) {
  const int64_t n = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (n >= N) return;
  double A[36];
  bool fx[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) fx[k] = fixed && fixed[6 * n + k];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double v = 0.5 * (B[36 * n + 6 * i + j] + B[36 * n + 6 * j + i]);
      A[6 * i + j] = (fx[i] || fx[j]) ? 0.0 : v;           // (spd6_inverse drops modes without stiffness)
    }
  spd6_inverse(A);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) B[36 * n + 6 * i + j] = (fx[i] || fx[j]) ? 0.0 : A[6 * i + j];
}
// z = B^-1 r node by node, dot_out[slot] += r.z
__global__ __launch_bounds__(kBlock) void k_ddm_node_blocks_apply(int64_t N, const double *__restrict__ B,
                                                                  const double *__restrict__ r, double *__restrict__ z,
                                                                  double *__restrict__ dot_out) {
  __shared__ double red[kBlock / kWave];
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;     // one lane per row of a node block
  double acc = 0.0;
  if (t < 6 * N) {
    const int64_t n = t / 6;
    const double *row = B + 6 * t;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) v += row[j] * r[6 * n + j];
    z[t] = v;
    acc = v * r[t];
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0 && dot_out) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
}

// The reference's CG preconditioner for this operator (LatticeSim.build_preconditioner, lattice_sim.py:1351-1415, with
// Cell.build_local_preconditioner, cell.py:783-827): G = sum_c B_c^T Shat_c B_c on the free dofs, factorised once.
// Here G is assembled DENSE (ld = n padded to the block size of pl_dense.h) and handed to the device Cholesky; rows
// and columns of constrained dofs and of the padding are left out and get a unit diagonal from k_ddm_dense_unit.
__global__ __launch_bounds__(kBlock) void k_ddm_dense_assemble(int64_t C, int nb, const int32_t *__restrict__ cell_nodes,
                                                               const int32_t *__restrict__ cell_S,
                                                               const double *__restrict__ St,
                                                               const uint8_t *__restrict__ fixed, int ld,
                                                               double *__restrict__ G) {
  const int m = 6 * nb;
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= C * m * m) return;
  const int64_t c = e / ((int64_t)m * m);
  const int ij = (int)(e - c * m * m), j = ij / m, i = ij - j * m;       // consecutive lanes: consecutive i of St[j][i]
  const int64_t gi = 6 * (int64_t)cell_nodes[c * nb + i / 6] + i % 6;
  const int64_t gj = 6 * (int64_t)cell_nodes[c * nb + j / 6] + j % 6;
  if (fixed && (fixed[gi] || fixed[gj])) return;
  unsafeAtomicAdd(G + gi * ld + gj, St[(size_t)cell_S[c] * m * m + (size_t)j * m + i]);
}
// BSR(6 x 6) -> dense (precond = 5 of a strut-operator handle): G[6 i + a][6 j + b] = vals[block (i, j)][a][b]
__global__ __launch_bounds__(kBlock) void k_bsr_to_dense(int64_t N, int64_t nblk, const int64_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col, const double *__restrict__ vals,
                                                         int ld, double *__restrict__ G) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= nblk * 36) return;
  const int64_t blk = e / 36;
  const int ab = (int)(e - 36 * blk), a = ab / 6, b = ab - 6 * a;
  int64_t lo = 0, hi = N;                        // row of the block: the last i with rowptr[i] <= blk
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= blk) lo = mid;
    else hi = mid;
  }
  G[(6 * lo + a) * (int64_t)ld + 6 * (int64_t)col[blk] + b] = vals[e];
}
__global__ void k_ddm_dense_unit(int64_t n6, int ld, const uint8_t *__restrict__ fixed, double *__restrict__ G) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ld) return;
  if (d >= n6 || (fixed && fixed[d]) || G[d * ld + d] == 0.0) G[d * ld + d] = 1.0;
}

}  // namespace pl
