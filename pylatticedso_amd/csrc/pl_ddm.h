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

// (fixed != null: rows of constrained dofs come out as zero; dot_out != null: dot_out[slot] += x . y - what k_mask_dot did
// in a launch of its own until round 5: 4.8 us of the 54-us iteration at 32^3 cells)
__global__ __launch_bounds__(kBlock) void k_ddm_node_gather(int64_t N, const int64_t *__restrict__ node_ptr,
                                                            const int32_t *__restrict__ node_ent,
                                                            const double *__restrict__ stage,
                                                            double *__restrict__ y,
                                                            const uint8_t *__restrict__ fixed = nullptr,
                                                            const double *__restrict__ x = nullptr,
                                                            double *__restrict__ dot_out = nullptr) {
  __shared__ double red[kBlock / kWave];
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  double part = 0.0;
  if (t < 6 * N) {
    const int64_t n = t / 6;
    const int k = (int)(t - 6 * n);
    double acc = 0.0;
    for (int64_t q = node_ptr[n]; q < node_ptr[n + 1]; ++q) acc += stage[6 * (int64_t)node_ent[q] + k];
    if (fixed && fixed[t]) acc = 0.0;
    y[t] = acc;
    if (dot_out) part = x[t] * acc;
  }
  if (dot_out) {                       // (uniform over the launch)
    const double s = block_sum(part, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
  }
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
// St[s][j][i] = S[s][i][j] for a whole palette (pl_ddm_update_matrices uploads the caller's matrices as they are and
// transposes here: a host transposition needs the caller's threads, which a numpy BLAS call may have left spinning)
__global__ __launch_bounds__(kBlock) void k_ddm_transpose_palette(int64_t n_S, int m, const double *__restrict__ S,
                                                                  double *__restrict__ St) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= n_S * m * m) return;
  const int64_t s = e / ((int64_t)m * m);
  const int r = (int)(e - s * m * m), j = r / m, i = r - j * m;       // writes coalesced
  St[e] = S[(size_t)s * m * m + (size_t)i * m + j];
}

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

// ---------------------------------------------------------------------------------------------------------------
// Two-level preconditioner of the DDM operator (opts.precond = 4 on a DDM handle; round 5):
//     M^-1 = B^-1 + Z A_c^-1 Z^T ,   A_c = Z^T P G P Z ,   G = sum_c B_c^T S_c B_c
// B = the 6 x 6 node blocks of G (precond = 3), Z = 12 modes per aggregate of nodes (six rigid-body motions about the
// aggregate's reference point + the six uniform strains u = eps (x - c), no rotation), aggregates = boxes of a regular grid
// over the nodes' bounding box (pl_ddm_set_geometry).  What it replaces: the reference factorises the assembled G with
// SuperLU (LatticeSim.build_preconditioner, lattice_sim.py:1351-1415), which a dense device factorisation follows up to
// PL_DDM_DENSE_MAX dofs only.  Host prototype: tools/experiments/ddm_two_level_host.py (24^3 BCC cells: 356 iterations with
// the node blocks, 141 / 169 / 207 with aggregates of 4 / 5 / 6 cells per axis; 32^3: 474 -> 213 with 125 aggregates).
// ---------------------------------------------------------------------------------------------------------------
// dof d (0-2 translations, 3-5 rotations) of a node at `rel` from the reference point under mode q
__device__ __forceinline__ double ddm_mode(int q, int d, double rx, double ry, double rz) {
  if (q < 3) return d == q ? 1.0 : 0.0;
  if (q < 6) {
    if (d >= 3) return d == q ? 1.0 : 0.0;
    const int a = q - 3;                               // u = e_a x r
    if (a == 0) return d == 1 ? -rz : (d == 2 ? ry : 0.0);
    if (a == 1) return d == 0 ? rz : (d == 2 ? -rx : 0.0);
    return d == 0 ? -ry : (d == 1 ? rx : 0.0);
  }
  if (d >= 3) return 0.0;
  switch (q) {
    case 6: return d == 0 ? rx : 0.0;
    case 7: return d == 1 ? ry : 0.0;
    case 8: return d == 2 ? rz : 0.0;
    case 9: return d == 0 ? 0.5 * ry : (d == 1 ? 0.5 * rx : 0.0);
    case 10: return d == 1 ? 0.5 * rz : (d == 2 ? 0.5 * ry : 0.0);
    default: return d == 2 ? 0.5 * rx : (d == 0 ? 0.5 * rz : 0.0);
  }
}
constexpr int kDdmModes = 12;
// A_c += Z_c^T S_c Z_c cell by cell: ONE wave per cell.  The cell's nodes fall into k <= 8 aggregates ("slots"); for every
// slot s the wave forms T_s = S_c Z_s (m x 12, Z_s = the masked mode rows of the nodes in s, zero elsewhere) in LDS and then,
// for every slot t whose aggregate number is not below s's, the 12 x 12 block Z_t^T T_s, added to A_c[agg_t][agg_s] - the
// block LOWER triangle, which is what the factorisation reads; diagonal blocks are symmetrised (a surrogate S need not be
// symmetric to the last bit).  Rows of fixed dofs are zero in Z: A_c is the Galerkin operator of P G P.
__global__ __launch_bounds__(kWave) void k_ddm_coarse_cells(int64_t C, int nb, const int32_t *__restrict__ cell_nodes,
                                                            const int32_t *__restrict__ cell_S,
                                                            const double *__restrict__ St,
                                                            const int32_t *__restrict__ agg_of_node,
                                                            const double *__restrict__ cen,
                                                            const double *__restrict__ xyz,
                                                            const uint8_t *__restrict__ fixed /* [6N], may be null */,
                                                            int ld, double *__restrict__ Ac) {
  extern __shared__ double lds[];
  const int m = 6 * nb;
  double *Z = lds;                      // [m][12]
  double *T = lds + m * kDdmModes;      // [m][12]
  __shared__ int s_slot[32], s_agg[8], s_k;
  const int64_t c = blockIdx.x;
  if (c >= C) return;
  const int lane = threadIdx.x;
  const int32_t *nd = cell_nodes + c * nb;
  if (lane == 0) {
    int k = 0;
    for (int i = 0; i < nb; ++i) {
      const int a = agg_of_node[nd[i]];
      int sl = -1;
      for (int q = 0; q < k; ++q)
        if (s_agg[q] == a) sl = q;
      if (sl < 0 && k < 8) {
        s_agg[k] = a;
        sl = k++;
      }
      s_slot[i] = sl;                   // (-1: a ninth aggregate - cannot happen while aggregates are boxes at least a cell wide;
    }                                   //  such a node is left out of the coarse operator, which stays SPD)
    s_k = k;
  }
  for (int e = lane; e < m * kDdmModes; e += kWave) {
    const int row = e / kDdmModes, q = e - row * kDdmModes, i = row / 6, d = row - 6 * i;
    const int64_t n = nd[i];
    const int a = agg_of_node[n];
    double v = ddm_mode(q, d, xyz[3 * n] - cen[3 * a], xyz[3 * n + 1] - cen[3 * a + 1], xyz[3 * n + 2] - cen[3 * a + 2]);
    if (fixed && fixed[6 * n + d]) v = 0.0;
    Z[e] = v;
  }
  __syncthreads();
  const int k = s_k;
  const double *S = St + (size_t)cell_S[c] * m * m;              // St[col][row]
  for (int s = 0; s < k; ++s) {
    for (int row = lane; row < m; row += kWave) {                // T_s[row][0..11]: a lane owns a row, so every entry of S is
      double acc[kDdmModes];                                     // read ONCE per slot (coalesced over the rows) and meets the
#pragma unroll                                                   // twelve mode values of its column, which all lanes read at
      for (int q = 0; q < kDdmModes; ++q) acc[q] = 0.0;          // the same LDS address (first version: one lane per (row, mode),
      for (int j = 0; j < nb; ++j) {                             // S read twelve times - 700 us at 32^3 cells)
        if (s_slot[j] != s) continue;
        for (int d = 0; d < 6; ++d) {
          const double sv = S[(size_t)(6 * j + d) * m + row];
          const double *zr = Z + (6 * j + d) * kDdmModes;
#pragma unroll
          for (int q = 0; q < kDdmModes; ++q) acc[q] += sv * zr[q];
        }
      }
#pragma unroll
      for (int q = 0; q < kDdmModes; ++q) T[row * kDdmModes + q] = acc[q];
    }
    __syncthreads();
    for (int t = 0; t < k; ++t) {
      if (s_agg[t] < s_agg[s]) continue;
      for (int e = lane; e < kDdmModes * kDdmModes; e += kWave) {
        const int p = e / kDdmModes, q = e - p * kDdmModes;
        double v = 0.0, w = 0.0;
        for (int i = 0; i < nb; ++i) {
          if (s_slot[i] != t) continue;
#pragma unroll
          for (int d = 0; d < 6; ++d) {
            v += Z[(6 * i + d) * kDdmModes + p] * T[(6 * i + d) * kDdmModes + q];
            if (t == s) w += Z[(6 * i + d) * kDdmModes + q] * T[(6 * i + d) * kDdmModes + p];
          }
        }
        if (t == s) v = 0.5 * (v + w);
        if (v != 0.0)
          unsafeAtomicAdd(Ac + (size_t)(kDdmModes * s_agg[t] + p) * ld + kDdmModes * s_agg[s] + q, v);
      }
    }
    __syncthreads();
  }
}
// r_c = Z^T r: one workgroup per aggregate over its node list (no atomics; r is zero on fixed dofs - the mask is applied
// all the same, the modes of A_c are the masked ones)
__global__ __launch_bounds__(kBlock) void k_ddm_restrict(const int32_t *__restrict__ agg_ptr,
                                                         const int32_t *__restrict__ agg_nodes,
                                                         const double *__restrict__ cen, const double *__restrict__ xyz,
                                                         const uint8_t *__restrict__ fixed /* may be null */,
                                                         const double *__restrict__ r, double *__restrict__ rc) {
  __shared__ double red[kDdmModes][kBlock / kWave];
  const int a = blockIdx.x;
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double acc[kDdmModes];
#pragma unroll
  for (int q = 0; q < kDdmModes; ++q) acc[q] = 0.0;
  for (int e = agg_ptr[a] + (int)threadIdx.x; e < agg_ptr[a + 1]; e += kBlock) {
    const int64_t n = agg_nodes[e];
    double v[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) v[d] = (fixed && fixed[6 * n + d]) ? 0.0 : r[6 * n + d];
    const double rx = xyz[3 * n] - c0, ry = xyz[3 * n + 1] - c1, rz = xyz[3 * n + 2] - c2;
    acc[0] += v[0];
    acc[1] += v[1];
    acc[2] += v[2];
    acc[3] += v[3] + (ry * v[2] - rz * v[1]);
    acc[4] += v[4] + (rz * v[0] - rx * v[2]);
    acc[5] += v[5] + (rx * v[1] - ry * v[0]);
    acc[6] += rx * v[0];
    acc[7] += ry * v[1];
    acc[8] += rz * v[2];
    acc[9] += 0.5 * (ry * v[0] + rx * v[1]);
    acc[10] += 0.5 * (rz * v[1] + ry * v[2]);
    acc[11] += 0.5 * (rz * v[0] + rx * v[2]);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < kDdmModes; ++q) {
    const double s = wave_sum(acc[q]);
    if (lane == 0) red[q][wv] = s;
  }
  __syncthreads();
  if (threadIdx.x < kDdmModes) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += red[threadIdx.x][w];
    rc[kDdmModes * a + threadIdx.x] = s;
  }
}
// The CG update of an iteration and the restriction of the NEW residual in one pass over the nodes, aggregate by aggregate:
// x += alpha p, r -= alpha K p (alpha from the scalar set, clamped as k_pcg_update does), r.r (and, REF, x.x and the norm of
// the vector the next direction is built on) into the scalar set, r_c += Z^T r.  kDdmSplit workgroups share an aggregate (a
// launch of one workgroup per aggregate leaves half the chip idle), so r_c is summed atomically: the caller's previous
// k_ddm_two_level_apply has cleared it.  What it replaces: k_pcg_update + k_ddm_restrict, 6.2 + 7.0 us at 32^3 cells.
constexpr int kDdmSplit = 4;
template <bool REF>
__global__ __launch_bounds__(kBlock) void k_ddm_update_restrict(const int32_t *__restrict__ agg_ptr,
                                                                const int32_t *__restrict__ agg_nodes,
                                                                const double *__restrict__ cen,
                                                                const double *__restrict__ xyz,
                                                                const uint8_t *__restrict__ fixed /* may be null */,
                                                                const double *__restrict__ p, const double *__restrict__ Ap,
                                                                double *__restrict__ x, double *__restrict__ r,
                                                                double *__restrict__ scal, double alpha_max,
                                                                const double *__restrict__ pn /* REF: may be null */,
                                                                const int *__restrict__ stop /* may be null */,
                                                                double *__restrict__ rc) {
  __shared__ double red[kDdmModes + 3][kBlock / kWave];
  if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // the iterate is final
  const double pap = scalar_read(scal, S_PAP);
  double alpha = (pap != 0.0) ? scalar_read(scal, S_RZ_OLD) / pap : 0.0;
  if (alpha_max > 0.0 && alpha > alpha_max) alpha = alpha_max;
  if (REF && blockIdx.x == 0 && threadIdx.x == 0) scal[S_ALPHA * kSlots] = alpha;
  const int a = blockIdx.x / kDdmSplit, part = blockIdx.x - a * kDdmSplit;
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double acc[kDdmModes];
#pragma unroll
  for (int q = 0; q < kDdmModes; ++q) acc[q] = 0.0;
  double rr = 0.0, xx = 0.0, pp = 0.0;
  // one lane per PAIR of a node's dofs (three lanes per node: 48 contiguous bytes of every vector), each adding its share of
  // the twelve restriction sums (one lane per node left three quarters of a workgroup idle: 16.7 us against 9 us)
  const int e0 = agg_ptr[a], cnt3 = 3 * (agg_ptr[a + 1] - e0);
  for (int j = part * kBlock + (int)threadIdx.x; j < cnt3; j += kDdmSplit * kBlock) {
    const int e = j / 3, h = j - 3 * e;
    const int64_t n = agg_nodes[e0 + e], i2 = 3 * n + h;
    const double2 pv = reinterpret_cast<const double2 *>(p)[i2], av = reinterpret_cast<const double2 *>(Ap)[i2];
    double2 xv = reinterpret_cast<double2 *>(x)[i2], rv = reinterpret_cast<double2 *>(r)[i2];
    xv.x += alpha * pv.x;
    xv.y += alpha * pv.y;
    rv.x -= alpha * av.x;
    rv.y -= alpha * av.y;
    reinterpret_cast<double2 *>(x)[i2] = xv;
    reinterpret_cast<double2 *>(r)[i2] = rv;
    rr += rv.x * rv.x + rv.y * rv.y;
    if (REF) {
      xx += xv.x * xv.x + xv.y * xv.y;
      if (pn) {
        const double2 q = reinterpret_cast<const double2 *>(pn)[i2];
        pp += q.x * q.x + q.y * q.y;
      }
    }
    const double v0 = (fixed && fixed[6 * n + 2 * h]) ? 0.0 : rv.x, v1 = (fixed && fixed[6 * n + 2 * h + 1]) ? 0.0 : rv.y;
    const double rx = xyz[3 * n] - c0, ry = xyz[3 * n + 1] - c1, rz = xyz[3 * n + 2] - c2;
    if (h == 0) {            // (u_x, u_y)
      acc[0] += v0;
      acc[1] += v1;
      acc[3] -= rz * v1;
      acc[4] += rz * v0;
      acc[5] += rx * v1 - ry * v0;
      acc[6] += rx * v0;
      acc[7] += ry * v1;
      acc[9] += 0.5 * (ry * v0 + rx * v1);
      acc[10] += 0.5 * rz * v1;
      acc[11] += 0.5 * rz * v0;
    } else if (h == 1) {     // (u_z, theta_x)
      acc[2] += v0;
      acc[3] += v1 + ry * v0;
      acc[4] -= rx * v0;
      acc[8] += rz * v0;
      acc[10] += 0.5 * ry * v0;
      acc[11] += 0.5 * rx * v0;
    } else {                 // (theta_y, theta_z)
      acc[4] += v0;
      acc[5] += v1;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < kDdmModes; ++q) {
    const double s = wave_sum(acc[q]);
    if (lane == 0) red[q][wv] = s;
  }
  {
    const double s0 = wave_sum(rr), s1 = wave_sum(xx), s2 = wave_sum(pp);
    if (lane == 0) {
      red[kDdmModes][wv] = s0;
      red[kDdmModes + 1][wv] = s1;
      red[kDdmModes + 2][wv] = s2;
    }
  }
  __syncthreads();
  if (threadIdx.x < kDdmModes + 3) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) s += red[threadIdx.x][w];
    const int q = threadIdx.x;
    if (q < kDdmModes) {
      if (s != 0.0) unsafeAtomicAdd(rc + kDdmModes * a + q, s);
    } else if (q == kDdmModes) {
      scalar_add(scal, S_RR, s);
      if (REF && !pn) scalar_add(scal, S_PP, s);      // (no separate source vector: ||r_new||, as k_pcg_update)
    } else if (REF && q == kDdmModes + 1) {
      scalar_add(scal, S_XX, s);
    } else if (REF && pn) {
      scalar_add(scal, S_PP, s);
    }
  }
}
// z = B^-1 r + P Z y_c node by node; dot_out[slot] += r . B^-1 r (the dense level's share r_c . y_c comes from its GEMV)
__global__ __launch_bounds__(kBlock) void k_ddm_two_level_apply(int64_t N, const double *__restrict__ B,
                                                                const int32_t *__restrict__ agg_of_node,
                                                                const double *__restrict__ cen,
                                                                const double *__restrict__ xyz,
                                                                const uint8_t *__restrict__ fixed /* may be null */,
                                                                const double *__restrict__ yc,
                                                                const double *__restrict__ r, double *__restrict__ z,
                                                                double *__restrict__ dot_out,
                                                                double *__restrict__ rc_clear = nullptr, int n_rc = 0) {
  __shared__ double red[kBlock / kWave];
  // (the dense solve has consumed r_c: cleared here for the atomic sums of the next k_ddm_update_restrict)
  if (rc_clear && (blockIdx.x == 1 || gridDim.x == 1))
    for (int e = threadIdx.x; e < n_rc; e += kBlock) rc_clear[e] = 0.0;
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;     // one lane per row of a node block
  double acc = 0.0;
  if (t < 6 * N) {
    const int64_t n = t / 6;
    const int d = (int)(t - 6 * n);
    const double *row = B + 6 * t;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 6; ++j) v += row[j] * r[6 * n + j];
    acc = v * r[t];
    if (!(fixed && fixed[t])) {
      const int a = agg_of_node[n];
      const double *y = yc + kDdmModes * a;
      const double rx = xyz[3 * n] - cen[3 * a], ry = xyz[3 * n + 1] - cen[3 * a + 1], rz = xyz[3 * n + 2] - cen[3 * a + 2];
      double zc;
      if (d >= 3) {
        zc = y[d];
      } else {
        const double U[3] = {y[0] + (y[4] * rz - y[5] * ry), y[1] + (y[5] * rx - y[3] * rz), y[2] + (y[3] * ry - y[4] * rx)};
        const double E[3] = {y[6] * rx + 0.5 * (y[9] * ry + y[11] * rz), y[7] * ry + 0.5 * (y[9] * rx + y[10] * rz),
                             y[8] * rz + 0.5 * (y[10] * ry + y[11] * rx)};
        zc = U[d] + E[d];
      }
      v += zc;
    }
    z[t] = v;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0 && dot_out) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
}

}  // namespace pl
