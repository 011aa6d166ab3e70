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
  double tot = part;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
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
  double tot = part;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
  const bool skip = tot == 0.0;            // lattice_sim.py:1239: np.sum(displacement_cell) == 0 -> zero reactions
  const double *S = St + (size_t)cell_S[c] * m * m;
  for (int i = lane; i < m; i += 64) {
    double acc = 0.0;
    if (!skip)
      for (int j = 0; j < m; ++j) acc += S[(size_t)j * m + i] * ucell[wv][j];
    stage[c * m + i] = acc;
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
__global__ void k_ddm_dense_unit(int64_t n6, int ld, const uint8_t *__restrict__ fixed, double *__restrict__ G) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ld) return;
  if (d >= n6 || (fixed && fixed[d]) || G[d * ld + d] == 0.0) G[d * ld + d] = 1.0;
}

}  // namespace pl
