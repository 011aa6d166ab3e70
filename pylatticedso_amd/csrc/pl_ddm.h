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

}  // namespace pl
