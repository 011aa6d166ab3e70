// Joint-penalisation lengths on the device: for every strut end, over the OTHER struts meeting at that node, the
// largest  L = r_other / tan(angle / 2)  - what the reference computes in a Python double loop over the node valence
// (Lattice.define_angles_between_beams, lattice.py:871-904; Beam.get_angle_between_beams, beam.py:204-277;
// function_penalization_Lzone, utils.py:432-453: 1e-7 when the angle exceeds 170 degrees, pairs at angle 0 skipped).
// One thread per half-edge (strut end), node -> half-edge lists in CSR form.  Same operation order as the reference's
// arithmetic: cos = ((x1 x2 + y1 y2) + z1 z2) / (|u| |v|) clamped to [-1, 1], angle in degrees, back to radians.
#pragma once
#include <hip/hip_runtime.h>

#include "pl_kernels.h"

namespace pl {

__global__ __launch_bounds__(kBlock) void k_lzone(int64_t n_half, const double *__restrict__ xyz,
                                                  const int32_t *__restrict__ conn, const double *__restrict__ radius,
                                                  const int64_t *__restrict__ node_ptr,
                                                  const int32_t *__restrict__ node_half, double *__restrict__ lzone) {
  const int64_t h = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (h >= n_half) return;
  const int32_t node = conn[h], far = conn[h ^ 1];
  const double ux = xyz[3 * (int64_t)far] - xyz[3 * (int64_t)node];
  const double uy = xyz[3 * (int64_t)far + 1] - xyz[3 * (int64_t)node + 1];
  const double uz = xyz[3 * (int64_t)far + 2] - xyz[3 * (int64_t)node + 2];
  const double un = sqrt(ux * ux + uy * uy + uz * uz);
  constexpr double kDeg = 57.29577951308232, kRad = 0.017453292519943295;
  double best = -1.0;
  for (int64_t q = node_ptr[node]; q < node_ptr[node + 1]; ++q) {
    const int32_t g = node_half[q];
    if (g == h) continue;
    const int32_t far2 = conn[g ^ 1];
    const double vx = xyz[3 * (int64_t)far2] - xyz[3 * (int64_t)node];
    const double vy = xyz[3 * (int64_t)far2 + 1] - xyz[3 * (int64_t)node + 1];
    const double vz = xyz[3 * (int64_t)far2 + 2] - xyz[3 * (int64_t)node + 2];
    const double vn = sqrt(vx * vx + vy * vy + vz * vz);
    double c = ((ux * vx + uy * vy) + uz * vz) / (un * vn);
    c = fmin(1.0, fmax(-1.0, c));
    const double ang = acos(c) * kDeg;
    if (!(ang > 1e-12)) continue;
    const double L = ang > 170.0 ? 0.0000001 : radius[g >> 1] / tan(ang * kRad / 2.0);
    best = fmax(best, L);
  }
  lzone[h] = best < 0.0 ? 0.0 : best;
}

}  // namespace pl
