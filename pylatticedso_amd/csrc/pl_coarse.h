// Multi-level additive preconditioner of libpylattice_hip (SPD):
//     M^-1 = D^-1 + sum_t Z_t (Z_t^T K Z_t)^-1 Z_t^T + Z (Z^T K Z)^-1 Z^T
// (Jacobi level, tile level: the 6 rigid-body modes of every K*p tile with a 6 x 6 block solve, dense level).
//
// D = diag(K) (Jacobi) is the fine level.  The coarse space Z holds the 6 rigid-body modes (3 translations, 3 rotations
// about the aggregate centroid) of every AGGREGATE = a g x g x g group of the node bricks the K*p tiles are made of,
// restricted to the aggregate's nodes (plain aggregation, every node in exactly one aggregate) and to the free dofs.
// Rigid-body modes of a strut carry no energy, so only struts that cross aggregates (or touch Dirichlet dofs)
// contribute to A_c = Z^T P K P Z, which is small (6 n_agg <= ~3000) and kept as a DENSE inverse on the device
// (blocked Cholesky + explicit inverse factor, pl_dense.h); applying it is two triangular GEMVs.
// Effect (CPU experiment + GPU bench): PCG iterations of the n^3 Octet cantilever stop growing with n
// (263 / 397 / 812 with Jacobi at n = 16 / 24 / 50 -> ~100-180).
//
// Per iteration (all device-side, no host round trip):
//   K*p (+ p.Ap)                                                            k_spmv_tile
//   r -= a Ap, r_c += Z^T r (tile partial sums, atomics), r.r, r.D^-1 r,
//   y_t = B_t^-1 Z_t^T r and r.D^-1 r += r_t.y_t (tile level)                k_pcg_update_tile
//   y_c = A_c^-1 r_c, r.z = r.D^-1 r + r_c.y_c                              k_tri_gemv, k_tri_gemv_t
//   x += a p; p = D^-1 r + P Z (y_c + y_t) + beta p  (z is never stored)    k_pcg_direction_coarse
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "pl_dense.h"
#include "pl_kernels.h"
#include "pl_tile.h"

namespace pl {


struct Coarse {
  bool enabled = false;     // topology prepared (pl_create)
  bool ready = false;       // A_c^-1 valid (pl_assemble)
  int n_agg = 0, nc = 0, ncp = 0;   // ncp = nc rounded up to the dense block size (padding rows are identity)
  // modes per aggregate: 6 rigid-body modes, or 12 = rigid + the six uniform strains (as the tile level's; needs
  // tile_modes = 12, whose per-tile sums it shares).  Host experiment (tools/experiments/multilevel_aggregates.py, 24^3
  // Octet): 5^3 aggregates x 12 modes (1500 dofs) need 11 % fewer iterations than 7^3 x 6 (2058 dofs).
  int cm = 6;
  int bw_blocks = 0;                // block bandwidth of A_c (aggregates couple to their 26 neighbours only)
  int64_t n_tiles = 0;
  int vblock = kBlock;     // threads per workgroup of the per-tile vector kernels: the longest tile rounded up to whole waves
  TBuf<int32_t> agg_of_node, agg_of_tile;
  TBuf<int32_t> tile_of_node;   // K*p tile of every node (k_pcg_direction_flat)
  TBuf<double> cen;
  TBuf<int32_t> cross_idx;      // struts whose ends lie in different aggregates, sorted by (agg(a), agg(b))
  int64_t n_cross = 0;
  float *Ainv = nullptr;               // explicit A_c^-1 = W^T W in fp32 (short form of the iteration on small lattices, pl_small.h)
  bool ainv_ready = false;             // ... valid for the current factorisation
  float *W = nullptr, *Wt = nullptr;   // inverse Cholesky factor and its transpose, fp32 storage (pl_dense.h)
  bool w16 = false;                    // ... or bfloat16 in the same buffers (large levels: the GEMVs are bandwidth-bound)
  double *Ac = nullptr, *Lf = nullptr, *Dinv = nullptr, *rc = nullptr, *yc = nullptr, *tv = nullptr;
  int *info = nullptr;
  unsigned *bar = nullptr;   // grid-barrier counter of the persistent Cholesky chain (pl_dense.h)
  bool ac_clean = false;   // Ac was zeroed after the previous factorisation (off the critical path of pl_assemble)
  // tile level: every K*p tile is an aggregate of its own between the Jacobi level and the dense level
  bool tile_level = true;
  std::vector<int32_t> h_tile_start;
  double *Bt_inv = nullptr, *yt = nullptr;   // [n_tiles * tm * tm], [n_tiles * tm], tm = tile_modes
  // Modes of the tile level: 6 (rigid body) or 12 (+ the six uniform strains u = eps (x - c), no rotation).  A piecewise
  // rigid field follows a macroscopic strain only with jumps across tile faces; the strain modes let a tile deform with
  // it.  Measured: 134 -> 126 iterations at 50^3 Octet, 396 -> 369 at 100^3 BCC (rtol 1e-6), DESIGN.md section 7.
  int tile_modes = 6;
  double *Bt_raw = nullptr;                  // [n_tiles * 144] B_t before the inversion (12 modes only)
  double *Bt_rawA = nullptr;                 // several GPUs, 12-mode dense level: the tile blocks on ALL nodes (for A_c)
  // fp32 copy of D^-1 [6N] read by the two per-iteration vector kernels (a preconditioner only has to be the SAME
  // symmetric operator in every iteration, so rounding the Jacobi weights is free).  The node positions stay fp64:
  // the coarse modes must be EXACTLY rigid per aggregate - their energy is tiny next to ||K||, and a 1e-7 error in
  // the lever arms costs iterations (measured)
  float *dinv32 = nullptr;
  // ... but where every lever arm (node position minus its aggregate's reference point) happens to be a float EXACTLY - unit
  // cells of size 1, 1/2, 1/4 ...: every BASELINE configuration - the two vector kernels read it as 12 bytes per node instead
  // of 24 (+ the reference point): the same numbers, bit for bit (rel_exact; checked node by node in coarse_setup)
  TBuf<float> rel32;
  bool rel_exact = false;
  // struts inside one aggregate that touch a Dirichlet dof (the only non-crossing struts with coarse energy);
  // rebuilt on the device after every pl_set_bc
  TBuf<int32_t> fix_list;
  int *fix_count = nullptr;
  int64_t n_fix = -1;                        // -1: stale
  ~Coarse() {
    for (void *q : {(void *)Ac, (void *)Lf, (void *)W, (void *)Wt, (void *)Dinv, (void *)rc, (void *)yc, (void *)tv,
                    (void *)info, (void *)Bt_inv, (void *)yt, (void *)fix_count, (void *)dinv32, (void *)bar,
                    (void *)Bt_raw, (void *)Bt_rawA, (void *)Ainv})
      if (q) (void)hipFree(q);
  }
};

// Host: aggregates = groups of g^3 bricks of the (global) brick grid; ids are dense over the whole aggregate grid so
// that all ranks of a multi-GPU run agree (aggregates without nodes give identity rows).  The reference point of an
// aggregate's rigid-body modes is the geometric centre of its cell (any point spans the same space).
// local = true (the rank-local level of a multi-GPU run, see pl_api.hip): only the aggregates that hold tiles of THIS
// handle are numbered, and g is chosen from their count - the level is never summed over ranks.
inline int coarse_setup(Coarse &c, const std::vector<int32_t> &tile_start, const std::vector<int64_t> &tile_brick,
                        const BrickGrid &grid, const double *xyz_dev_order, int64_t N, int max_dofs,
                        const std::vector<int32_t> &conn, bool local = false, bool multi_rank = false, int modes = 6) {
  const int64_t T = (int64_t)tile_start.size() - 1;
  c.cm = modes;
  const int64_t *nbrick = grid.nb;
  // aggregate index of a brick along axis k: floor(b * na_k / nb_k) - groups of bricks whose sizes differ by at most one,
  // so that the aggregate grid can use the dofs it is allowed (13 bricks per axis -> 8 aggregates, not ceil(13/2) = 7)
  // Aggregates are numbered with the LONGEST axis of the brick grid slowest (ax[0]): the coarse operator couples
  // neighbouring aggregates only, so its band is ~ the product of the two SHORT extents - weak-scaling slabs grow along
  // one axis, and the band (fill of the Cholesky factor, cost of the inverse factor) then stays what it is on one GPU.
  int ax[3] = {0, 1, 2};
  std::stable_sort(ax, ax + 3, [&](int l, int r) { return nbrick[l] > nbrick[r]; });
  auto grid_agg = [&](int64_t key, int /*g*/, const int64_t *na) {
    const int64_t b[3] = {key / (nbrick[2] * nbrick[1]), (key / nbrick[2]) % nbrick[1], key % nbrick[2]};
    const int64_t a0 = b[ax[0]] * na[ax[0]] / nbrick[ax[0]], a1 = b[ax[1]] * na[ax[1]] / nbrick[ax[1]],
                  a2 = b[ax[2]] * na[ax[2]] / nbrick[ax[2]];
    return (a0 * na[ax[1]] + a1) * na[ax[2]] + a2;
  };
  int g = 2;
  int64_t na[3];
  std::vector<int64_t> used;                       // local: sorted grid ids of the aggregates that hold tiles
  for (int step = 0;; ++step) {
    const double scale = 1.5 + 0.125 * step;       // bricks per aggregate and axis, at least 1.5
    for (int k = 0; k < 3; ++k) na[k] = std::max<int64_t>(1, (int64_t)std::floor((double)nbrick[k] / scale));
    if (!local && step == 0 && grid.na[0] > 0 && modes * grid.na[0] * grid.na[1] * grid.na[2] <= max_dofs)
      for (int k = 0; k < 3; ++k) na[k] = grid.na[k];   // the aggregate grid the bricks were cut to fit (spatial_order)
    int64_t count = na[0] * na[1] * na[2];
    if (local) {
      used.clear();
      for (int64_t t = 0; t < T; ++t) used.push_back(grid_agg(tile_brick[t], g, na));
      std::sort(used.begin(), used.end());
      used.erase(std::unique(used.begin(), used.end()), used.end());
      count = (int64_t)used.size();
    }
    if (count * modes <= max_dofs || scale > 64.0) break;
  }
  const int n_agg = local ? (int)used.size() : (int)(na[0] * na[1] * na[2]);
  std::vector<int32_t> agg_of_tile(T), agg_of_node(N);
  std::vector<double> cen((size_t)n_agg * 3, 0.0);
  for (int a = 0; a < n_agg; ++a) {
    const int64_t ga = local ? used[a] : a;
    int64_t ai[3];
    ai[ax[0]] = ga / (na[ax[1]] * na[ax[2]]);
    ai[ax[1]] = (ga / na[ax[2]]) % na[ax[1]];
    ai[ax[2]] = ga % na[ax[2]];
    for (int k = 0; k < 3; ++k) {   // centre of the aggregate's range of bricks
      const int64_t b_lo = (ai[k] * nbrick[k] + na[k] - 1) / na[k], b_hi = ((ai[k] + 1) * nbrick[k] + na[k] - 1) / na[k];
      cen[3 * a + k] = grid.lo[k] + 0.5 * (double)(b_lo + b_hi) * grid.side[k];
    }
  }
  for (int64_t t = 0; t < T; ++t) {
    const int64_t ga = grid_agg(tile_brick[t], g, na);
    const int a = local ? (int)(std::lower_bound(used.begin(), used.end(), ga) - used.begin()) : (int)ga;
    agg_of_tile[t] = a;
    for (int32_t i = tile_start[t]; i < tile_start[t + 1]; ++i) agg_of_node[i] = a;
  }
  {
    if (hipMalloc((void **)&c.dinv32, (size_t)N * 6 * sizeof(float)) != hipSuccess) return 2;
    // lever arms as floats, kept only if not one bit is lost (see rel32)
    c.rel_exact = false;
    if (!local && xyz_dev_order && !std::getenv("PL_NO_REL32")) {
      std::vector<float> rel((size_t)N * 3);
      bool exact = true;
      for (int64_t i = 0; i < N && exact; ++i)
        for (int k = 0; k < 3; ++k) {
          const double d = xyz_dev_order[3 * i + k] - cen[3 * (size_t)agg_of_node[i] + k];
          const float f = (float)d;
          if ((double)f != d) {
            exact = false;
            break;
          }
          rel[3 * (size_t)i + k] = f;
        }
      if (exact) {
        if (c.rel32.upload(rel) != hipSuccess) return 1;
        c.rel_exact = true;
      }
    }
  }
  c.n_agg = n_agg;
  c.nc = modes * n_agg;
  c.n_tiles = T;
  {
    int longest = 1;
    for (int64_t t = 0; t < T; ++t) longest = std::max(longest, (int)(tile_start[t + 1] - tile_start[t]));
    c.vblock = std::min(kBlock, (longest + kWave - 1) / kWave * kWave);
  }
  {
    // aggregate-crossing struts, grouped by ordered aggregate pair: a wave of the assembly kernel then works on ONE
    // coarse block and can reduce in registers before touching memory
    const int64_t B = (int64_t)conn.size() / 2;
    std::vector<std::pair<int64_t, int32_t>> cross;
    // 12 modes: the list holds every strut whose ends lie in different TILES (k_coarse_cross12)
    std::vector<int32_t> tile_of_node(N);
    for (int64_t t = 0; t < T; ++t)
      for (int32_t i = tile_start[t]; i < tile_start[t + 1]; ++i) tile_of_node[i] = (int32_t)t;
    if (c.tile_of_node.upload(tile_of_node) != hipSuccess) return 1;
    for (int64_t b = 0; b < B; ++b) {
      const int I = agg_of_node[conn[2 * b]], J = agg_of_node[conn[2 * b + 1]];
      if (I != J || (modes == 12 && tile_of_node[conn[2 * b]] != tile_of_node[conn[2 * b + 1]]))
        cross.push_back({(int64_t)I * n_agg + J, (int32_t)b});
    }
    // struts join neighbouring bricks only, so aggregates couple to their 26 neighbours: a bound from the aggregate
    // grid alone (identical on every rank of a multi-GPU run, whatever struts this rank holds)
    int64_t max_diff = std::min<int64_t>(n_agg - 1, na[ax[1]] * na[ax[2]] + na[ax[2]] + 1);
    if (local) {                   // compact numbering: take the band from the couplings that exist
      max_diff = 0;
      for (const auto &pr : cross) max_diff = std::max<int64_t>(max_diff, std::abs(pr.first / n_agg - pr.first % n_agg));
    }
    c.bw_blocks = (int)((modes * (max_diff + 1) + kNB - 1) / kNB + 1);
    for (const auto &pr : cross)   // struts longer than an aggregate (degenerate tiling): no band assumption
      if (std::abs(pr.first / n_agg - pr.first % n_agg) > max_diff) {
        // ... on one GPU.  On several, bw_blocks decides the size of the all-reduce of A_c and must be the same on
        // every rank, but only the ranks that hold such a strut would see it: refuse instead of hanging in RCCL
        if (multi_rank && !local) return 4;
        c.bw_blocks = 0;
      }
    std::sort(cross.begin(), cross.end());
    // every ordered pair starts at a wave boundary (padding = -1), so no wave mixes two coarse blocks
    std::vector<int32_t> idx2;
    idx2.reserve(cross.size() + cross.size() / 2);
    for (size_t q = 0; q < cross.size(); ++q) {
      if (q > 0 && cross[q].first != cross[q - 1].first)
        while (idx2.size() % kWave) idx2.push_back(-1);
      idx2.push_back(cross[q].second);
    }
    while (idx2.size() % kWave) idx2.push_back(-1);
    c.n_cross = (int64_t)idx2.size();
    if (c.cross_idx.upload(idx2) != hipSuccess) return 1;
  }
  if (c.agg_of_node.upload(agg_of_node) != hipSuccess || c.agg_of_tile.upload(agg_of_tile) != hipSuccess ||
      c.cen.upload(cen) != hipSuccess)
    return 1;
  c.ncp = (c.nc + kNB - 1) / kNB * kNB;
  const size_t n2 = (size_t)c.ncp * c.ncp;
  if (hipMalloc((void **)&c.Ac, n2 * sizeof(double)) != hipSuccess) return 2;
  c.ac_clean = false;
  if (hipMalloc((void **)&c.W, n2 * sizeof(float)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.Lf, n2 * sizeof(double)) != hipSuccess) return 2;
  if (hipMemset(c.Lf, 0, n2 * sizeof(double)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.Wt, n2 * sizeof(float)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.Dinv, (size_t)c.ncp * kNB * sizeof(double)) != hipSuccess) return 2;
  // r_c plus, in its tail, the kSlots partial sums each of r.r and r.D^-1 r (+ tile terms): on several GPUs the
  // whole buffer travels in ONE all-reduce and nothing has to be copied in or out of it
  if (hipMalloc((void **)&c.rc, (size_t)(c.ncp + 2 * kSlots) * sizeof(double)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.yc, (size_t)c.ncp * sizeof(double)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.tv, (size_t)c.ncp * sizeof(double)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.info, 2 * sizeof(int)) != hipSuccess) return 2;
  if (hipMalloc((void **)&c.bar, sizeof(unsigned)) != hipSuccess) return 2;
  {
    c.h_tile_start = tile_start;
    if (hipMalloc((void **)&c.Bt_inv, (size_t)T * 144 * sizeof(double)) != hipSuccess) return 2;
    if (hipMalloc((void **)&c.Bt_raw, (size_t)T * 144 * sizeof(double)) != hipSuccess) return 2;
    if (hipMalloc((void **)&c.yt, (size_t)T * 12 * sizeof(double)) != hipSuccess) return 2;
    if (hipMemset(c.Bt_inv, 0, (size_t)T * 144 * sizeof(double)) != hipSuccess) return 2;
    if (hipMemset(c.Bt_raw, 0, (size_t)T * 144 * sizeof(double)) != hipSuccess) return 2;
    if (hipMemset(c.yt, 0, (size_t)T * 12 * sizeof(double)) != hipSuccess) return 2;
  }
  if (hipMemset(c.rc, 0, (size_t)(c.ncp + 2 * kSlots) * sizeof(double)) != hipSuccess) return 2;
  if (hipMemset(c.W, 0, n2 * sizeof(float)) != hipSuccess) return 2;
  if (hipMemset(c.Wt, 0, n2 * sizeof(float)) != hipSuccess) return 2;
  c.enabled = true;
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// A_c = Z^T P K P Z : one thread per strut, only aggregate-crossing or Dirichlet-touching struts contribute
// (k_coarse_assemble_cross over the sorted crossing list, k_coarse_assemble over the device-built fixed list).
// ---------------------------------------------------------------------------------------------------------------
// keys == nullptr: count only.  key = aggregate << 32 | strut, so that the host can group the list by aggregate.
__global__ __launch_bounds__(kBlock) void k_list_fixed_struts(int64_t B, const int32_t *__restrict__ conn,
                                                              const int32_t *__restrict__ agg,
                                                              const uint8_t *__restrict__ fixedbits,
                                                              int64_t *__restrict__ keys, int *__restrict__ count) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const int ia = conn[2 * b], ib = conn[2 * b + 1];
  const int I = agg[ia];
  if (I != agg[ib]) return;
  if ((fixedbits[ia] | fixedbits[ib]) == 0u) return;
  const int slot = atomicAdd(count, 1);
  if (keys) keys[slot] = ((int64_t)I << 32) | (int64_t)b;
}

// Coarse block C = Zp^T (mask K mask) Zq into registers (no memory traffic).
__device__ __forceinline__ void coarse_block(const double *K, unsigned frow, unsigned fcol, const double *relp,
                                             const double *relq, double *C) {
  double Km[36];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) Km[i * 6 + j] = (((frow >> i) & 1u) || ((fcol >> j) & 1u)) ? 0.0 : K[i * 6 + j];
  const double Sq[3][3] = {{0, -relq[2], relq[1]}, {relq[2], 0, -relq[0]}, {-relq[1], relq[0], 0}};
  const double Sp[3][3] = {{0, -relp[2], relp[1]}, {relp[2], 0, -relp[0]}, {-relp[1], relp[0], 0}};
  double M[36];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i * 6 + j] = Km[i * 6 + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = Km[i * 6 + 3 + j];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc -= Km[i * 6 + k] * Sq[k][j];
      M[i * 6 + 3 + j] = acc;
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) C[i * 6 + j] = M[i * 6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double acc = M[(3 + i) * 6 + j];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc += Sp[i][k] * M[k * 6 + j];
      C[(3 + i) * 6 + j] = acc;
    }
  }
}

// In-aggregate struts that touch Dirichlet dofs: all four terms land in the diagonal block (I, I).  The list is
// grouped by aggregate and padded to wave boundaries (like the crossing list): sum across the wave, one lane issues
// the 36 atomics.
__global__ __launch_bounds__(kBlock) void k_coarse_assemble(int64_t n_list, const int32_t *__restrict__ list,
                                                            const int32_t *__restrict__ conn,
                                                            const Record *__restrict__ rec,
                                                            const int32_t *__restrict__ agg,
                                                            const double *__restrict__ cen,
                                                            const double *__restrict__ xyz,
                                                            const uint8_t *__restrict__ fixedbits, int nc,
                                                            double *__restrict__ Ac) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((q & ~(int64_t)(kWave - 1)) >= n_list) return;         // whole wave past the end (no block-level sync below)
  const bool live = list[q] >= 0;                            // -1 = padding (aggregates start at wave boundaries)
  const int64_t b = live ? list[q] : list[q & ~(int64_t)(kWave - 1)];
  const int ia = conn[2 * b], ib = conn[2 * b + 1];
  const int I = agg[ia];
  const unsigned fa = fixedbits[ia], fb = fixedbits[ib];
  const Record r = load_record(rec, b);
  double rela[3], relb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rela[k] = xyz[3 * (int64_t)ia + k] - cen[3 * I + k];
    relb[k] = xyz[3 * (int64_t)ib + k] - cen[3 * I + k];
  }
  double Kss[36], Kso[36], C[36], T[36];
  tip_blocks(r, Kss, Kso);                                   // K_bb, K_ba
  coarse_block(Kss, fb, fb, relb, relb, T);
  coarse_block(Kso, fb, fa, relb, rela, C);
#pragma unroll
  for (int e = 0; e < 36; ++e) T[e] += C[e];
  tip_blocks(reversed(r), Kss, Kso);                         // K_aa, K_ab
  coarse_block(Kss, fa, fa, rela, rela, C);
#pragma unroll
  for (int e = 0; e < 36; ++e) T[e] += C[e];
  coarse_block(Kso, fa, fb, rela, relb, C);
#pragma unroll
  for (int e = 0; e < 36; ++e) T[e] += C[e];
  double *dII = Ac + ((size_t)6 * I) * nc + 6 * I;
  const bool uniform = __all(I == __shfl(I, 0, 64));
  if (uniform) {
#pragma unroll
    for (int e = 0; e < 36; ++e) {
      const double v = wave_sum(live ? T[e] : 0.0);
      if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(dII + (size_t)(e / 6) * nc + e % 6, v);
    }
  } else if (live) {
#pragma unroll
    for (int e = 0; e < 36; ++e) unsafeAtomicAdd(dII + (size_t)(e / 6) * nc + e % 6, T[e]);
  }
}

// Aggregate-crossing struts in (I, J)-sorted order: when the whole wave works on one ordered pair (the common case,
// a pair has hundreds of struts) the three 6 x 6 contributions are summed across the wave with shuffles and ONE lane
// issues the atomics; mixed waves fall back to per-lane atomics.
__global__ __launch_bounds__(kBlock) void k_coarse_assemble_cross(int64_t n_cross, const int32_t *__restrict__ cross,
                                                                  const int32_t *__restrict__ conn,
                                                                  const Record *__restrict__ rec,
                                                                  const int32_t *__restrict__ agg,
                                                                  const double *__restrict__ cen,
                                                                  const double *__restrict__ xyz,
                                                                  const uint8_t *__restrict__ fixedbits, int nc,
                                                                  double *__restrict__ Ac) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((q & ~(int64_t)(kWave - 1)) >= n_cross) return;        // whole wave past the end (no block-level sync below)
  const bool live = cross[q] >= 0;
  const int64_t b = live ? cross[q] : cross[q & ~(int64_t)(kWave - 1)];   // lane 0 of a wave is never padding
  const int ia = conn[2 * b], ib = conn[2 * b + 1];
  const int I = agg[ia], J = agg[ib];
  const unsigned fa = fixedbits ? fixedbits[ia] : 0u, fb = fixedbits ? fixedbits[ib] : 0u;
  const Record r = load_record(rec, b);
  double rela[3], relb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rela[k] = xyz[3 * (int64_t)ia + k] - cen[3 * I + k];
    relb[k] = xyz[3 * (int64_t)ib + k] - cen[3 * J + k];
  }
  double Kss[36], Kso[36], Cjj[36], Cii[36], Cx[36];
  tip_blocks(r, Kss, Kso);                                   // K_bb, K_ba
  coarse_block(Kss, fb, fb, relb, relb, Cjj);
  if (J > I) coarse_block(Kso, fb, fa, relb, rela, Cx);      // lower block (J, I)
  tip_blocks(reversed(r), Kss, Kso);                         // K_aa, K_ab
  coarse_block(Kss, fa, fa, rela, rela, Cii);
  if (I > J) coarse_block(Kso, fa, fb, rela, relb, Cx);      // lower block (I, J)
  const int hi = I > J ? I : J, lo = I > J ? J : I;
  double *dII = Ac + ((size_t)6 * I) * nc + 6 * I, *dJJ = Ac + ((size_t)6 * J) * nc + 6 * J;
  double *dX = Ac + ((size_t)6 * hi) * nc + 6 * lo;
  const int I0 = __shfl(I, 0, 64), J0 = __shfl(J, 0, 64);
  const bool uniform = __all(I == I0 && J == J0);            // always, by the padding; dead lanes replicate lane 0
  if (uniform) {
#pragma unroll
    for (int e = 0; e < 36; ++e) {
      const double a = wave_sum(live ? Cii[e] : 0.0), c = wave_sum(live ? Cjj[e] : 0.0), x = wave_sum(live ? Cx[e] : 0.0);
      if ((threadIdx.x & 63) == 0) {
        const int i = e / 6, j = e - 6 * i;
        unsafeAtomicAdd(dII + (size_t)i * nc + j, a);
        unsafeAtomicAdd(dJJ + (size_t)i * nc + j, c);
        unsafeAtomicAdd(dX + (size_t)i * nc + j, x);
      }
    }
  } else if (live) {
#pragma unroll
    for (int e = 0; e < 36; ++e) {
      const int i = e / 6, j = e - 6 * i;
      unsafeAtomicAdd(dII + (size_t)i * nc + j, Cii[e]);
      unsafeAtomicAdd(dJJ + (size_t)i * nc + j, Cjj[e]);
      unsafeAtomicAdd(dX + (size_t)i * nc + j, Cx[e]);
    }
  }
}

// ---- 12-mode tile level: the strain rows / columns of B_t.  Mode 6 + q of a node at r = x - c (translations only):
//   q = 0, 1, 2: eps_xx, eps_yy, eps_zz -> u = (rx, 0, 0), (0, ry, 0), (0, 0, rz);
//   q = 3, 4, 5: eps_xy, eps_yz, eps_xz -> u = (ry, rx, 0) / 2, (0, rz, ry) / 2, (rz, 0, rx) / 2.
__device__ __forceinline__ V3 strain_disp(int q, const double *r) {
  switch (q) {
    case 0: return {r[0], 0.0, 0.0};
    case 1: return {0.0, r[1], 0.0};
    case 2: return {0.0, 0.0, r[2]};
    case 3: return {0.5 * r[1], 0.5 * r[0], 0.0};
    case 4: return {0.0, 0.5 * r[2], 0.5 * r[1]};
    default: return {0.5 * r[2], 0.0, 0.5 * r[0]};
  }
}
// Z_S^T f for a force F at r: the six strain restrictions
__device__ __forceinline__ void strain_restrict(const V3 &F, const double *r, double *s6, double sign) {
  s6[0] += sign * r[0] * F.x;
  s6[1] += sign * r[1] * F.y;
  s6[2] += sign * r[2] * F.z;
  s6[3] += sign * 0.5 * (r[1] * F.x + r[0] * F.y);
  s6[4] += sign * 0.5 * (r[2] * F.y + r[1] * F.z);
  s6[5] += sign * 0.5 * (r[2] * F.x + r[0] * F.z);
}
__device__ __forceinline__ V3 mask3(V3 v, unsigned bits) {
  return {(bits & 1u) ? 0.0 : v.x, (bits & 2u) ? 0.0 : v.y, (bits & 4u) ? 0.0 : v.z};
}
// ---- 12 modes per aggregate (rigid + uniform strains).  With the tile blocks B_t = Z_t^T P K P Z_t taken about the
// aggregate's reference point (k_tile_blocks + k_tile_blocks_strain, 12 x 12),
//     A_c = sum over tiles of B_t (into the diagonal block of the tile's aggregate)
//         + sum over struts whose ends lie in different tiles of Z_A^T K_AB Z_B and its transpose,
// since a tile block already holds K_AA / K_BB of every strut that touches the tile.
__global__ __launch_bounds__(kBlock) void k_agg_add_tiles(int64_t T, const int32_t *__restrict__ agg_of_tile,
                                                          const double *__restrict__ raw, int nc,
                                                          double *__restrict__ Ac) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t t = q / 144;
  if (t >= T) return;
  const int e = (int)(q - 144 * t), i = e / 12, j = e - 12 * i;
  if (j > i) return;                                       // the factorisation reads the lower triangle
  const int a = agg_of_tile[t];
  const double v = 0.5 * (raw[t * 144 + i * 12 + j] + raw[t * 144 + j * 12 + i]);
  if (v != 0.0) unsafeAtomicAdd(Ac + ((size_t)12 * a + i) * nc + 12 * a + j, v);
}
// displacement of mode n (0-5 rigid, 6-11 strain) of an aggregate at a node r = x - c
__device__ __forceinline__ void mode_disp(int n, const double *r, V3 &u, V3 &th) {
  th = {0.0, 0.0, 0.0};
  if (n >= 6) {
    u = strain_disp(n - 6, r);
    return;
  }
  switch (n) {
    case 0: u = {1.0, 0.0, 0.0}; break;
    case 1: u = {0.0, 1.0, 0.0}; break;
    case 2: u = {0.0, 0.0, 1.0}; break;
    case 3: u = {0.0, -r[2], r[1]}; th = {1.0, 0.0, 0.0}; break;      // e_x x r
    case 4: u = {r[2], 0.0, -r[0]}; th = {0.0, 1.0, 0.0}; break;
    default: u = {-r[1], r[0], 0.0}; th = {0.0, 0.0, 1.0}; break;
  }
}
// Cross-tile struts in (I, J)-sorted order, every ordered pair padded to whole waves (dead lanes replicate lane 0 and add
// nothing).  Column n of X = Z_A^T K_AB Z_B is the force on end A under mode n of B's aggregate at B (A held), restricted
// by the twelve modes of A's aggregate; X goes to (I, J) and its transpose to (J, I) (for I = J both into the same block).
__global__ __launch_bounds__(kBlock) void k_coarse_cross12(int64_t n_cross, const int32_t *__restrict__ list,
                                                           const int32_t *__restrict__ conn,
                                                           const Record *__restrict__ rec,
                                                           const int32_t *__restrict__ agg,
                                                           const double *__restrict__ cen,
                                                           const double *__restrict__ xyz,
                                                           const uint8_t *__restrict__ fixedbits, int nc,
                                                           double *__restrict__ Ac) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((q & ~(int64_t)(kWave - 1)) >= n_cross) return;        // whole wave past the end
  const bool live = list[q] >= 0;
  const int64_t b = live ? list[q] : list[q & ~(int64_t)(kWave - 1)];   // lane 0 of a wave is never padding
  const int ia = conn[2 * b], ib = conn[2 * b + 1];
  const int I = agg[ia], J = agg[ib];
  const unsigned fa = fixedbits ? fixedbits[ia] : 0u, fb = fixedbits ? fixedbits[ib] : 0u;
  const Record r = load_record(rec, b);
  double rela[3], relb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rela[k] = xyz[3 * (int64_t)ia + k] - cen[3 * I + k];
    relb[k] = xyz[3 * (int64_t)ib + k] - cen[3 * J + k];
  }
  const V3 d = {r.dx, r.dy, r.dz}, zero = {0.0, 0.0, 0.0};
  const int lane = threadIdx.x & 63;
  for (int n = 0; n < 12; ++n) {
    V3 uB, tB;
    mode_disp(n, relb, uB, tB);
    uB = mask3(uB, fb);
    tB = mask3(tB, fb >> 3);
    V3 F, M;
    tip_force(r, zero, zero, uB, tB, F, M);                  // force on B; on A: -F, -M - d x F
    const V3 FA = mask3((-1.0) * F, fa), MA = mask3((-1.0) * M - cross(d, F), fa >> 3);
    double x[12] = {FA.x, FA.y, FA.z,
                    MA.x + (rela[1] * FA.z - rela[2] * FA.y), MA.y + (rela[2] * FA.x - rela[0] * FA.z),
                    MA.z + (rela[0] * FA.y - rela[1] * FA.x), 0, 0, 0, 0, 0, 0};
    strain_restrict(FA, rela, x + 6, 1.0);
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const double v = wave_sum(live ? x[m] : 0.0);
      if (lane == 0 && v != 0.0) {
        const size_t ri = (size_t)12 * I + m, cj = (size_t)12 * J + n;
        // lower triangle only: X[m][n] at (ri, cj) or its mirror image; inside one aggregate X and X^T meet in the same
        // block, so its diagonal takes X[m][m] twice
        if (ri > cj) unsafeAtomicAdd(Ac + ri * nc + cj, v);
        else if (cj > ri) unsafeAtomicAdd(Ac + cj * nc + ri, v);
        else unsafeAtomicAdd(Ac + ri * nc + cj, 2.0 * v);
      }
    }
  }
}

// Rows/cols with a zero diagonal (aggregate without free support for that mode) -> identity; symmetrise round-off.
__global__ void k_coarse_regularize(int nc, double *__restrict__ Ac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nc) return;
  if (!(Ac[(size_t)i * nc + i] > 0.0)) Ac[(size_t)i * nc + i] = 1.0;
}
// ---------------------------------------------------------------------------------------------------------------
// Tile level: B_t = Z_t^T P K P Z_t (6 x 6) for the rigid-body modes of tile t about its aggregate's centre, inverted
// in place.  One workgroup per tile walks the tile's home + foreign strut lists (pl_tile.h), every lane keeps a
// 6 x 6 partial in registers, so there are no atomics.  Modes without stiffness (all their dofs fixed) are dropped.
// `fixedbits` is the Dirichlet mask, on several GPUs OR-ed with "shared with another rank": the tile modes live on
// this rank's own nodes only (a block that includes shared nodes would need the other rank's struts).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void spd6_inverse(double *A /* 36, in/out */) {
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) dmax = fmax(dmax, A[i * 6 + i]);
  double L[36];
  bool keep[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) L[e] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double s = A[j * 6 + j];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      if (k < j) s -= L[j * 6 + k] * L[j * 6 + k];
    keep[j] = s > 1e-12 * dmax;
    if (!keep[j]) {            // identity row/column
#pragma unroll
      for (int k = 0; k < 6; ++k) L[j * 6 + k] = 0.0;
      L[j * 6 + j] = 1.0;
      continue;
    }
    const double d = sqrt(s);
    L[j * 6 + j] = d;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i > j) {
        double v = 0.5 * (A[i * 6 + j] + A[j * 6 + i]);
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (k < j) v -= L[i * 6 + k] * L[j * 6 + k];
        L[i * 6 + j] = v / d;
      }
  }
  // W = L^-1 (lower), then A^-1 = W^T W
  double W[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) W[e] = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    W[j * 6 + j] = 1.0 / L[j * 6 + j];
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i > j) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (k >= j && k < i) v -= L[i * 6 + k] * W[k * 6 + j];
        W[i * 6 + j] = v / L[i * 6 + i];
      }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (k >= i && k >= j) v += W[k * 6 + i] * W[k * 6 + j];
      A[i * 6 + j] = (keep[i] && keep[j]) ? v : 0.0;
    }
}

// Multi-GPU: every rank holds the coarse-operator contribution of ITS struts and all ranks factor the sum.  Only the
// block band of the lower triangle is non-zero (and read by the factorisation), so that is what travels: rows of width
// (bwb + 1) * kNB starting at block column (i / kNB - bwb), packed into P (3 072 dofs, band of 4 blocks: 7.9 MB instead
// of the 75 MB of the full matrix).  Packing clears the band in A, unpacking rewrites it with the sum.
__global__ void k_band_pack(int n, int ld, int bwb, double *__restrict__ A, double *__restrict__ P) {
  const int W = (bwb + 1) * kNB;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * W) return;
  const int i = (int)(e / W), j = (i / kNB - bwb) * kNB + (int)(e % W);
  double v = 0.0;
  if (j >= 0) {
    v = A[(size_t)i * ld + j];
    A[(size_t)i * ld + j] = 0.0;
  }
  P[e] = v;
}
__global__ void k_band_unpack(int n, int ld, int bwb, const double *__restrict__ P, double *__restrict__ A) {
  const int W = (bwb + 1) * kNB;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n * W) return;
  const int i = (int)(e / W), j = (i / kNB - bwb) * kNB + (int)(e % W);
  if (j >= 0) A[(size_t)i * ld + j] = P[e];
}

__global__ __launch_bounds__(kBlock) void k_tile_blocks(const int32_t *__restrict__ tile_start,
                                                        const int64_t *__restrict__ home_ptr,
                                                        const int64_t *__restrict__ foreign_ptr,
                                                        const int32_t *__restrict__ foreign_idx,
                                                        const int2 *__restrict__ conn2, const Record *__restrict__ rec,
                                                        const int32_t *__restrict__ agg_of_tile,
                                                        const double *__restrict__ cen, const double *__restrict__ xyz,
                                                        const uint8_t *__restrict__ fixedbits,
                                                        double *__restrict__ Bt_inv,
                                                        double *__restrict__ raw /* 12 modes: [T][144], no inversion */) {
  __shared__ double red[36][kBlock / kWave];
  __shared__ double A[36];
  const int t = blockIdx.x;
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double acc[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
  const int64_t h0 = home_ptr[t], h1 = home_ptr[t + 1], f0 = foreign_ptr[t], f1 = foreign_ptr[t + 1];
  const int64_t total = (h1 - h0) + (f1 - f0);
  for (int64_t q = threadIdx.x; q < total; q += kBlock) {
    const int64_t b = q < (h1 - h0) ? h0 + q : (int64_t)foreign_idx[f0 + (q - (h1 - h0))];
    const int2 cn = conn2[b];
    const bool ina = cn.x >= n0 && cn.x < n1, inb = cn.y >= n0 && cn.y < n1;
    const unsigned fa = fixedbits ? fixedbits[cn.x] : 0u, fb = fixedbits ? fixedbits[cn.y] : 0u;
    if (ina && inb && fa == 0u && fb == 0u) continue;       // rigid motion of the whole strut
    const Record r = load_record(rec, b);
    const double rela[3] = {xyz[3 * (int64_t)cn.x] - c0, xyz[3 * (int64_t)cn.x + 1] - c1, xyz[3 * (int64_t)cn.x + 2] - c2};
    const double relb[3] = {xyz[3 * (int64_t)cn.y] - c0, xyz[3 * (int64_t)cn.y + 1] - c1, xyz[3 * (int64_t)cn.y + 2] - c2};
    double Kss[36], Kso[36], C[36];
    if (inb) {
      tip_blocks(r, Kss, Kso);                               // K_bb, K_ba
      coarse_block(Kss, fb, fb, relb, relb, C);
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[e] += C[e];
      if (ina) {
        coarse_block(Kso, fb, fa, relb, rela, C);
#pragma unroll
        for (int e = 0; e < 36; ++e) acc[e] += C[e];
      }
    }
    if (ina) {
      tip_blocks(reversed(r), Kss, Kso);                     // K_aa, K_ab
      coarse_block(Kss, fa, fa, rela, rela, C);
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[e] += C[e];
      if (inb) {
        coarse_block(Kso, fa, fb, rela, relb, C);
#pragma unroll
        for (int e = 0; e < 36; ++e) acc[e] += C[e];
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < 36; ++e) {
    const double s = wave_sum(acc[e]);
    if (lane == 0) red[e][wv] = s;
  }
  __syncthreads();
  if (threadIdx.x < 36) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < kBlock / kWave; ++q) s += red[threadIdx.x][q];
    A[threadIdx.x] = s;
  }
  __syncthreads();
  if (raw) {                                 // rigid x rigid block of the 12 x 12 matrix (k_tile_blocks_strain adds the rest)
    if (threadIdx.x < 36) raw[(size_t)t * 144 + (threadIdx.x / 6) * 12 + threadIdx.x % 6] = A[threadIdx.x];
    return;
  }
  if (threadIdx.x == 0) {
    double M[36];
#pragma unroll
    for (int e = 0; e < 36; ++e) M[e] = A[e];
    spd6_inverse(M);
#pragma unroll
    for (int e = 0; e < 36; ++e) Bt_inv[(size_t)t * 36 + e] = M[e];
  }
}

// One workgroup per tile over ALL the struts the tile visits (a strain mode stores energy in every strut, unlike a rigid
// one): per strut six evaluations of the tip force under the masked strain displacements of its in-tile ends, restricted
// back by the strain modes (S x S, 6 x 6 symmetric) and - only for struts that cross the tile boundary or touch a
// Dirichlet dof: the others answer a strain with a self-equilibrated force pair - by the rigid modes (R x S).
// RIGID = true: INSTEAD the rigid x rigid block (what k_tile_blocks(raw) computes from explicit 6 x 6 stiffness blocks and
// two 6 x 6 x 6 products per strut end) in the same force form: struts that cross the tile boundary or touch a Dirichlet dof
// take six tip forces under the masked rigid motions of their in-tile ends, restricted back by the rigid modes (symmetric,
// 21 sums).  (Both parts in ONE walk hold 78 accumulators: 256 VGPRs + 58 AGPRs, one wave per SIMD, 592 us against
// 517 || 226 us of the two kernels side by side.)
template <bool RIGID>
__global__ __launch_bounds__(kBlock) void k_tile_blocks_strain(const int32_t *__restrict__ tile_start,
                                                               const int64_t *__restrict__ home_ptr,
                                                               const int64_t *__restrict__ foreign_ptr,
                                                               const int32_t *__restrict__ foreign_idx,
                                                               const int2 *__restrict__ conn2,
                                                               const Record *__restrict__ rec,
                                                               const int32_t *__restrict__ agg_of_tile,
                                                               const double *__restrict__ cen,
                                                               const double *__restrict__ xyz,
                                                               const uint8_t *__restrict__ fixedbits,
                                                               double *__restrict__ raw) {
  __shared__ double red[57][kBlock / kWave];
  const int t = blockIdx.x;
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double SS[21], RS[RIGID ? 1 : 36];         // [p][q]: restriction p of the force answering strain q (S x S: p <= q only)
  double RR[RIGID ? 21 : 1];                 // R x R, p <= q, packed like SS
#pragma unroll
  for (int e = 0; e < (RIGID ? 21 : 1); ++e) RR[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 21; ++e) SS[e] = 0.0;
#pragma unroll
  for (int e = 0; e < (RIGID ? 1 : 36); ++e) RS[e] = 0.0;
  const int64_t h0 = home_ptr[t], h1 = home_ptr[t + 1], f0 = foreign_ptr[t], f1 = foreign_ptr[t + 1];
  const int64_t total = (h1 - h0) + (f1 - f0);
  for (int64_t v = threadIdx.x; v < total; v += kBlock) {
    const int64_t b = v < (h1 - h0) ? h0 + v : (int64_t)foreign_idx[f0 + (v - (h1 - h0))];
    const int2 cn = conn2[b];
    const bool ina = cn.x >= n0 && cn.x < n1, inb = cn.y >= n0 && cn.y < n1;
    if (!ina && !inb) continue;
    const unsigned fa = fixedbits ? fixedbits[cn.x] : 0u, fb = fixedbits ? fixedbits[cn.y] : 0u;
    const bool free_inside = ina && inb && fa == 0u && fb == 0u;
    if (RIGID && free_inside) continue;      // rigid motion of the whole strut
    const Record r = load_record(rec, b);
    const double rela[3] = {xyz[3 * (int64_t)cn.x] - c0, xyz[3 * (int64_t)cn.x + 1] - c1, xyz[3 * (int64_t)cn.x + 2] - c2};
    const double relb[3] = {xyz[3 * (int64_t)cn.y] - c0, xyz[3 * (int64_t)cn.y + 1] - c1, xyz[3 * (int64_t)cn.y + 2] - c2};
    const V3 d = {r.dx, r.dy, r.dz}, zero = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < (RIGID ? 0 : 6); ++q) {
      const V3 uA = ina ? mask3(strain_disp(q, rela), fa) : zero, uB = inb ? mask3(strain_disp(q, relb), fb) : zero;
      V3 F, M;
      tip_force(r, uA, zero, uB, zero, F, M);                  // on B; on A: -F, -M - d x F
      const V3 FB = mask3(F, fb), MB = mask3(M, fb >> 3);
      const V3 FA = mask3((-1.0) * F, fa), MA = mask3((-1.0) * M - cross(d, F), fa >> 3);
      double s6[6] = {0, 0, 0, 0, 0, 0};
      if (inb) strain_restrict(FB, relb, s6, 1.0);
      if (ina) strain_restrict(FA, rela, s6, 1.0);
#pragma unroll
      for (int p = 0; p <= q; ++p) SS[q * (q + 1) / 2 + p] += s6[p];        // upper triangle, packed by columns
      if (!free_inside) {
        if (inb) {
          RS[0 * 6 + q] += FB.x;
          RS[1 * 6 + q] += FB.y;
          RS[2 * 6 + q] += FB.z;
          RS[3 * 6 + q] += MB.x + (relb[1] * FB.z - relb[2] * FB.y);
          RS[4 * 6 + q] += MB.y + (relb[2] * FB.x - relb[0] * FB.z);
          RS[5 * 6 + q] += MB.z + (relb[0] * FB.y - relb[1] * FB.x);
        }
        if (ina) {
          RS[0 * 6 + q] += FA.x;
          RS[1 * 6 + q] += FA.y;
          RS[2 * 6 + q] += FA.z;
          RS[3 * 6 + q] += MA.x + (rela[1] * FA.z - rela[2] * FA.y);
          RS[4 * 6 + q] += MA.y + (rela[2] * FA.x - rela[0] * FA.z);
          RS[5 * 6 + q] += MA.z + (rela[0] * FA.y - rela[1] * FA.x);
        }
      }
    }
    if (RIGID && !free_inside) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        V3 uA = zero, tA = zero, uB = zero, tB = zero;
        if (ina) {
          mode_disp(q, rela, uA, tA);
          uA = mask3(uA, fa);
          tA = mask3(tA, fa >> 3);
        }
        if (inb) {
          mode_disp(q, relb, uB, tB);
          uB = mask3(uB, fb);
          tB = mask3(tB, fb >> 3);
        }
        V3 F, M;
        tip_force(r, uA, tA, uB, tB, F, M);
        double x6[6] = {0, 0, 0, 0, 0, 0};
        if (inb) {
          const V3 FB = mask3(F, fb), MB = mask3(M, fb >> 3);
          x6[0] += FB.x;
          x6[1] += FB.y;
          x6[2] += FB.z;
          x6[3] += MB.x + (relb[1] * FB.z - relb[2] * FB.y);
          x6[4] += MB.y + (relb[2] * FB.x - relb[0] * FB.z);
          x6[5] += MB.z + (relb[0] * FB.y - relb[1] * FB.x);
        }
        if (ina) {
          const V3 FA = mask3((-1.0) * F, fa), MA = mask3((-1.0) * M - cross(d, F), fa >> 3);
          x6[0] += FA.x;
          x6[1] += FA.y;
          x6[2] += FA.z;
          x6[3] += MA.x + (rela[1] * FA.z - rela[2] * FA.y);
          x6[4] += MA.y + (rela[2] * FA.x - rela[0] * FA.z);
          x6[5] += MA.z + (rela[0] * FA.y - rela[1] * FA.x);
        }
#pragma unroll
        for (int p = 0; p <= q; ++p) RR[q * (q + 1) / 2 + p] += x6[p];
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *B = raw + (size_t)t * 144;
  if (RIGID) {
#pragma unroll
    for (int e = 0; e < 21; ++e) {
      const double s0 = wave_sum(RR[e]);
      if (lane == 0) red[e][wv] = s0;
    }
    __syncthreads();
    if (threadIdx.x < 21) {
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < kBlock / kWave; ++q) v += red[threadIdx.x][q];
      int q = 0;
      while ((q + 1) * (q + 2) / 2 <= (int)threadIdx.x) ++q;     // column of the packed upper triangle
      const int p = (int)threadIdx.x - q * (q + 1) / 2;
      B[p * 12 + q] = v;                      // R x R, both halves
      B[q * 12 + p] = v;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 21; ++e) {
    const double s1 = wave_sum(SS[e]);
    if (lane == 0) red[e][wv] = s1;
  }
#pragma unroll
  for (int e = 0; e < (RIGID ? 1 : 36); ++e) {
    const double s2 = wave_sum(RS[e]);
    if (lane == 0) red[21 + e][wv] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 57) {
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < kBlock / kWave; ++q) v += red[threadIdx.x][q];
    if (threadIdx.x < 21) {
      int q = 0;
      while ((q + 1) * (q + 2) / 2 <= (int)threadIdx.x) ++q;     // column of the packed upper triangle
      const int p = (int)threadIdx.x - q * (q + 1) / 2;
      B[(6 + p) * 12 + 6 + q] = v;            // S x S, both halves
      B[(6 + q) * 12 + 6 + p] = v;
    } else {
      const int e = (int)threadIdx.x - 21, p = e / 6, q = e % 6;
      B[p * 12 + 6 + q] = v;                  // R x S and its transpose
      B[(6 + q) * 12 + p] = v;
    }
  }
}
// B_t^-1 of the 12-mode tile level: one thread per tile, the matrix in LDS (column of 64 threads: conflict-free), inverted
// in place by twelve symmetric sweeps (Gauss-Jordan on an SPD matrix needs no pivoting); a mode whose pivot has lost ten
// digits against its own diagonal entry has no stiffness of its own (all its dofs fixed, or it repeats earlier modes) and
// is dropped - zero row and column, as in spd6_inverse.
// (Round 5: one workgroup per tile, one lane per ENTRY of the 12 x 12 block - every pivot step updates all 144 entries at
// once from the values of the step before.  The first version swept the block serially, one lane per tile: 101 us for the
// few hundred tiles of configs[3], a quarter of its assembly.  Same arithmetic per entry, same dropping rule.)
constexpr int kInv12Block = 192;
__global__ __launch_bounds__(kInv12Block) void k_tile_invert12(int64_t T, const double *__restrict__ raw,
                                                              double *__restrict__ Bt_inv) {
  constexpr int n = 12;
  __shared__ double A[n][n + 1];
  __shared__ double diag0[n];
  const int64_t t = blockIdx.x;
  if (t >= T) return;
  const int e = threadIdx.x, i = e / n, j = e - n * i;
  const bool act = e < n * n;
  if (act) A[i][j] = 0.5 * (raw[t * 144 + i * n + j] + raw[t * 144 + j * n + i]);
  __syncthreads();
  if (e < n) diag0[e] = A[e][e];
  __syncthreads();
  unsigned dropped = 0u;
  for (int k = 0; k < n; ++k) {
    const double d = A[k][k];
    if (!(d > 1e-10 * diag0[k]) || !(diag0[k] > 0.0)) {      // (the same decision in every lane)
      dropped |= 1u << k;
      continue;
    }
    const double inv = 1.0 / d;
    double v = 0.0;
    if (act) {
      const bool di = (dropped >> i) & 1u, dj = (dropped >> j) & 1u;
      if (i == k && j == k) v = -inv;
      else if (i == k) v = A[k][j] * inv;                    // row k, scaled
      else if (j == k) v = A[k][i] * inv;                    // ... and its mirror image
      else if (di || dj) v = A[i][j];                        // rows / columns of dropped modes stay out of the sweep
      else v = A[i][j] - (A[i][k] * inv) * A[k][j];
    }
    __syncthreads();
    if (act) A[i][j] = v;
    __syncthreads();
  }
  // after sweeping every kept pivot the kept block holds -B^-1
  if (act) {
    const bool out = ((dropped >> i) & 1u) || ((dropped >> j) & 1u);
    Bt_inv[t * 144 + i * n + j] = out ? 0.0 : -A[i][j];
  }
}
// The dense level's factorisation and solve in whichever storage type the level uses.
inline void coarse_factor(Coarse &cs, int n, hipStream_t s, const std::function<void(int)> &after_chol, unsigned *bar,
                          TrtriPhases ph = TrtriPhases()) {
  if (cs.w16)
    dense_factor_inverse(cs.Ac, cs.Lf, reinterpret_cast<bf16_t *>(cs.W), reinterpret_cast<bf16_t *>(cs.Wt), cs.Dinv, n, n,
                         cs.info, cs.bw_blocks, s, after_chol, bar, ph);
  else
    dense_factor_inverse(cs.Ac, cs.Lf, cs.W, cs.Wt, cs.Dinv, n, n, cs.info, cs.bw_blocks, s, after_chol, bar, ph);
}
// levels up to this size apply their explicit inverse in ONE GEMV launch (k_full_gemv) instead of two triangular ones
constexpr int kOneGemvMaxDofs = 2048;
inline void coarse_apply(const Coarse &cs, const double *r, double *t, double *y, double *dot_out, const double *add0,
                         hipStream_t s) {
  static const bool one_off = [] { const char *e = std::getenv("PL_ONE_GEMV"); return e && e[0] == '0'; }();
  if (cs.ainv_ready && cs.Ainv && cs.ncp <= kOneGemvMaxDofs && !one_off) {
    hipLaunchKernelGGL(k_full_gemv, dim3((cs.ncp + 3) / 4), dim3(kBlock), 0, s, cs.ncp, (const float *)cs.Ainv, cs.ncp, r, y,
                       dot_out, add0);
    return;
  }
  if (cs.w16)
    dense_apply(reinterpret_cast<const bf16_t *>(cs.W), reinterpret_cast<const bf16_t *>(cs.Wt), cs.ncp, cs.ncp, r, t, y,
                dot_out, add0, s);
  else
    dense_apply(cs.W, cs.Wt, cs.ncp, cs.ncp, r, t, y, dot_out, add0, s);
}

// ---------------------------------------------------------------------------------------------------------------
// r -= alpha Ap ; per-tile partials (Z^T r [6], r.r, r.D^-1 r) — one workgroup per tile.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_to_float(int64_t n, const double *__restrict__ x, float *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) y[i] = (float)x[i];
}

// PT = storage type of the search direction p and of Ap, RT = of the iterate x and the residual r (double / float:
// opts.precision).  Every sum is accumulated in fp64 whatever the storage.
// TM = modes of the tile level: 6, or 12 (rigid + uniform strains, single-GPU handles: Coarse::tile_modes)
// MULTI = false (one GPU, no rank-local level): the weights, the shared-node flags and the rank-local aggregates are known
// to be absent at compile time, and with them go 18 accumulators - the general kernel holds 164 VGPRs (3 waves per SIMD).
// LOCAL = false: no rank-local dense level (precond = 4 only), six accumulators less on multi-rank handles.
template <typename PT, typename RT, int TM = 6, bool MULTI = true, bool LOCAL = true>
__global__ __launch_bounds__(kBlock) void k_pcg_update_tile(const int32_t *__restrict__ tile_start,
                                                            const int32_t *__restrict__ agg_of_tile,
                                                            const double *__restrict__ cen,
                                                            const double *__restrict__ xyz,
                                                            const PT *__restrict__ Ap,
                                                            const float *__restrict__ dinv32,
                                                            const double *__restrict__ w /* may be null */,
                                                            RT *__restrict__ r,
                                                            double *__restrict__ scal, double *__restrict__ rc,
                                                            const double *__restrict__ Bt_inv /* may be null */,
                                                            double *__restrict__ yt,
                                                            const int32_t *__restrict__ aggL_of_tile /* may be null */,
                                                            const double *__restrict__ cenL,
                                                            const uint8_t *__restrict__ shared /* may be null */,
                                                            double *__restrict__ rcL, int ncp,
                                                            const uint8_t *__restrict__ skip_rows /* may be null */,
                                                            int cm = 6 /* modes per aggregate of the dense level */,
                                                            const float *__restrict__ rel32 = nullptr /* Coarse::rel32 */) {
  __shared__ double red[32][4 * kBlock / kWave];   // one partial per row of 16 lanes (row_sums)
  if constexpr (!MULTI) {
    w = nullptr;
    shared = nullptr;
  }
  if constexpr (!MULTI || !LOCAL) aggL_of_tile = nullptr;
  double *rr_slot = rc + ncp + (blockIdx.x & (kSlots - 1)), *rdr_slot = rr_slot + kSlots;   // tail of r_c
  const int t = blockIdx.x;
  double pap, old_rz;
  scalar_read2(scal, S_PAP, S_RZ_OLD, pap, old_rz);
  const double alpha = (pap != 0.0) ? old_rz / pap : 0.0;
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  // row of B_t^-1 of the lane that will need it at the very end, fetched early.  TM = 12: lanes 0-5 hold the rigid
  // modes, 8-13 the strain modes (6, 7 carry r.r and r.D^-1 r as before)
  // (parked in LDS, not in registers: 2 TM VGPRs that every wave would carry through the loop cost a wave per SIMD)
  __shared__ double sbi[16][TM];
  double bi[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) bi[j] = 0.0;
  const int my_mode = (TM == 12) ? ((int)threadIdx.x < 6 ? (int)threadIdx.x : ((int)threadIdx.x >= 8 && (int)threadIdx.x < 14
                                                                                   ? (int)threadIdx.x - 2 : -1))
                                 : ((int)threadIdx.x < 6 ? (int)threadIdx.x : -1);
  if (Bt_inv && my_mode >= 0) {
#pragma unroll
    for (int j = 0; j < TM; ++j) bi[j] = Bt_inv[(size_t)t * (TM * TM) + TM * my_mode + j];
  }
  if (threadIdx.x < 16) {
#pragma unroll
    for (int j = 0; j < TM; ++j) sbi[threadIdx.x][j] = bi[j];
  }
  double accS[6] = {0, 0, 0, 0, 0, 0};   // TM = 12: strain restrictions of the tile
  double accD[6] = {0, 0, 0, 0, 0, 0};   // ... and, on several GPUs with a 12-mode dense level, of the aggregate (weighted,
                                         // shared nodes included; on one GPU the two are the same sums)
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // rank-local dense level (multi-GPU): its own aggregates / reference points, nodes shared with other ranks left out
  const int aL = aggL_of_tile ? aggL_of_tile[t] : 0;
  double l0 = 0.0, l1 = 0.0, l2 = 0.0;
  if (aggL_of_tile) {
    l0 = cenL[3 * aL];
    l1 = cenL[3 * aL + 1];
    l2 = cenL[3 * aL + 2];
  }
  double accL[6] = {0, 0, 0, 0, 0, 0};
  double accT[6] = {0, 0, 0, 0, 0, 0};      // tile level on several GPUs: the tile's restriction without shared nodes
  const bool own_t = Bt_inv && shared;
  for (int i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    if (skip_rows && skip_rows[i]) continue;    // eliminated node (opts.condense): not an unknown of this CG
    // (x += alpha p is done by k_pcg_direction_coarse, which reads p anyway: one vector pass less per iteration)
    double av[6], dv[6], rv[6];
    const float2 *d2 = reinterpret_cast<const float2 *>(dinv32 + 6 * (int64_t)i);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double2 aa = load_pair(Ap, 3 * (int64_t)i + k), rr = load_pair(r, 3 * (int64_t)i + k);
      const float2 dd = d2[k];
      av[2 * k] = aa.x; av[2 * k + 1] = aa.y;
      dv[2 * k] = dd.x; dv[2 * k + 1] = dd.y;
      rv[2 * k] = rr.x; rv[2 * k + 1] = rr.y;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) rv[k] -= alpha * av[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) store_pair(r, 3 * (int64_t)i + k, double2{rv[2 * k], rv[2 * k + 1]});
    if (sizeof(RT) == 4) {   // the sums below must see the residual as it is stored
#pragma unroll
      for (int k = 0; k < 6; ++k) rv[k] = (double)(float)rv[k];
    }
    double wt[6] = {1, 1, 1, 1, 1, 1};
    if (w) {
#pragma unroll
      for (int k = 0; k < 6; ++k) wt[k] = w[6 * (int64_t)i + k];
    }
    double rx, ry, rz;
    if (rel32) {          // the same three numbers as 12 bytes (exact by construction: Coarse::rel_exact)
      rx = (double)rel32[3 * (int64_t)i];
      ry = (double)rel32[3 * (int64_t)i + 1];
      rz = (double)rel32[3 * (int64_t)i + 2];
    } else {
      rx = xyz[3 * (int64_t)i] - c0;
      ry = xyz[3 * (int64_t)i + 1] - c1;
      rz = xyz[3 * (int64_t)i + 2] - c2;
    }
    const double ru[3] = {wt[0] * rv[0], wt[1] * rv[1], wt[2] * rv[2]};
    acc[0] += ru[0];
    acc[1] += ru[1];
    acc[2] += ru[2];
    acc[3] += wt[3] * rv[3] + (ry * ru[2] - rz * ru[1]);
    acc[4] += wt[4] * rv[4] + (rz * ru[0] - rx * ru[2]);
    acc[5] += wt[5] * rv[5] + (rx * ru[1] - ry * ru[0]);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      acc[6] += wt[k] * rv[k] * rv[k];
      acc[7] += wt[k] * dv[k] * rv[k] * rv[k];
    }
    if (TM == 12 && cm == 12 && shared) {
      accD[0] += rx * ru[0];
      accD[1] += ry * ru[1];
      accD[2] += rz * ru[2];
      accD[3] += 0.5 * (ry * ru[0] + rx * ru[1]);
      accD[4] += 0.5 * (rz * ru[1] + ry * ru[2]);
      accD[5] += 0.5 * (rz * ru[0] + rx * ru[2]);
    }
    if (TM == 12 && !(shared && shared[i])) {   // (several GPUs: the tile's strain modes live on this rank's own nodes)
      accS[0] += rx * ru[0];
      accS[1] += ry * ru[1];
      accS[2] += rz * ru[2];
      accS[3] += 0.5 * (ry * ru[0] + rx * ru[1]);
      accS[4] += 0.5 * (rz * ru[1] + ry * ru[2]);
      accS[5] += 0.5 * (rz * ru[0] + rx * ru[2]);
    }
    if (own_t && !shared[i]) {
      accT[0] += rv[0];
      accT[1] += rv[1];
      accT[2] += rv[2];
      accT[3] += rv[3] + (ry * rv[2] - rz * rv[1]);
      accT[4] += rv[4] + (rz * rv[0] - rx * rv[2]);
      accT[5] += rv[5] + (rx * rv[1] - ry * rv[0]);
    }
    if (aggL_of_tile && !(shared && shared[i])) {
      const double sx = xyz[3 * (int64_t)i] - l0, sy = xyz[3 * (int64_t)i + 1] - l1, sz = xyz[3 * (int64_t)i + 2] - l2;
      accL[0] += rv[0];
      accL[1] += rv[1];
      accL[2] += rv[2];
      accL[3] += rv[3] + (sy * rv[2] - sz * rv[1]);
      accL[4] += rv[4] + (sz * rv[0] - sx * rv[2]);
      accL[5] += rv[5] + (sx * rv[1] - sy * rv[0]);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nw = 4 * (blockDim.x >> 6);         // partial sums per quantity: one per row of 16 lanes
  const bool row_end = (lane & 15) == 15;
  const int slot = 4 * wv + (lane >> 4);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double s = row_sums(acc[k]);
    if (row_end) red[k][slot] = s;
  }
  if (aggL_of_tile) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double s = row_sums(accL[k]);
      if (row_end) red[8 + k][slot] = s;
    }
  }
  if (own_t) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double s = row_sums(accT[k]);
      if (row_end) red[14 + k][slot] = s;
    }
  }
  if (TM == 12) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double s = row_sums(accS[k]);
      if (row_end) red[20 + k][slot] = s;
    }
    if (cm == 12 && shared) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const double s = row_sums(accD[k]);
        if (row_end) red[26 + k][slot] = s;
      }
    }
  }
  __syncthreads();
  const int pubL = blockDim.x > kWave ? kWave : 8;   // wave 1 publishes the local restriction (lanes 8..13 of wave 0 if it is alone)
  if (aggL_of_tile && (int)threadIdx.x >= pubL && (int)threadIdx.x < pubL + 6) {
    const int k = threadIdx.x - pubL;
    double s = 0.0;
    for (int q = 0; q < nw; ++q) s += red[8 + k][q];
    unsafeAtomicAdd(rcL + 6 * aL + k, s);
  }
  if (TM == 12) {   // lanes 0-5 rigid sums, 6 r.r, 7 r.D^-1 r, 8-13 strain sums
    if (threadIdx.x < 16) {
      const int src = (int)threadIdx.x < 8 ? (int)threadIdx.x : 12 + (int)threadIdx.x;    // rows 20..25 of red for lanes 8..13
      double s = 0.0;
      if (threadIdx.x < 14)
        for (int q = 0; q < nw; ++q) s += red[src][q];
      if (threadIdx.x < 6) unsafeAtomicAdd(rc + cm * a + threadIdx.x, s);
      else if (threadIdx.x == 6) unsafeAtomicAdd(rr_slot, s);
      else if (cm == 12 && threadIdx.x >= 8 && threadIdx.x < 14) {
        double sd = s;                                       // the aggregate's strain sums
        if (shared) {
          sd = 0.0;
          for (int q = 0; q < nw; ++q) sd += red[18 + threadIdx.x][q];      // rows 26..31
        }
        unsafeAtomicAdd(rc + cm * a + threadIdx.x - 2, sd);
      }
      if (!Bt_inv) {
        if (threadIdx.x == 7) unsafeAtomicAdd(rdr_slot, s);
      } else {
        // the 12 x 12 product: the twelve restriction sums sit in lanes 0-5 and 8-13 of this wave and reach every lane by
        // v_readlane (lane numbers are compile-time constants).  (Rounds 2 - 4 passed them through LDS without a barrier, which
        // is a data race under the HIP memory model; with wavefront-scope fences the release waited for the atomics issued just
        // above: +2.3 us per iteration at 50^3 Octet - measured in round 5.)
        double st = my_mode >= 0 ? s : 0.0;                // this lane's component of the tile restriction
        if (own_t && threadIdx.x < 6) {                    // several GPUs: the rigid part without the shared nodes
          st = 0.0;
          for (int q = 0; q < nw; ++q) st += red[14 + threadIdx.x][q];
        }
        double y = 0.0;
#pragma unroll
        for (int j = 0; j < 12; ++j) y += sbi[threadIdx.x][j] * lane_value(st, j < 6 ? j : j + 2);
        if (my_mode >= 0) yt[12 * (size_t)t + my_mode] = y;
        const double v = row_sums(my_mode >= 0 ? y * st : (threadIdx.x == 7 ? s : 0.0));     // lane 15: the sum over lanes 0-15
        if (threadIdx.x == 15) unsafeAtomicAdd(rdr_slot, v);
      }
    }
    return;
  }
  if (threadIdx.x < 8) {
    double s = 0.0;
    for (int q = 0; q < nw; ++q) s += red[threadIdx.x][q];
    // r_c is zeroed by the previous direction kernel; ~8 tiles add into each aggregate's six entries
    if (threadIdx.x < 6) unsafeAtomicAdd(rc + 6 * a + threadIdx.x, s);
    else if (threadIdx.x == 6) unsafeAtomicAdd(rr_slot, s);
    if (!Bt_inv) {
      if (threadIdx.x == 7) unsafeAtomicAdd(rdr_slot, s);
    } else {
      // tile level: y_t = B_t^-1 (Z_t^T r); its share r_t . y_t of r.z joins r.D^-1 r (lanes 0..7 of wave 0)
      double st = s;                           // this lane's component of the TILE restriction
      if (own_t) {
        st = 0.0;
        if (threadIdx.x < 6) {
          for (int q = 0; q < nw; ++q) st += red[14 + threadIdx.x][q];
        }
      }
      double tj[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) tj[j] = __shfl(st, j, 8);
      double y = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) y += sbi[threadIdx.x][j] * tj[j];
      if (threadIdx.x < 6) yt[6 * (size_t)t + threadIdx.x] = y;
      double v = threadIdx.x < 6 ? y * st : (threadIdx.x == 7 ? s : 0.0);
      v += __shfl_xor(v, 1, 8);
      v += __shfl_xor(v, 2, 8);
      v += __shfl_xor(v, 4, 8);
      if (threadIdx.x == 0) unsafeAtomicAdd(rdr_slot, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Exact elimination of an independent set of nodes inside the PCG (opts.condense).  No two condensed nodes share a strut
// and none carries a Dirichlet dof, so K_cc is block diagonal (6 x 6 per node) and
//     S p = (K [p_v ; -K_cc^-1 K_cv p_v])_v
// is two passes of the ordinary K*x with a per-node 6 x 6 product in between: CG then runs on the Schur complement
// of the remaining nodes while x accumulates the condensed nodes' exact (equilibrium) displacements.  For lattices
// whose node graph is bipartite (BCC: cell centres vs corners) this removes half of the unknowns - the half the
// diagonal smoother handles worst.
// ---------------------------------------------------------------------------------------------------------------
// K_cc^-1 of every condensed node from the sliced-ELL incidence (one thread per node; SN = nodes per ELL slice).
__global__ __launch_bounds__(kBlock) void k_node_block_inverse(int64_t nc, const int32_t *__restrict__ cnodes, int SN,
                                                               const int64_t *__restrict__ slice_ptr,
                                                               const int2 *__restrict__ ent,
                                                               const Record *__restrict__ rec,
                                                               double *__restrict__ inv) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nc) return;
  const int64_t i = cnodes[q], s = i / SN, lane = i % SN;
  const int64_t p0 = slice_ptr[s], width = (slice_ptr[s + 1] - p0) / SN;
  double A[36];
#pragma unroll
  for (int e = 0; e < 36; ++e) A[e] = 0.0;
  for (int64_t j = 0; j < width; ++j) {
    const int2 e = ent[p0 + j * SN + lane];
    if (e.x < 0) continue;
    Record r = load_record(rec, e.y & 0x7fffffff);
    if (e.y < 0) r = reversed(r);
    double Kss[36], Kso[36];
    tip_blocks(r, Kss, Kso);
#pragma unroll
    for (int k = 0; k < 36; ++k) A[k] += Kss[k];
  }
  spd6_inverse(A);
#pragma unroll
  for (int k = 0; k < 36; ++k) inv[36 * q + k] = A[k];
}
// v[node] = sign * K_cc^-1 y[node] for every condensed node (sign -1: the equilibrium position under the forces y the
// other nodes exert; +1: the response to a load y)
// (cls != null: inv is the class table of k_cls_* below, cls[q] the node's entry - a lattice with a record palette has a
// few dozen distinct K_cc, and 288 bytes per eliminated node and iteration were the largest stream of the elimination)
template <typename VT>
__global__ __launch_bounds__(kBlock) void k_condense_solve(int64_t nc, const int32_t *__restrict__ cnodes,
                                                           const double *__restrict__ inv,
                                                           const uint16_t *__restrict__ cls, const VT *__restrict__ y,
                                                           VT *__restrict__ v, double sign) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / 6;
  const int k = (int)(t - 6 * q);
  if (q >= nc) return;
  const int64_t i = cnodes[q];
  const double *A = inv + 36 * (cls ? (int64_t)cls[q] : q);
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) acc += A[6 * k + j] * (double)y[6 * i + j];
  v[6 * i + k] = (VT)(sign * acc);
}
// (Round 3, tried and dropped: the first pass of the condensed operator by ROWS - one group of LPN lanes per eliminated node
// over the sliced-ELL incidence, one visit per strut, no LDS accumulator, the 6 x 6 solve in the lanes that hold the sums.
// 100^3 BCC: iteration 327 -> 344 us with 4 lanes per node, 360 / 369 / 461 us with 2 / 8 / 16: the dependent chain
// entry -> palette id -> record / entry -> x row with nothing prefetched costs more than the tile kernel's LDS scatter.)
// ---- classes of eliminated nodes with the same K_cc: nodes whose incident struts carry the same multiset of (record
// palette id, end) have the same block up to the order of the sum.  Same scheme as pl_palette.h: hash -> claim a slot ->
// the owner publishes its inverse -> every node verifies (1e-10: a hash collision would pair DIFFERENT blocks).
__global__ __launch_bounds__(kBlock) void k_cls_hash(int64_t nc, const int32_t *__restrict__ cnodes, int SN,
                                                     const int64_t *__restrict__ slice_ptr,
                                                     const int2 *__restrict__ ent, const uint16_t *__restrict__ pal,
                                                     unsigned long long *__restrict__ key) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nc) return;
  const int64_t i = cnodes[q], s = i / SN, lane = i % SN;
  const int64_t p0 = slice_ptr[s], width = (slice_ptr[s + 1] - p0) / SN;
  unsigned long long h = 0;
  for (int64_t j = 0; j < width; ++j) {
    const int2 e = ent[p0 + j * SN + lane];
    if (e.x < 0) continue;
    unsigned long long v = ((unsigned long long)pal[e.y & 0x7fffffff] << 1) | (e.y < 0 ? 1ull : 0ull);
    v = (v + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    v ^= v >> 29;
    v *= 0x94D049BB133111EBull;
    v ^= v >> 32;
    h += v;                                   // commutative: the order of the incident struts does not matter
  }
  key[q] = h == ~0ull ? 0 : h;
}
// (one probe per wave and distinct key, as k_pal_insert: a million nodes share a few dozen classes)
__device__ __forceinline__ int cls_probe(unsigned long long h, unsigned long long *__restrict__ keys,
                                         int *__restrict__ owner, int q) {
  unsigned slot = (unsigned)(h >> 48);
  for (int probe = 0; probe < 32; ++probe) {
    unsigned long long cur = __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == ~0ull) cur = atomicCAS(keys + slot, ~0ull, h);
    if (cur == ~0ull || cur == h) {
      if (q < __hip_atomic_load(owner + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(owner + slot, q);
      return (int)slot;
    }
    slot = (slot + 1) & 65535u;
  }
  return -1;
}
__global__ __launch_bounds__(kBlock) void k_cls_insert(int64_t nc, const unsigned long long *__restrict__ key,
                                                       unsigned long long *__restrict__ keys, int *__restrict__ owner,
                                                       uint16_t *__restrict__ cls, int *__restrict__ flags) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = q < nc;
  const unsigned long long h = valid ? key[q] : 0ull;
  const int lane = threadIdx.x & 63;
  int found = -2;                                   // -2: not settled yet, -1: table full
  unsigned long long todo = __ballot(valid);
  for (int round = 0; round < 12 && todo; ++round) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned long long hl =
        ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(h >> 32), leader) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(h & 0xFFFFFFFFull), leader);
    const unsigned long long same = __ballot(valid && found == -2 && h == hl);
    int sl = 0;
    if (lane == leader) sl = cls_probe(h, keys, owner, (int)q);   // the lowest lane of the group: the smallest node
    sl = __builtin_amdgcn_readlane(sl, leader);
    if ((same >> lane) & 1ull) found = sl;
    todo &= ~same;
  }
  if (valid && found == -2) found = cls_probe(h, keys, owner, (int)q);
  if (valid) {
    if (found < 0) flags[0] = 1;
    else cls[q] = (uint16_t)found;
  }
}
__global__ __launch_bounds__(kBlock) void k_cls_publish(int64_t nc, const double *__restrict__ inv,
                                                        const int *__restrict__ owner, const uint16_t *__restrict__ cls,
                                                        double *__restrict__ table) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / 36;
  if (q >= nc) return;
  if (owner[cls[q]] == (int)q) table[36 * (int64_t)cls[q] + (t - 36 * q)] = inv[t];
}
__global__ __launch_bounds__(kBlock) void k_cls_verify(int64_t nc, const double *__restrict__ inv,
                                                       const uint16_t *__restrict__ cls,
                                                       const double *__restrict__ table, int *__restrict__ flags) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q >= nc) return;
  const double *a = inv + 36 * q, *b = table + 36 * (int64_t)cls[q];
  double scale = 0.0, diff = 0.0;
#pragma unroll
  for (int e = 0; e < 36; ++e) {
    scale = fmax(scale, fabs(a[e]));
    diff = fmax(diff, fabs(a[e] - b[e]));
  }
  if (!(diff <= 1e-10 * scale)) flags[0] = 1;
}
// r_v <- r_v - y_v on the rows of the nodes that stay (the load the eliminated nodes pass on to them); the rows of the
// eliminated nodes keep their own right-hand side b_c for the back-substitution at the end
template <typename VT>
__global__ __launch_bounds__(kBlock) void k_condense_subtract(int64_t N, const uint8_t *__restrict__ cflag,
                                                              const VT *__restrict__ y, VT *__restrict__ r) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= 6 * N) return;
  if (!cflag[t / 6]) r[t] = (VT)((double)r[t] - (double)y[t]);
}
// x_c = K_cc^-1 (b_c - y_c): the eliminated nodes' displacements once the others are known (y = K [x_v ; 0])
template <typename VT>
__global__ __launch_bounds__(kBlock) void k_condense_backsubst(int64_t nc, const int32_t *__restrict__ cnodes,
                                                               const double *__restrict__ inv,
                                                               const uint16_t *__restrict__ cls,
                                                               const VT *__restrict__ b,
                                                               const VT *__restrict__ y, VT *__restrict__ x) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t q = t / 6;
  const int k = (int)(t - 6 * q);
  if (q >= nc) return;
  const int64_t i = cnodes[q];
  const double *A = inv + 36 * (cls ? (int64_t)cls[q] : q);
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < 6; ++j) acc += A[6 * k + j] * ((double)b[6 * i + j] - (double)y[6 * i + j]);
  x[6 * i + k] = (VT)acc;
}

// p = D^-1 r + P Z (y_c + y_t) + beta p, plus the end-of-iteration scalar bookkeeping (as k_pcg_direction).
// One workgroup per tile: aggregate, centre and the two rigid motions are wave-uniform (scalar loads).
// (MULTI / LOCAL as in k_pcg_update_tile: what a handle cannot have is compiled out, with its registers)
template <typename PT, typename RT, int TM = 6, bool MULTI = true, bool LOCAL = true>
__global__ __launch_bounds__(kBlock) void k_pcg_direction_coarse(const int32_t *__restrict__ tile_start,
                                                                 const RT *__restrict__ r,
                                                                 const float *__restrict__ dinv32,
                                                                 const double *__restrict__ xyz,
                                                                 const int32_t *__restrict__ agg_of_tile,
                                                                 const double *__restrict__ cen,
                                                                 const double *__restrict__ yc,
                                                                 const double *__restrict__ yt /* may be null */,
                                                                 const uint8_t *__restrict__ fixedbits,
                                                                 PT *__restrict__ p, RT *__restrict__ x,
                                                                 const double *__restrict__ scal,
                                                                 double *__restrict__ scal_next,
                                                                 double *__restrict__ hist, int k,
                                                                 double *__restrict__ rc, int ncp,
                                                                 const int32_t *__restrict__ aggL_of_tile /* may be null */,
                                                                 const double *__restrict__ cenL,
                                                                 const double *__restrict__ ycL,
                                                                 const uint8_t *__restrict__ shared /* may be null */,
                                                                 double *__restrict__ rcL, int ncpL,
                                                                 const uint8_t *__restrict__ zero_rows /* may be null */,
                                                                 int cm = 6,
                                                                 const float *__restrict__ rel32 = nullptr /* Coarse::rel32 */) {
  if constexpr (!MULTI) shared = nullptr;
  if constexpr (!MULTI || !LOCAL) aggL_of_tile = nullptr;
  double old, rz_new, pap;
  scalar_read3(scal, S_RZ_OLD, S_RZ_NEW, S_PAP, old, rz_new, pap);
  const double beta = (old != 0.0) ? rz_new / old : 0.0;
  const double alpha = (pap != 0.0) ? old / pap : 0.0;      // the step k_pcg_update_tile took: x += alpha p_old here
  if (blockIdx.x == 1 || gridDim.x == 1)      // r_c was consumed by the coarse solve: clear it for the next restriction
    for (int e = threadIdx.x; e < ncp; e += blockDim.x) rc[e] = 0.0;
  if (aggL_of_tile && (blockIdx.x == 2 || gridDim.x <= 2))
    for (int e = threadIdx.x; e < ncpL; e += blockDim.x) rcL[e] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x < kWave) {
    const int s = threadIdx.x;
    double rr = 0.0;                                          // ||r||^2 slots in the tail of r_c (this wave alone
    for (int q = s; q < kSlots; q += kWave) rr += rc[ncp + q];   // reads and then clears the tail)
    rr = wave_sum(rr);
    if (s == 0) hist[k] = rr;
    for (int q = s; q < kSlots; q += kWave) {
      rc[ncp + q] = 0.0;
      rc[ncp + kSlots + q] = 0.0;
      scal_next[S_RZ_OLD * kSlots + q] = scal[S_RZ_NEW * kSlots + q];
      scal_next[S_RZ_NEW * kSlots + q] = 0.0;
      scal_next[S_PAP * kSlots + q] = 0.0;
    }
  }
  const int t = blockIdx.x;
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int a = agg_of_tile[t];
  const double *y = yc + cm * a;
  double U0 = y[0], U1 = y[1], U2 = y[2], W0 = y[3], W1 = y[4], W2 = y[5];
  double T[TM];
#pragma unroll
  for (int k = 0; k < TM; ++k) T[k] = 0.0;
  if (yt) {   // the tile's modes use the same reference point, so the two rigid motions just add ...
    const double *q = yt + TM * (size_t)t;
#pragma unroll
    for (int k = 0; k < TM; ++k) T[k] = q[k];
    if (!shared) {
      U0 += T[0]; U1 += T[1]; U2 += T[2]; W0 += T[3]; W1 += T[4]; W2 += T[5];
    }
  }
  double ED[6] = {0, 0, 0, 0, 0, 0};   // the aggregate's uniform strains (12-mode dense level)
  if constexpr (TM == 12) {
    if (cm == 12) {
#pragma unroll
      for (int k = 0; k < 6; ++k) ED[k] = y[6 + k];
      if (!shared) {            // one GPU: same reference point, same nodes - they just add to the tile's
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          T[6 + k] += ED[k];
          ED[k] = 0.0;
        }
      }
    }
  }
  const bool own_t = yt && shared;   // ... except on several GPUs, where nodes shared with other ranks are left out
  const double c0 = cen[3 * a], c1 = cen[3 * a + 1], c2 = cen[3 * a + 2];
  double L[6] = {0, 0, 0, 0, 0, 0}, l0 = 0.0, l1 = 0.0, l2 = 0.0;     // rank-local dense level (multi-GPU)
  if (aggL_of_tile) {
    const int aL = aggL_of_tile[t];
#pragma unroll
    for (int q = 0; q < 6; ++q) L[q] = ycL[6 * aL + q];
    l0 = cenL[3 * aL];
    l1 = cenL[3 * aL + 1];
    l2 = cenL[3 * aL + 2];
  }
  for (int64_t i = n0 + threadIdx.x; i < n1; i += blockDim.x) {
    // eliminated node: not an unknown of this CG (its row of p is rewritten by the first pass of every product, which
    // takes the old content as zero: k_spmv_tile<.., kEndsCondensedSolve>)
    if (zero_rows && zero_rows[i]) continue;
    double rx, ry, rz;
    if (rel32) {
      rx = (double)rel32[3 * i];
      ry = (double)rel32[3 * i + 1];
      rz = (double)rel32[3 * i + 2];
    } else {
      rx = xyz[3 * i] - c0;
      ry = xyz[3 * i + 1] - c1;
      rz = xyz[3 * i + 2] - c2;
    }
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
    if (aggL_of_tile && !(shared && shared[i])) {
      const double sx = xyz[3 * i] - l0, sy = xyz[3 * i + 1] - l1, sz = xyz[3 * i + 2] - l2;
      zc[0] += L[0] + (L[4] * sz - L[5] * sy);
      zc[1] += L[1] + (L[5] * sx - L[3] * sz);
      zc[2] += L[2] + (L[3] * sy - L[4] * sx);
      zc[3] += L[3];
      zc[4] += L[4];
      zc[5] += L[5];
    }
    const unsigned fb = fixedbits[i];
    const float2 *d2 = reinterpret_cast<const float2 *>(dinv32 + 6 * i);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const double2 rr = load_pair(r, 3 * i + q);
      const float2 dd = d2[q];
      double2 pp = load_pair(p, 3 * i + q);
      double2 xx = load_pair(x, 3 * i + q);
      xx.x += alpha * pp.x;
      xx.y += alpha * pp.y;
      store_pair(x, 3 * i + q, xx);
      const double z0 = dd.x * rr.x + (((fb >> (2 * q)) & 1u) ? 0.0 : zc[2 * q]);
      const double z1 = dd.y * rr.y + (((fb >> (2 * q + 1)) & 1u) ? 0.0 : zc[2 * q + 1]);
      pp.x = z0 + beta * pp.x;
      pp.y = z1 + beta * pp.y;
      store_pair(p, 3 * i + q, pp);
    }
  }
}

// The same update with a flat mapping (no rank-local level): one lane per PAIR of vector entries (three lanes per
// node), so a wave's loads and stores of r, p, x are 1 KB contiguous (the per-tile kernel strides lanes by 48 bytes and
// leaves the last wave of a 152-node tile at 24 of 64 lanes); the tile's and the aggregate's coefficients come through
// the caches, the same address for almost every lane of a wave.  50^3 Octet: iteration 99.4 -> 97.0 us.
// (All six components are formed and the lane's pair selected afterwards: an if / else-if chain over the pair index with
// the strain terms inside was miscompiled by hipcc 7.2 for TM = 12 - components 4, 5 came out as 0 / garbage.)
template <typename PT, typename RT, int TM>
__global__ __launch_bounds__(kBlock) void k_pcg_direction_flat(int64_t N, const int32_t *__restrict__ tile_of_node,
                                                               const RT *__restrict__ r,
                                                               const float *__restrict__ dinv32,
                                                               const double *__restrict__ xyz,
                                                               const int32_t *__restrict__ agg_of_tile,
                                                               const double *__restrict__ cen,
                                                               const double *__restrict__ yc,
                                                               const double *__restrict__ yt /* may be null */,
                                                               const uint8_t *__restrict__ fixedbits,
                                                               PT *__restrict__ p, RT *__restrict__ x,
                                                               const double *__restrict__ scal,
                                                               double *__restrict__ scal_next,
                                                               double *__restrict__ hist, int k,
                                                               double *__restrict__ rc, int ncp,
                                                               const int32_t *__restrict__ keep /* may be null */,
                                                               int cm, const uint8_t *__restrict__ shared /* may be null */,
                                                               const float *__restrict__ rel32 = nullptr /* Coarse::rel32 */) {
  // keep: with node elimination (opts.condense) N counts the nodes that stay unknowns and keep[] lists them - the lanes
  // map onto those only (a flag test per node left half of the lanes of a BCC lattice idle: 80 us for 1.03 M kept nodes at
  // 100^3 against 26 us for 0.5 M nodes of the Octet lattice)
  double old, rz_new, pap;
  scalar_read3(scal, S_RZ_OLD, S_RZ_NEW, S_PAP, old, rz_new, pap);
  const double beta = (old != 0.0) ? rz_new / old : 0.0;
  const double alpha = (pap != 0.0) ? old / pap : 0.0;
  if (blockIdx.x == 1 || gridDim.x == 1)
    for (int e = threadIdx.x; e < ncp; e += blockDim.x) rc[e] = 0.0;
  if (blockIdx.x == 0 && threadIdx.x < kWave) {
    const int s = threadIdx.x;
    double rr = 0.0;
    for (int q = s; q < kSlots; q += kWave) rr += rc[ncp + q];
    rr = wave_sum(rr);
    if (s == 0) hist[k] = rr;
    for (int q = s; q < kSlots; q += kWave) {
      rc[ncp + q] = 0.0;
      rc[ncp + kSlots + q] = 0.0;
      scal_next[S_RZ_OLD * kSlots + q] = scal[S_RZ_NEW * kSlots + q];
      scal_next[S_RZ_NEW * kSlots + q] = 0.0;
      scal_next[S_PAP * kSlots + q] = 0.0;
    }
  }
  const int64_t lane_pair = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (lane_pair >= 3 * N) return;
  const int64_t slot = lane_pair / 3;
  const int q = (int)(lane_pair - 3 * slot);
  const int64_t i = keep ? (int64_t)keep[slot] : slot;
  const int64_t pair = 3 * i + q;
  const int t = tile_of_node[i];
  const int a = agg_of_tile[t];
  const double *y = yc + cm * a;
  double C[12];
#pragma unroll
  for (int m = 0; m < 6; ++m) C[m] = y[m];
#pragma unroll
  for (int m = 6; m < 12; ++m) C[m] = 0.0;
  if constexpr (TM == 12) {
    if (cm == 12) {
#pragma unroll
      for (int m = 6; m < 12; ++m) C[m] = y[m];
    }
  }
  if (yt && !(shared && shared[i])) {   // same reference point: the tile's coefficients add to the aggregate's (several
    const double *w = yt + TM * (size_t)t;   // GPUs: on this rank's own nodes only, as in k_pcg_direction_coarse)
#pragma unroll
    for (int m = 0; m < TM; ++m) C[m] += w[m];
  }
  double rx, ry, rz;
  if (rel32) {
    rx = (double)rel32[3 * i];
    ry = (double)rel32[3 * i + 1];
    rz = (double)rel32[3 * i + 2];
  } else {
    rx = xyz[3 * i] - cen[3 * a];
    ry = xyz[3 * i + 1] - cen[3 * a + 1];
    rz = xyz[3 * i + 2] - cen[3 * a + 2];
  }
  double zc[6] = {C[0] + (C[4] * rz - C[5] * ry), C[1] + (C[5] * rx - C[3] * rz), C[2] + (C[3] * ry - C[4] * rx),
                  C[3], C[4], C[5]};
  if constexpr (TM == 12) {
    zc[0] += C[6] * rx + 0.5 * (C[9] * ry + C[11] * rz);
    zc[1] += C[7] * ry + 0.5 * (C[9] * rx + C[10] * rz);
    zc[2] += C[8] * rz + 0.5 * (C[10] * ry + C[11] * rx);
  }
  double z0 = q == 0 ? zc[0] : (q == 1 ? zc[2] : zc[4]);
  double z1 = q == 0 ? zc[1] : (q == 1 ? zc[3] : zc[5]);
  const unsigned fb = fixedbits[i];
  const float2 dd = reinterpret_cast<const float2 *>(dinv32)[pair];
  const double2 rr = load_pair(r, pair);
  double2 pp = load_pair(p, pair);
  double2 xx = load_pair(x, pair);
  xx.x += alpha * pp.x;
  xx.y += alpha * pp.y;
  store_pair(x, pair, xx);
  z0 = dd.x * rr.x + (((fb >> (2 * q)) & 1u) ? 0.0 : z0);
  z1 = dd.y * rr.y + (((fb >> (2 * q + 1)) & 1u) ? 0.0 : z1);
  pp.x = z0 + beta * pp.x;
  pp.y = z1 + beta * pp.y;
  store_pair(p, pair, pp);
}

}  // namespace pl
