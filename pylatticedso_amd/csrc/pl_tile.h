// Node ordering + the LDS-tile K*x operator of libpylattice_hip (gfx950).
//
// Variant 3 of K*x ("one thread per strut, accumulate in LDS"):
//   * nodes are renumbered brick by brick (spatial_order) and cut into TILES of consecutive nodes (<= kTileMaxNodes);
//   * struts are renumbered by HOME tile = the lower of their two end tiles, so a tile's home struts are one
//     contiguous range whose 64-byte records and connectivity stream in fully coalesced; struts whose other end
//     lies in a higher tile are additionally listed in that tile's FOREIGN list (their record is read a second time,
//     from L2 when the two tiles run on the same XCD);
//   * one 256-thread workgroup per tile: every thread evaluates whole struts (both end forces), adds the ends that
//     belong to the tile into an LDS accumulator with ds_add_f64, and the tile's rows of y are written once,
//     coalesced, with the Dirichlet mask and the p.Ap partial dot product fused in.
// No global atomics; the only non-determinism is the order of the LDS adds.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "pl_kernels.h"
#include "pl_parallel.h"

namespace pl {

constexpr int kTileMaxNodes = 512;   // LDS accumulator: 512 nodes * 6 * 8 B = 24 KiB per workgroup

// Spatial order of the nodes: the bounding box is cut into cubic bricks holding ~nodes_per_brick nodes, bricks are
// walked x-slab by x-slab (an XCD's contiguous share of the tile range is then a slab of the lattice) and the nodes
// of one brick are contiguous.  perm[new] = old.  tile_start gets the brick boundaries (bricks larger than
// kTileMaxNodes are cut), ending with N.
struct BrickGrid {
  double lo[3] = {0, 0, 0};
  double side[3] = {1.0, 1.0, 1.0};   // per axis: the box is cut into equal bricks
  int64_t nb[3] = {1, 1, 1};
  int64_t na[3] = {0, 0, 0};          // aggregate grid the bricks were fitted to (0: none - coarse_setup chooses)
};

// If `global` is non-null it holds {lo[3], hi[3], node count} of the WHOLE lattice (multi-GPU: every rank must cut
// the same grid); otherwise the grid is derived from these nodes.
inline void spatial_order(const double *xyz, int64_t N, std::vector<int32_t> &perm, std::vector<int32_t> &tile_start,
                          double nodes_per_brick, std::vector<int64_t> &tile_brick, BrickGrid &grid,
                          const double *global = nullptr, int agg_max_dofs = 0) {
  int64_t *nbrick = grid.nb;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  int64_t Ntot = N;
  if (global) {
    for (int k = 0; k < 3; ++k) {
      lo[k] = global[k];
      hi[k] = global[3 + k];
    }
    Ntot = (int64_t)global[6];
  } else {
    for (int64_t i = 0; i < N; ++i)
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], xyz[3 * i + k]);
        hi[k] = std::max(hi[k], xyz[3 * i + k]);
      }
  }
  double vol = 1.0;
  int dims = 0;
  for (int k = 0; k < 3; ++k)
    if (hi[k] - lo[k] > 0) {
      vol *= hi[k] - lo[k];
      ++dims;
    }
  const double side = dims ? std::pow(vol * nodes_per_brick / (double)std::max<int64_t>(Ntot, 1), 1.0 / dims) : 1.0;
  // equal bricks per axis (the count that comes closest to the cubic target): a thin leftover layer of bricks at the far
  // faces would give the tile level and the dense level starved blocks there (measured: iteration counts jumping by
  // 15 % with the target size)
  for (int k = 0; k < 3; ++k) grid.lo[k] = lo[k];
  int64_t *nb = nbrick;
  double sk[3];
  for (int k = 0; k < 3; ++k) nb[k] = std::max<int64_t>(1, (int64_t)std::llround((hi[k] - lo[k]) / side));
  if (agg_max_dofs > 0 && dims > 0) {
    // Multi-level preconditioner: fit the bricks (tile level) INTO the aggregates of the dense level - equal aggregates,
    // each a whole number (>= 2 per axis) of equal bricks.  With independent grids 13 bricks per axis fall into 7 aggregates
    // as 2,2,2,2,2,2,1: measured 116 iterations at 50^3 Octet against 106 with 14 = 7 x 2.
    // Aggregate edge: the smallest one whose grid fits the dof budget.
    double a = 1.5 * side;
    int64_t na[3] = {1, 1, 1};
    for (int it = 0; it < 4000; ++it, a *= 1.01) {
      int64_t count = 1;
      for (int k = 0; k < 3; ++k) {
        na[k] = (hi[k] - lo[k]) > 0 ? std::max<int64_t>(1, (int64_t)std::llround((hi[k] - lo[k]) / a)) : 1;
        count *= na[k];
      }
      if (6 * count <= agg_max_dofs) break;
    }
    // bricks per aggregate and axis: the count that brings the brick edge closest to the target - near-cubic bricks
    // (splitting an aggregate 3 x 2 x 1 to hit the target volume exactly costs 30 % more iterations: measured); an
    // aggregate no larger than a brick or two leaves the two grids independent
    bool fits = true;
    int64_t m[3] = {1, 1, 1};
    for (int k = 0; k < 3; ++k)
      if (hi[k] - lo[k] > 0) {
        m[k] = (int64_t)std::llround((hi[k] - lo[k]) / (double)na[k] / side);
        if (m[k] < 2) fits = false;
      }
    if (fits)
      for (int k = 0; k < 3; ++k) {
        nb[k] = m[k] * na[k];
        grid.na[k] = na[k];
      }
  }
  for (int k = 0; k < 3; ++k) {
    sk[k] = (hi[k] - lo[k]) > 0 ? (hi[k] - lo[k]) / (double)nb[k] : 1.0;
    grid.side[k] = sk[k];
  }
  // key = position of the node's brick in the walk over the bricks; gkey = the brick's id in the grid (x, y, z
  // lexicographic: what coarse_setup maps to aggregates).  With bricks nested into aggregates the walk goes aggregate by
  // aggregate (all m0 x m1 x m2 bricks of one, then the next in z, y, x): the rows a tile gathers from its neighbours are
  // then still in the XCD's 4 MiB L2 when the neighbour runs - walked plane by plane, a y-z plane of bricks is 10 MB at
  // 100^3 Octet and the K*p fetched 1.19 x its algorithmic bytes (profiles/r03_c_pmc_s100.json).
  const bool nested = grid.na[0] > 0 && std::getenv("PL_BRICK_ORDER_PLANES") == nullptr;
  int64_t mpa[3] = {1, 1, 1};
  if (nested)
    for (int k = 0; k < 3; ++k) mpa[k] = std::max<int64_t>(1, nb[k] / std::max<int64_t>(1, grid.na[k]));
  std::vector<int64_t> key(N), gkey(N);
  parallel_for(N, [&](int64_t i0, int64_t i1, unsigned) {
    for (int64_t i = i0; i < i1; ++i) {
      int64_t c[3];
      for (int k = 0; k < 3; ++k)
        c[k] = std::max<int64_t>(0, std::min<int64_t>(nb[k] - 1, (int64_t)std::floor((xyz[3 * i + k] - lo[k]) / sk[k])));
      gkey[i] = (c[0] * nb[1] + c[1]) * nb[2] + c[2];
      if (nested) {
        const int64_t a[3] = {c[0] / mpa[0], c[1] / mpa[1], c[2] / mpa[2]};
        const int64_t l[3] = {c[0] - a[0] * mpa[0], c[1] - a[1] * mpa[1], c[2] - a[2] * mpa[2]};
        const int64_t nak[3] = {nb[0] / mpa[0], nb[1] / mpa[1], nb[2] / mpa[2]};
        key[i] = ((a[0] * nak[1] + a[1]) * nak[2] + a[2]) * (mpa[0] * mpa[1] * mpa[2]) + (l[0] * mpa[1] + l[1]) * mpa[2] + l[2];
      } else {
        key[i] = gkey[i];
      }
    }
  });
  const int64_t n_keys = nb[0] * nb[1] * nb[2];
  if (n_keys <= 4 * N + 1024) {   // stable counting sort by brick (the usual case: about N / nodes_per_brick bricks)
    std::vector<int64_t> start((size_t)n_keys + 1, 0);
    for (int64_t i = 0; i < N; ++i) start[key[i] + 1]++;
    for (int64_t q = 0; q < n_keys; ++q) start[q + 1] += start[q];
    std::vector<int32_t> sorted(N);
    for (int64_t i = 0; i < N; ++i) sorted[start[key[perm[i]]]++] = perm[i];
    perm.swap(sorted);
  } else {
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return key[a] < key[b]; });
  }
  tile_start.clear();
  tile_brick.clear();
  int64_t run0 = 0;
  for (int64_t i = 0; i <= N; ++i) {
    if (i == N || (i > 0 && key[perm[i]] != key[perm[i - 1]])) {
      // close run [run0, i): cut into equal pieces of at most kTileMaxNodes
      const int64_t len = i - run0;
      const int64_t pieces = (len + kTileMaxNodes - 1) / kTileMaxNodes;
      for (int64_t q = 0; q < pieces; ++q) {
        tile_start.push_back((int32_t)(run0 + q * len / pieces));
        tile_brick.push_back(gkey[perm[run0]]);
      }
      run0 = i;
    }
  }
  tile_start.push_back((int32_t)N);
}

// Tiles for an un-reordered numbering: plain chunks of 256 consecutive nodes.
inline void chunk_tiles(int64_t N, std::vector<int32_t> &tile_start, int chunk = 256) {
  tile_start.clear();
  for (int64_t i = 0; i < N; i += chunk) tile_start.push_back((int32_t)i);
  tile_start.push_back((int32_t)N);
}

template <typename T>
struct TBuf {
  T *p = nullptr;
  ~TBuf() { if (p) (void)hipFree(p); }
  hipError_t upload(const std::vector<T> &v) {
    if (p) (void)hipFree(p);
    p = nullptr;
    const size_t n = std::max<size_t>(1, v.size());
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T));
    if (e != hipSuccess) return e;
    if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
};

struct TilePlan {
  bool ready = false;
  int64_t n_tiles = 0;
  TBuf<int32_t> tile_start;     // [n_tiles+1] node range of each tile
  TBuf<int64_t> home_ptr;       // [n_tiles+1] strut range of each tile (struts are numbered by home tile)
  TBuf<int64_t> foreign_ptr;    // [n_tiles+1]
  TBuf<int32_t> foreign_idx;    // strut ids
  int64_t n_foreign = 0;
  int max_nodes = 0;            // largest tile: sizes the LDS accumulator of the K*p launch
};

// Computes the strut permutation (bperm[new] = old) that numbers struts by home tile and, inside a tile, so that
// consecutive struts touch different nodes (LDS atomics of one wave instruction then hit distinct addresses).
inline void tile_strut_order(const std::vector<int32_t> &conn, int64_t N, int64_t B,
                             const std::vector<int32_t> &tile_start, std::vector<int32_t> &tile_of,
                             std::vector<int32_t> &bperm) {
  const int64_t T = (int64_t)tile_start.size() - 1;
  tile_of.assign(N, 0);
  for (int64_t t = 0; t < T; ++t)
    for (int32_t i = tile_start[t]; i < tile_start[t + 1]; ++i) tile_of[i] = (int32_t)t;
  std::vector<int32_t> home(B), occ(B);
  std::vector<int32_t> cnt(N, 0);
  for (int64_t b = 0; b < B; ++b) {
    const int32_t a = conn[2 * b], d = conn[2 * b + 1];
    home[b] = std::min(tile_of[a], tile_of[d]);
    const int32_t pivot = tile_of[d] == home[b] ? d : a;   // an end that lies in the home tile
    occ[b] = cnt[pivot]++;
  }
  // order by (home tile, occurrence, original index): counting sort into the home tiles, then every tile's short run
  // sorted on its own - the same permutation as one stable sort of all struts, in O(B) and in parallel
  std::vector<int64_t> hp(T + 1, 0);
  for (int64_t b = 0; b < B; ++b) hp[home[b] + 1]++;
  for (int64_t t = 0; t < T; ++t) hp[t + 1] += hp[t];
  bperm.resize(B);
  {
    std::vector<int64_t> fill(hp.begin(), hp.end() - 1);
    for (int64_t b = 0; b < B; ++b) bperm[fill[home[b]]++] = (int32_t)b;      // ascending original index inside a tile
  }
  parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
    for (int64_t t = t0; t < t1; ++t)
      std::stable_sort(bperm.begin() + hp[t], bperm.begin() + hp[t + 1],
                       [&](int32_t l, int32_t r) { return occ[l] < occ[r]; });
  }, 16);
}

// conn must already be in the NEW strut numbering.
inline int build_tile_plan(TilePlan &plan, const std::vector<int32_t> &conn, int64_t N, int64_t B,
                           const std::vector<int32_t> &tile_start, const std::vector<int32_t> &tile_of) {
  const int64_t T = (int64_t)tile_start.size() - 1;
  std::vector<int64_t> home_ptr(T + 1, 0), foreign_ptr(T + 1, 0);
  for (int64_t b = 0; b < B; ++b) {
    const int32_t ta = tile_of[conn[2 * b]], tb = tile_of[conn[2 * b + 1]];
    home_ptr[std::min(ta, tb) + 1]++;
    if (ta != tb) foreign_ptr[std::max(ta, tb) + 1]++;
  }
  for (int64_t t = 0; t < T; ++t) {
    home_ptr[t + 1] += home_ptr[t];
    foreign_ptr[t + 1] += foreign_ptr[t];
  }
  std::vector<int32_t> foreign_idx((size_t)foreign_ptr[T]);
  std::vector<int64_t> fill(foreign_ptr.begin(), foreign_ptr.end() - 1);
  int64_t prev_home = -1;
  for (int64_t b = 0; b < B; ++b) {
    const int32_t ta = tile_of[conn[2 * b]], tb = tile_of[conn[2 * b + 1]];
    const int64_t h = std::min(ta, tb);
    if (h < prev_home) return 1;   // struts are not numbered by home tile
    prev_home = h;
    if (ta != tb) foreign_idx[fill[std::max(ta, tb)]++] = (int32_t)b;
  }
  int max_nodes = 0;
  for (int64_t t = 0; t < T; ++t) {
    if (tile_start[t + 1] - tile_start[t] > kTileMaxNodes) return 2;
    max_nodes = std::max(max_nodes, (int)(tile_start[t + 1] - tile_start[t]));
  }
  plan.max_nodes = max_nodes;
  plan.n_tiles = T;
  plan.n_foreign = foreign_ptr[T];
  if (plan.tile_start.upload(tile_start) != hipSuccess) return 3;
  if (plan.home_ptr.upload(home_ptr) != hipSuccess) return 3;
  if (plan.foreign_ptr.upload(foreign_ptr) != hipSuccess) return 3;
  if (plan.foreign_idx.upload(foreign_idx) != hipSuccess) return 3;
  plan.ready = true;
  return 0;
}

// The LDS accumulator is component-major, ys[k][node] (pitch = stride): the 64 lanes of a ds_add_f64 then spread over
// node mod 16 bank pairs instead of the (6 node + k) mod 16 = 8 classes of a node-major layout - half the bank
// conflicts on the instruction this kernel issues most (12 per strut visit).
__device__ __forceinline__ void lds_add6(double *dst, int stride, V3 f, V3 m) {
  unsafeAtomicAdd(dst, f.x);
  unsafeAtomicAdd(dst + stride, f.y);
  unsafeAtomicAdd(dst + 2 * stride, f.z);
  unsafeAtomicAdd(dst + 3 * stride, m.x);
  unsafeAtomicAdd(dst + 4 * stride, m.y);
  unsafeAtomicAdd(dst + 5 * stride, m.z);
}

// Where a strut's record comes from (template parameter REC of the tile kernel):
//   kRecAoS      rec[b], 64 B per strut;
//   kRecPalette  the palette table through a 2-byte id (pl_palette.h): periodic lattices, the table stays in L2;
//   kRecCompact  5 stiffness scalars per strut (40 B) and d = x_B - x_A recomputed from the node coordinates, which
//                are gathered like the x rows and mostly hit L2: the streaming path (graded / optimised lattices) is
//                bound by HBM bytes, and this takes a third off the record stream.
enum { kRecAoS = 0, kRecPalette = 1, kRecCompact = 2 };
struct __attribute__((aligned(8))) Rec5 {
  double a, c, e1, e2, e3;
};

// ENDS: which strut ends are accumulated - kEndsAll, or (node elimination, pl_coarse.h) only the ends at condensed nodes
// / only the others, told apart by the byte flag cflag[node].
// kEndsCondensedSolve: as kEndsCondensed, and the tile then puts v = -K_cc^-1 (accumulated row) into row `node` of y for
// each of its condensed nodes - the equilibrium position under the other nodes' x.  Called with y = x, this is the whole
// first half of the Schur-complement product in one launch (pl_coarse.h): x of a condensed end counts as zero on the way
// in, so nobody reads the rows another tile is writing.
enum { kEndsAll = 0, kEndsCondensed = 1, kEndsOthers = 2, kEndsCondensedSolve = 3 };
struct CondSolve {            // K_cc^-1 of the condensed nodes (pl_coarse.h)
  const double *inv = nullptr;     // 6 x 6 blocks: the class table, or one block per condensed node
  const int32_t *base = nullptr;   // node -> offset of its block in inv (doubles); < 0: not a condensed node
  const uint8_t *cend = nullptr;   // strut -> bit 0 / 1: end A / B is a condensed node (every pass of the condensed operator:
                                   // fetched one visit ahead with conn, instead of two dependent byte loads per visit)
};
// (c = conn2[b] and pid = pal[b] come from the caller, which fetches them one visit ahead: tile_struts)
// `after_loads()` runs once this visit's gathers have been requested and before anything waits for them: the caller's
// fetch of the NEXT visit's indices goes there - vector loads return in order, so requested earlier it would stand
// between this visit's gathers and the arithmetic that needs them.
template <int REC, int ENDS, typename VT, typename AfterLoads>
__device__ __forceinline__ void tile_strut(int64_t b, const int2 c, const unsigned pid, int n0, int n1,
                                           const Record *__restrict__ rec,
                                           const double *__restrict__ xyz, const uint8_t *__restrict__ cflag,
                                           const VT *__restrict__ x, double *ys, int stride, AfterLoads &&after_loads) {
  constexpr bool kToCondensed = ENDS == kEndsCondensed || ENDS == kEndsCondensedSolve;
  bool takeB = c.y >= n0 && c.y < n1, takeA = c.x >= n0 && c.x < n1;
  // (pid carries the strut's condensed-end bits above the palette id: CondSolve::cend)
  const bool cA = ENDS != kEndsAll && ((pid >> 16) & 1u), cB = ENDS != kEndsAll && ((pid >> 17) & 1u);
  if (ENDS != kEndsAll) {
    takeB = takeB && (cB == kToCondensed);
    takeA = takeA && (cA == kToCondensed);
    // a pass over one kind of strut ends: the tile that does not own an end of that kind has nothing to take from this
    // visit (bipartite lattices: every strut has one end of each kind, so 20-25 % of the visits of either pass) - known
    // from the indices alone, before any gather is requested
    if (!(takeA || takeB)) {
      after_loads();
      return;
    }
  }
  Record r;
  if (REC == kRecCompact) {
    const Rec5 q = reinterpret_cast<const Rec5 *>(rec)[b];
    const double *pa = xyz + 3 * (int64_t)c.x, *pb = xyz + 3 * (int64_t)c.y;
    r.a = q.a; r.c = q.c; r.e1 = q.e1; r.e2 = q.e2; r.e3 = q.e3;
    r.dx = pb[0] - pa[0]; r.dy = pb[1] - pa[1]; r.dz = pb[2] - pa[2];
  } else {
    r = (REC == kRecPalette) ? load_record(rec, pid & 0xFFFFu) : load_record(rec, b);
  }
  V3 uA = {0, 0, 0}, tA = {0, 0, 0}, uB = {0, 0, 0}, tB = {0, 0, 0}, F, M;
  // a condensed end's own row is being rewritten by its tile in the fused first pass: it counts as zero and is not read
  if (!(ENDS == kEndsCondensedSolve && cA)) load6(x + 6 * (int64_t)c.x, uA, tA);
  if (!(ENDS == kEndsCondensedSolve && cB)) load6(x + 6 * (int64_t)c.y, uB, tB);
  after_loads();
  tip_force(r, uA, tA, uB, tB, F, M);
  if (takeB) lds_add6(ys + (c.y - n0), stride, F, M);
  if (takeA) {
    const V3 d = {r.dx, r.dy, r.dz};
    lds_add6(ys + (c.x - n0), stride, (-1.0) * F, (-1.0) * M - cross(d, F));
  }
}

// Workgroup size of the tile K*p.  Measured on the 50^3 Octet (256-node tiles): 128 / 256 / 384 / 512 / 640 / 768 / 1024
// threads -> 45.9 / 40.6 / 39.7 / 36.2 / 47.8 / 42.8 / 55.0 us: with 512 a tile's ~1900 strut visits are 3-4 per thread, so a
// workgroup lives half as long while 4 of them still fit a CU.  (End of round 2, 152-node tiles of ~1 080 visits:
// 384 / 448 / 512 / 576 / 640 threads -> 39.4 / 39.2 / 37.9 / 43.7 / 47.0 us.)
#ifndef PL_TILE_BLOCK
#define PL_TILE_BLOCK 512
#endif
constexpr int kTileBlock = PL_TILE_BLOCK;
// -DPL_TILE_STAMPS: thread 0 of the first 4096 workgroups of the tile K*p records the constant-rate clock (100 MHz) at five
// points of its life; pl_debug_tile_stamps() reads them (tools/experiments/tile_stamps.py).  Experiment builds only.
#ifdef PL_TILE_STAMPS
__device__ unsigned long long g_tile_stamps[8 * 4096];
#define PL_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_tile_stamps[8 * blockIdx.x + (k)] = wall_clock64();   \
  } while (0)
#else
#define PL_STAMP(k) do { } while (0)
#endif
// VT = storage type of x and y (double, or float for the fp32 solver modes: the strut forces are still evaluated
// and accumulated in fp64 - the forces on a node nearly cancel for the smooth fields a solve is made of, so rounding
// them to fp32 before the sum would cost cond(K) * 6e-8, rounding the stored result costs 6e-8).
template <bool MASK, bool DOT, int REC, typename VT, int ENDS = kEndsAll>
__global__ __launch_bounds__(kTileBlock) void k_spmv_tile(const int32_t *__restrict__ tile_start,
                                                      const int64_t *__restrict__ home_ptr,
                                                      const int64_t *__restrict__ foreign_ptr,
                                                      const int32_t *__restrict__ foreign_idx,
                                                      const int2 *__restrict__ conn2, const Record *__restrict__ rec,
                                                      const uint16_t *__restrict__ pal,
                                                      const double *__restrict__ xyz,
                                                      const uint8_t *__restrict__ fixedbits,
                                                      const VT *__restrict__ x, VT *__restrict__ y,
                                                      double *__restrict__ dot_out, int stride,
                                                      const uint8_t *__restrict__ cflag = nullptr,
                                                      CondSolve cs = CondSolve(),
                                                      const int32_t *__restrict__ tile_list = nullptr) {
  extern __shared__ double ys[];             // [6][stride], stride >= nodes of the largest tile (launch_tile_spmv)
  __shared__ double red[kTileBlock / kWave];
  __shared__ int32_t sbase[ENDS == kEndsCondensedSolve ? kTileMaxNodes : 1];
  // tile_list: a launch over a subset of the tiles (multi-GPU overlap: the tiles that own interface rows first, the
  // exchange under the others) - ascending tile numbers, so the XCD mapping keeps its contiguous eighths
  PL_STAMP(0);
  unsigned t = xcd_block(blockIdx.x, gridDim.x);
  if (tile_list) t = (unsigned)tile_list[t];
  const int n0 = tile_start[t], n1 = tile_start[t + 1];
  const int nn = n1 - n0;
  for (int i = threadIdx.x; i < 6 * stride; i += kTileBlock) ys[i] = 0.0;
  if (ENDS == kEndsCondensedSolve)           // fetched now, needed after the strut loops: no dependent load in the tail
    for (int i = threadIdx.x; i < nn; i += kTileBlock) sbase[i] = cs.base[n0 + i];
  __syncthreads();
  PL_STAMP(1);
  // A thread makes 2-3 visits per tile, each a chain of two memory hops (conn -> x rows, palette id -> record) before
  // the arithmetic.  The first hop of the NEXT visit (10 bytes) is requested right AFTER the current visit's second hop
  // (tile_strut's after_loads), so only a thread's first visit pays both (SQ counters: the waves of this kernel spent 52 %
  // of their life in s_waitcnt).  (Requested before it - vector loads return in order - K*p is 1 us slower, 39.0 against
  // 38.0 us; requesting the FIRST visit's indices above the clearing of the accumulator takes that microsecond back again.)
  const int64_t h0 = home_ptr[t], h1 = home_ptr[t + 1];
  const int64_t f0 = foreign_ptr[t], f1 = foreign_ptr[t + 1];
  int64_t b = h0 + threadIdx.x, kf = f0 + threadIdx.x;
  bool home = b < h1;
  if (!home && kf < f1) b = foreign_idx[kf];
  bool live = home || kf < f1;
  int2 cn = {0, 0};
  unsigned pid = 0;
  if (live) {
    cn = conn2[b];
    if (REC == kRecPalette) pid = pal[b];
    if (ENDS != kEndsAll) pid |= (unsigned)cs.cend[b] << 16;
  }
  PL_STAMP(2);
  while (live) {
    // next visit: the following home strut, else this thread's first / next foreign one
    int64_t bn = b;
    bool home_n = false, live_n = false;
    if (home && b + kTileBlock < h1) {
      bn = b + kTileBlock;
      home_n = live_n = true;
    } else {
      if (!home) kf += kTileBlock;
      if (kf < f1) {
        bn = foreign_idx[kf];
        live_n = true;
      }
    }
    int2 cn_n = {0, 0};
    unsigned pid_n = 0;
    tile_strut<REC, ENDS, VT>(b, cn, pid, n0, n1, rec, xyz, cflag, x, ys, stride, [&]() {
      if (live_n) {
        cn_n = conn2[bn];
        if (REC == kRecPalette) pid_n = pal[bn];
        if (ENDS != kEndsAll) pid_n |= (unsigned)cs.cend[bn] << 16;
      }
    });
    b = bn;
    cn = cn_n;
    pid = pid_n;
    home = home_n;
    live = live_n;
  }
  PL_STAMP(3);
  __syncthreads();
  PL_STAMP(4);
  if (ENDS == kEndsCondensedSolve) {
    for (int i = threadIdx.x; i < nn * 6; i += kTileBlock) {
      const int node = i / 6, k = i - 6 * node;
      const int32_t b0 = sbase[node];
      if (b0 < 0) continue;
      const double *A = cs.inv + b0 + 6 * k;
      double v = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) v += A[j] * ys[j * stride + node];
      y[6 * (int64_t)(n0 + node) + k] = (VT)(-v);
    }
    return;
  }
  double acc = 0.0;
  const int64_t pair0 = 3 * (int64_t)n0;
  for (int i = threadIdx.x; i < nn * 3; i += kTileBlock) {
    const int node = i / 3, part = i - 3 * node;
    if (ENDS != kEndsAll && ((cflag[n0 + node] != 0) != (ENDS == kEndsCondensed))) continue;   // rows of the other kind
    double2 v = {ys[(2 * part) * stride + node], ys[(2 * part + 1) * stride + node]};
    if (MASK) {
      const unsigned fb = fixedbits[n0 + node] >> (2 * part);
      if (fb & 1u) v.x = 0.0;
      if (fb & 2u) v.y = 0.0;
    }
    store_pair(y, pair0 + i, v);
    if (DOT) {
      const double2 xv = load_pair(x, pair0 + i);
      acc += xv.x * v.x + xv.y * v.y;
    }
  }
  if (DOT) {
    double s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = 0.0;
    if (threadIdx.x == 0)
      for (int q = 0; q < kTileBlock / kWave; ++q) s += red[q];
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
  }
  PL_STAMP(5);
}

// (Round 3, tried and dropped: PERSISTENT workgroups - each walks several tiles of its XCD's eighth, the LDS accumulator
// cleared once (the epilogue zeroes what it reads), the next tile's first visit requested before the barrier and the
// epilogue of the current one, the fused dot reduced once per workgroup, 32-bit strut ids to stay at 64 VGPRs / 8 waves.
// In-kernel clock stamps of the kernel above (tools/experiments/tile_stamps.py, 50^3 Octet, palette) had shown a workgroup
// living 9.3 us of which 1.0 us clear + barrier, 0.8 us until the first visit's indices are requested, 5.2 us strut loop,
// 1.2 us barrier + epilogue.  Measured: 848 workgroups of 4 tiles each 49.3 us, 1 024 of 3-4 tiles 43.3 us, 2 048 of 1-2
// tiles 41.7 us against 40.8 us of one tile per workgroup, with or without the prefetch: the per-workgroup overheads are
// already hidden behind the other resident workgroups' strut loops, and workgroups in lockstep hide them worse.)
// pal != nullptr: `rec` is the palette table and pal[b] the strut's entry; xyz != nullptr: `rec` is the compact
// 5-scalar table (Rec5) and the strut vectors come from the node coordinates.  ends / cflag: see tile_strut.
template <typename VT>
inline void launch_tile_spmv(const TilePlan &plan, const int32_t *conn, const Record *rec, const uint16_t *pal,
                             const uint8_t *fixedbits, const VT *x, VT *y, double *dot_dev, hipStream_t s,
                             const double *xyz = nullptr, int ends = kEndsAll, const uint8_t *cflag = nullptr,
                             CondSolve cs = CondSolve(), const int32_t *tile_list = nullptr, int64_t n_list = 0) {
  if (tile_list && n_list <= 0) return;
  const int64_t n_units = tile_list ? n_list : plan.n_tiles;
  const dim3 g((unsigned)n_units), blk(kTileBlock);
  const int stride = plan.max_nodes | 1;                             // odd pitch of the component-major accumulator
  const size_t lds = (size_t)stride * 6 * sizeof(double);            // sized by the largest tile: more resident waves
  const int2 *conn2 = reinterpret_cast<const int2 *>(conn);
#define PL_T(M, D, P, E)                                                                                          \
  hipLaunchKernelGGL((k_spmv_tile<M, D, P, VT, E>), g, blk, lds, s, plan.tile_start.p, plan.home_ptr.p,           \
                     plan.foreign_ptr.p, plan.foreign_idx.p, conn2, rec, pal, xyz, fixedbits, x, y, dot_dev, stride, cflag, cs, \
                     tile_list)
#define PL_TT(P, E)                                           \
  do {                                                        \
    if (fixedbits && dot_dev) PL_T(true, true, P, E);         \
    else if (fixedbits) PL_T(true, false, P, E);              \
    else if (dot_dev) PL_T(false, true, P, E);                \
    else PL_T(false, false, P, E);                            \
  } while (0)
#define PL_TE(P)                                   \
  do {                                             \
    if (ends == kEndsCondensed) PL_TT(P, kEndsCondensed); \
    else if (ends == kEndsCondensedSolve) PL_T(false, false, P, kEndsCondensedSolve); \
    else if (ends == kEndsOthers) PL_TT(P, kEndsOthers);  \
    else PL_TT(P, kEndsAll);                       \
  } while (0)
  if (pal) PL_TE(kRecPalette);
  else if (xyz) PL_TE(kRecCompact);
  else PL_TE(kRecAoS);
#undef PL_TE
#undef PL_TT
#undef PL_T
}

}  // namespace pl
