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
#include <cstring>
#include <mutex>
#include <numeric>
#include <type_traits>
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

struct TileDesc {
  int32_t n0, n1;                   // node range
  int32_t n_int, n_cross, n_ch;     // visits: interior, crossing (of which the first n_ch are home struts, the rest foreign)
  int32_t pad;
  int64_t v0, c0;                   // first visit (interior, then crossing at v0 + n_int), first entry of vother
  int64_t h0, f0;                   // first home strut (visits 0 .. n_int + n_ch - 1 are struts h0, h0 + 1, ...), first
                                    // entry of foreign_idx (crossing visit kc >= n_ch is strut foreign_idx[f0 + kc - n_ch])
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
  // Visit lists of the LDS-resident K*p (k_spmv_tile_lds).  Per tile: first the INTERIOR visits (struts with both ends in
  // the tile; 32-bit word = local row of end A | local row of end B << 10), then the CROSSING ones (home struts whose
  // other end lies in a higher tile, and the foreign struts; word = local row of the tile's own end | (own end is B) << 10,
  // vother = node id of the other end).  vstrut = the strut of a visit (palette id / condensed-end bits per visit are
  // derived from it at assembly time); tdesc = everything a workgroup needs to know about its tile in one 32-byte read.
  bool vis_ready = false;
  int64_t n_visits = 0, n_cross = 0;
  TBuf<TileDesc> tdesc;         // [n_tiles]
  TBuf<uint32_t> vloc;          // [n_visits]
  TBuf<int32_t> vstrut;         // [n_visits]
  TBuf<int32_t> vother;         // [n_cross]
  // direction palette: the distinct end-to-end vectors of the struts, compared bit for bit (<= kDirMax of them, else
  // n_dir = 0): bits 21..28 of vloc hold a strut's entry, so that the streaming form of the LDS-resident kernel reads
  // d = x_B - x_A from an LDS table and only five stiffness scalars per strut from HBM
  int n_dir = 0;
  TBuf<double> dir_table;       // [n_dir][4]
};
constexpr int kDirMax = 256;
constexpr int kVisRowBits = 10;                       // local rows of a visit word: < 1024 (a tile has <= 512 nodes)

// Quantised end-to-end vector of a strut as one integer (1/4096 of a length unit per axis): the strut's DIRECTION class.
inline uint64_t strut_dir_code(const double *xyz, int32_t a, int32_t d) {
  uint64_t dir = 0;
  for (int k = 0; k < 3; ++k) {
    const int64_t q = (int64_t)std::llround((xyz[3 * (size_t)d + k] - xyz[3 * (size_t)a + k]) * 4096.0) + (1 << 19);
    dir = (dir << 20) | (uint64_t)(q & 0xFFFFF);
  }
  return dir;
}
// Sort key of a strut inside its tile [n0, n1): interior struts (both ends in the tile) first, then the crossing ones;
// inside each group by direction class, then by the local row of the end that lies in the tile (end A for interior
// struts).  In a periodic lattice the struts of one direction join row i to row i + const: the 64 lanes of a wave then
// touch (nearly) consecutive rows on both ends - few LDS bank conflicts in the tile kernels, contiguous gathers - and
// share one record.
struct StrutKey {
  uint64_t hi, lo;      // hi: crossing flag | direction; lo: row
  bool operator<(const StrutKey &o) const { return hi != o.hi ? hi < o.hi : lo < o.lo; }
};
inline StrutKey strut_key(const double *xyz, int32_t a, int32_t d, int32_t n0, int32_t n1) {
  const bool inA = a >= n0 && a < n1, inB = d >= n0 && d < n1;
  const uint64_t dir = xyz ? strut_dir_code(xyz, a, d) : 0;
  if (inA && inB) return {dir, (uint64_t)(a - n0)};
  return {(1ull << 63) | dir, (uint64_t)((inB ? d : a) - n0)};
}

// Computes the strut permutation (bperm[new] = old) that numbers struts by home tile and, inside a tile, by strut_key
// (xyz = node coordinates in the numbering of conn; without them: interior first, by row).
inline void tile_strut_order(const std::vector<int32_t> &conn, int64_t N, int64_t B,
                             const std::vector<int32_t> &tile_start, std::vector<int32_t> &tile_of,
                             std::vector<int32_t> &bperm, const double *xyz = nullptr) {
  const int64_t T = (int64_t)tile_start.size() - 1;
  tile_of.assign(N, 0);
  for (int64_t t = 0; t < T; ++t)
    for (int32_t i = tile_start[t]; i < tile_start[t + 1]; ++i) tile_of[i] = (int32_t)t;
  std::vector<int32_t> home(B);
  parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
    for (int64_t b = b0; b < b1; ++b) home[b] = std::min(tile_of[conn[2 * b]], tile_of[conn[2 * b + 1]]);
  }, 1 << 16);
  // counting sort into the home tiles, then every tile's short run sorted on its own
  std::vector<int64_t> hp(T + 1, 0);
  for (int64_t b = 0; b < B; ++b) hp[home[b] + 1]++;
  for (int64_t t = 0; t < T; ++t) hp[t + 1] += hp[t];
  bperm.resize(B);
  {
    std::vector<int64_t> fill(hp.begin(), hp.end() - 1);
    for (int64_t b = 0; b < B; ++b) bperm[fill[home[b]]++] = (int32_t)b;      // ascending original index inside a tile
  }
  parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
    std::vector<std::pair<StrutKey, int32_t>> run;
    for (int64_t t = t0; t < t1; ++t) {
      const int32_t n0 = tile_start[t], n1 = tile_start[t + 1];
      run.clear();
      for (int64_t q = hp[t]; q < hp[t + 1]; ++q) {
        const int32_t b = bperm[q];
        run.push_back({strut_key(xyz, conn[2 * (size_t)b], conn[2 * (size_t)b + 1], n0, n1), b});
      }
      std::stable_sort(run.begin(), run.end(), [](const auto &l, const auto &r) { return l.first < r.first; });
      for (int64_t q = hp[t]; q < hp[t + 1]; ++q) bperm[q] = run[q - hp[t]].second;
    }
  }, 16);
}

// conn must already be in the NEW strut numbering.
inline int build_tile_plan(TilePlan &plan, const std::vector<int32_t> &conn, int64_t N, int64_t B,
                           const std::vector<int32_t> &tile_start, const std::vector<int32_t> &tile_of,
                           const double *xyz = nullptr) {
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
  // every tile's foreign list in strut_key order (the tile kernels walk it in wave-sized pieces: see strut_key)
  parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
    std::vector<std::pair<StrutKey, int32_t>> run;
    for (int64_t t = t0; t < t1; ++t) {
      const int32_t n0 = tile_start[t], n1 = tile_start[t + 1];
      run.clear();
      for (int64_t k = foreign_ptr[t]; k < foreign_ptr[t + 1]; ++k) {
        const int32_t b = foreign_idx[k];
        run.push_back({strut_key(xyz, conn[2 * (size_t)b], conn[2 * (size_t)b + 1], n0, n1), b});
      }
      std::stable_sort(run.begin(), run.end(), [](const auto &l, const auto &r) { return l.first < r.first; });
      for (int64_t k = foreign_ptr[t]; k < foreign_ptr[t + 1]; ++k) foreign_idx[k] = run[k - foreign_ptr[t]].second;
    }
  }, 16);
  if (plan.foreign_idx.upload(foreign_idx) != hipSuccess) return 3;
  plan.ready = true;
  // ---- visit lists of the LDS-resident kernel ----
  plan.vis_ready = false;
  // direction palette (exact end-to-end vectors)
  struct D3 {
    uint64_t k[3];
    bool operator<(const D3 &o) const { return k[0] != o.k[0] ? k[0] < o.k[0] : k[1] != o.k[1] ? k[1] < o.k[1] : k[2] < o.k[2]; }
    bool operator==(const D3 &o) const { return k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2]; }
  };
  auto d3_of = [&](int64_t b) {
    D3 q;
    for (int k = 0; k < 3; ++k) {
      const double v = xyz[3 * (size_t)conn[2 * b + 1] + k] - xyz[3 * (size_t)conn[2 * b] + k] + 0.0;   // (-0 -> +0)
      std::memcpy(&q.k[k], &v, 8);
    }
    return q;
  };
  std::vector<D3> dirs;
  if (xyz) {
    std::mutex mu;
    bool too_many = false;
    parallel_for(B, [&](int64_t b0, int64_t b1, unsigned) {
      std::vector<D3> loc;
      for (int64_t b = b0; b < b1 && loc.size() <= (size_t)kDirMax; ++b) {
        const D3 q = d3_of(b);
        if (std::find(loc.begin(), loc.end(), q) == loc.end()) loc.push_back(q);
      }
      std::lock_guard<std::mutex> lk(mu);
      if (loc.size() > (size_t)kDirMax) too_many = true;
      dirs.insert(dirs.end(), loc.begin(), loc.end());
    }, 1 << 18);
    std::sort(dirs.begin(), dirs.end());
    dirs.erase(std::unique(dirs.begin(), dirs.end()), dirs.end());
    if (too_many || dirs.size() > (size_t)kDirMax) dirs.clear();
  }
  plan.n_dir = (int)dirs.size();
  {
    std::vector<double> table((size_t)std::max(1, plan.n_dir) * 4, 0.0);
    for (int q = 0; q < plan.n_dir; ++q)
      for (int k = 0; k < 3; ++k) std::memcpy(&table[4 * (size_t)q + k], &dirs[q].k[k], 8);
    if (plan.dir_table.upload(table) != hipSuccess) return 3;
  }
  std::vector<TileDesc> td((size_t)T);
  {
    int64_t v = 0, cx = 0;
    for (int64_t t = 0; t < T; ++t) {
      const int32_t n0 = tile_start[t], n1 = tile_start[t + 1];
      // home struts are numbered interior first (tile_strut_order): count them
      int64_t n_int = 0;
      for (int64_t b = home_ptr[t]; b < home_ptr[t + 1]; ++b) {
        const int32_t a = conn[2 * b], d = conn[2 * b + 1];
        const bool interior = a >= n0 && a < n1 && d >= n0 && d < n1;
        if (interior && b != home_ptr[t] + n_int) return 1;      // struts are not in tile_strut_order's order
        n_int += interior;
      }
      const int64_t nf = foreign_ptr[t + 1] - foreign_ptr[t], nh = home_ptr[t + 1] - home_ptr[t];
      td[t] = {n0, n1, (int32_t)n_int, (int32_t)(nh - n_int + nf), (int32_t)(nh - n_int), 0, v, cx, home_ptr[t], foreign_ptr[t]};
      v += nh + nf;
      cx += nh - n_int + nf;
    }
    plan.n_visits = v;
    plan.n_cross = cx;
  }
  std::vector<uint32_t> vloc((size_t)plan.n_visits);
  std::vector<int32_t> vstrut((size_t)plan.n_visits), vother((size_t)std::max<int64_t>(1, plan.n_cross));
  parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
    for (int64_t t = t0; t < t1; ++t) {
      const int32_t n0 = tile_start[t], n1 = tile_start[t + 1];
      int64_t v = td[t].v0, cx = td[t].c0;
      auto put = [&](int64_t b) {
        const int32_t a = conn[2 * b], d = conn[2 * b + 1];
        const bool inA = a >= n0 && a < n1, inB = d >= n0 && d < n1;
        uint32_t dir = 0;
        if (plan.n_dir > 0) dir = (uint32_t)(std::lower_bound(dirs.begin(), dirs.end(), d3_of(b)) - dirs.begin());
        uint32_t w;
        if (inA && inB) {
          w = (uint32_t)(a - n0) | ((uint32_t)(d - n0) << kVisRowBits);
        } else {      // exactly one end is this tile's (home and foreign lists hold nothing else)
          w = inB ? ((uint32_t)(d - n0) | (1u << kVisRowBits)) : (uint32_t)(a - n0);
          vother[cx++] = inB ? a : d;
        }
        vloc[v] = w | (dir << 21);
        vstrut[v++] = (int32_t)b;
      };
      for (int64_t b = home_ptr[t]; b < home_ptr[t + 1]; ++b) put(b);
      for (int64_t k = foreign_ptr[t]; k < foreign_ptr[t + 1]; ++k) put(foreign_idx[k]);
    }
  }, 16);
  if (plan.tdesc.upload(td) != hipSuccess || plan.vloc.upload(vloc) != hipSuccess ||
      plan.vstrut.upload(vstrut) != hipSuccess || plan.vother.upload(vother) != hipSuccess)
    return 3;
  plan.vis_ready = true;
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

// ----------------------------------------------------------------------------------------------------------
// LDS-resident K*p (round 3).  A tile copies its OWN rows of x (coalesced, one 16-byte load per lane) and the record
// palette (a periodic lattice has a few dozen distinct records) into LDS.  An interior visit - two thirds of all - is
// then one streamed 32-bit word + a 2-byte dense palette id, ten ds_read_b128, the arithmetic and twelve ds_add_f64:
// no gather from global memory and no dependent memory hop inside the loop (the gather kernel above spends 52 % of its
// waves' life in s_waitcnt behind two hops per visit).  A crossing visit gathers the one out-of-tile row from global
// memory - requested before the interior loop starts - and accumulates only the tile's own end.
// (First form, measured and replaced: the out-of-tile rows staged in LDS as a per-tile halo list - three dependent hops
// (descriptor, halo index, row) before the barrier: 4.2 us of a 10.3-us workgroup life, K*p 45.6 us against 41.0.)
// Same accumulation, masks, fused dot and node-elimination passes as k_spmv_tile; used when the palette applies with
// <= kPalDenseMax entries (TilePlan::vis_ready, pl_context::pal_lds).
// ----------------------------------------------------------------------------------------------------------
constexpr int kPalDenseMax = 256;
// 16-byte chunks per record of the LDS palette: 5 (80 B) spreads records p and p + 4 over different banks - with 4 the
// twelve records of an Octet lattice fall on four bank groups, three deep (measured: K*p 47.0 against 45.6 us)
#ifndef PL_LDS_PALSTRIDE
#define PL_LDS_PALSTRIDE 5
#endif
constexpr int kPalLdsChunks = PL_LDS_PALSTRIDE;
#ifndef PL_LDS_BLOCK
#define PL_LDS_BLOCK 512
#endif
constexpr int kLdsBlock = PL_LDS_BLOCK;
// waves per SIMD the compiler must allow (8 = 64 VGPRs: four 512-thread workgroups per CU; left alone it interleaves the
// unrolled visits and takes 78)
#ifndef PL_LDS_WAVES
#define PL_LDS_WAVES 8
#endif
#ifndef PL_LDS_CROSS_FIRST
#define PL_LDS_CROSS_FIRST 0
#endif
#ifndef PL_LDS_WAVES_STREAM
#define PL_LDS_WAVES_STREAM 6
#endif
#ifndef PL_LDS_PRE
#define PL_LDS_PRE 3
#endif
constexpr int kLdsPre = PL_LDS_PRE;               // interior visits per thread whose words are fetched before the barrier
constexpr unsigned kNoVisit = 0xFFFFFFFFu;         // (bit 31 of a visit word is never set)

// One 32-bit word per visit: the plan's static bits (TilePlan::vloc: local rows in bits 0..20, direction-palette entry in
// bits 21..28) with, in the palette form, the strut's dense RECORD-palette id in place of the direction entry, and the
// condensed-end bits of the strut (end A, end B) in bits 29..30.  k_visit_words writes both forms: `vword_pal` behind every
// palette build (ids change with the radii), `vword_dir` whenever the set of eliminated nodes changes.
constexpr int kVisPidShift = 21, kVisCendShift = 29;
__global__ __launch_bounds__(kBlock) void k_visit_words(int64_t n_visits, const uint32_t *__restrict__ vloc,
                                                        const int32_t *__restrict__ vstrut,
                                                        const uint16_t *__restrict__ pal /* null: direction form */,
                                                        const int *__restrict__ dense_of_slot,
                                                        const uint8_t *__restrict__ cend /* may be null */,
                                                        uint32_t *__restrict__ vword) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (v >= n_visits) return;
  const int32_t b = vstrut[v];
  unsigned w = vloc[v];
  if (pal) {      // (dense ids are only read when the palette has <= 256 entries)
    const unsigned d = (unsigned)dense_of_slot[pal[b]] & 0xFFu;
    w = (w & ~(0xFFu << kVisPidShift)) | (d << kVisPidShift);
  }
  if (cend) w |= (unsigned)(cend[b] & 3u) << kVisCendShift;
  vword[v] = w;
}

__device__ __forceinline__ Rec5 load_rec5(const Rec5 *__restrict__ rec5, int64_t b) { return rec5[b]; }

// REC = kRecPalette: tab = the dense record palette (Record[n_tab]), the whole record of a visit comes from LDS;
// REC = kRecCompact: tab = the direction palette (double[n_tab][4]), d from LDS and the five stiffness scalars of the strut
//   streamed from rec5 - home visits are consecutive struts (TileDesc::h0), so a wave reads 2.5 KB contiguous; crossing
//   visits run FIRST in this form (their record and out-of-tile row are requested before the barrier; nothing of them
//   stays live across the interior visits, which hold the next visit's record in registers instead).
// DEFER (short form of the PCG iteration, pl_small.h): the operand is not stored yet - the previous launch wrote z = M^-1 r and
// the slots of r.z, so beta is known only now.  The kernel forms p = z + beta p_old itself: for its own rows while staging them
// (and stores them to p_new, together with the iterate's update x += alpha_prev p_old, which nobody else would make), for the
// out-of-tile row of a crossing visit from the two gathered rows.  p_old and p_new are different buffers: another tile may still
// be reading the old rows.  Scalars: sc_prev / sc_cur = the slotted scalar sets of the previous / this iteration
// (S_RZ_OLD = r.z, S_PAP = p.Kp): alpha_prev = rz_prev / pap_prev, beta = rz_cur / rz_prev (0 where the denominator is 0:
// the first iteration starts from p_old = 0).  fp64 only.
struct Defer {
  const double *z = nullptr, *p_old = nullptr;
  double *p_new = nullptr, *xsol = nullptr;
  const double *sc_prev = nullptr, *sc_cur = nullptr;
};
template <bool MASK, bool DOT, int REC, typename VT, int ENDS = kEndsAll, bool DEFER = false>
__global__ __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(DEFER ? 1 : (REC == kRecPalette ? PL_LDS_WAVES : PL_LDS_WAVES_STREAM), 8)))
void k_spmv_tile_lds_t(const TileDesc *__restrict__ tdesc, const uint32_t *__restrict__ vword,
                       const int32_t *__restrict__ vother, const double2 *__restrict__ tab, int n_tab,
                     const Rec5 *__restrict__ rec5, const int32_t *__restrict__ foreign_idx,
                     const uint8_t *__restrict__ fixedbits, const VT *__restrict__ x, VT *__restrict__ y,
                     double *__restrict__ dot_out, int stride, const uint8_t *__restrict__ cflag = nullptr,
                     CondSolve cs = CondSolve(), const int32_t *__restrict__ tile_list = nullptr, Defer df = Defer()) {
  static_assert(!DEFER || sizeof(VT) == 8, "the deferred direction is an fp64 path");
  constexpr bool kToCondensed = ENDS == kEndsCondensed || ENDS == kEndsCondensedSolve;
  double d_alpha = 0.0, d_beta = 0.0;
  if constexpr (DEFER) {
    const double rz_prev = scalar_read(df.sc_prev, 0 /* S_RZ_OLD */), pap_prev = scalar_read(df.sc_prev, 1 /* S_PAP */);
    const double rz_cur = scalar_read(df.sc_cur, 0);
    d_alpha = (pap_prev != 0.0) ? rz_prev / pap_prev : 0.0;
    d_beta = (rz_prev != 0.0) ? rz_cur / rz_prev : 0.0;
  }
  // an out-of-tile row of the operand (crossing visits)
  auto load_other = [&](int32_t node, V3 &u, V3 &t) {
    if constexpr (DEFER) {
      V3 uz, tz, up, tp;
      load6(reinterpret_cast<const double *>(df.z) + 6 * (int64_t)node, uz, tz);
      load6(reinterpret_cast<const double *>(df.p_old) + 6 * (int64_t)node, up, tp);
      u = uz + d_beta * up;
      t = tz + d_beta * tp;
    } else {
      load6(x + 6 * (int64_t)node, u, t);
    }
  };
  constexpr bool kStream = REC == kRecCompact;
  constexpr int kSrcChunks = kStream ? 2 : 4, kTabChunks = kStream ? 2 : kPalLdsChunks;
  struct NoRec {};
  using RQ = typename std::conditional<kStream, Rec5, NoRec>::type;      // a visit's streamed scalars (nothing in the palette form)
  extern __shared__ double ys[];                                        // [6][stride] accumulator, component-major
  double2 *xs2 = reinterpret_cast<double2 *>(ys + 6 * stride);          // [stride][3]: the tile's rows of x, node-major
  double2 *ps2 = xs2 + 3 * stride;                                      // [n_tab][kTabChunks]: record / direction palette
  uint8_t *sflag = reinterpret_cast<uint8_t *>(ps2 + kTabChunks * n_tab);      // [stride] condensed flags (ENDS != All)
  __shared__ double red[kLdsBlock / kWave];
  __shared__ int32_t sbase[ENDS == kEndsCondensedSolve ? kTileMaxNodes : 1];
  PL_STAMP(0);
  unsigned t = xcd_block(blockIdx.x, gridDim.x);
  if (tile_list) t = (unsigned)tile_list[t];
  const TileDesc td = tdesc[t];
  const int n0 = td.n0, nn = td.n1 - td.n0;
  // this thread's first kLdsPre interior visits and its first crossing visit: requested together with the rows of x, so
  // that the interior visits hold no load of a visit word at all (a load inside a loop makes the compiler wait for
  // vmcnt(0) at the top of every iteration - for the word it has just requested AND for whatever else is meant to stay
  // in flight behind the loop)
  unsigned wv[kLdsPre];
#pragma unroll
  for (int j = 0; j < kLdsPre; ++j) {
    const int k = (int)threadIdx.x + j * kLdsBlock;
    wv[j] = k < td.n_int ? vword[td.v0 + k] : kNoVisit;
  }
  int kc = threadIdx.x;
  bool clive = kc < td.n_cross;
  unsigned cw = 0;
  int32_t co = 0;
  int64_t cb = 0;                      // (streaming form) the crossing visit's strut
  (void)cb;
  const int64_t vc0 = td.v0 + td.n_int;
  auto cross_strut = [&](int k) -> int64_t {
    return k < td.n_ch ? td.h0 + td.n_int + k : (int64_t)foreign_idx[td.f0 + (k - td.n_ch)];
  };
  if (clive) {
    cw = vword[vc0 + kc];
    co = vother[td.c0 + kc];
    if constexpr (kStream) cb = cross_strut(kc);
  }
  RQ qn = RQ();                        // (streaming form) record of the next interior visit
  if constexpr (kStream)
    if (wv[0] != kNoVisit) qn = load_rec5(rec5, td.h0 + threadIdx.x);
  for (int i = threadIdx.x; i < 6 * stride; i += kLdsBlock) ys[i] = 0.0;
  for (int i = threadIdx.x; i < 3 * nn; i += kLdsBlock) {               // own rows: 16 B per lane, contiguous
    uint8_t f = 0;
    if (ENDS != kEndsAll) {
      const int node = i / 3;
      f = cflag[n0 + node];
      if (i - 3 * node == 0) sflag[node] = f;
    }
    // (fused first pass of the condensed operator: a condensed node's row is being rewritten by its tile - it counts as
    // zero and is not read)
    double2 val = {0.0, 0.0};
    if constexpr (DEFER) {
      if (!(ENDS == kEndsCondensedSolve && f)) {
        const int64_t pr = 3 * (int64_t)n0 + i;
        const double2 zz = load_pair(df.z, pr), po = load_pair(df.p_old, pr);
        double2 xx = load_pair(df.xsol, pr);
        val.x = zz.x + d_beta * po.x;
        val.y = zz.y + d_beta * po.y;
        xx.x += d_alpha * po.x;
        xx.y += d_alpha * po.y;
        store_pair(df.p_new, pr, val);
        store_pair(df.xsol, pr, xx);
      }
    } else {
      if (!(ENDS == kEndsCondensedSolve && f)) val = load_pair(x, 3 * (int64_t)n0 + i);
    }
    xs2[i] = val;
  }
  for (int i = threadIdx.x; i < kSrcChunks * n_tab; i += kLdsBlock)
    ps2[(i / kSrcChunks) * kTabChunks + (i % kSrcChunks)] = tab[i];
  if (ENDS == kEndsCondensedSolve)
    for (int i = threadIdx.x; i < nn; i += kLdsBlock) sbase[i] = cs.base[n0 + i];
  // a crossing visit: is the tile's own end of the kind this pass accumulates, and does the other end's row count?
  auto cross_take = [&](unsigned cwv) -> bool {
    if (ENDS == kEndsAll) return true;
    const bool ownB = (cwv >> kVisRowBits) & 1u;
    return (((cwv >> (kVisCendShift + (ownB ? 1 : 0))) & 1u) != 0) == kToCondensed;
  };
  auto other_zero = [&](unsigned cwv) -> bool {        // (first pass: condensed rows count as zero)
    if (ENDS != kEndsCondensedSolve) return false;
    const bool ownB = (cwv >> kVisRowBits) & 1u;
    return ((cwv >> (kVisCendShift + (ownB ? 0 : 1))) & 1u) != 0;
  };
  // the first crossing visit's out-of-tile row (and, streaming, its record): in flight during the barrier - and, in the
  // palette form, during the interior visits
  V3 uO = {0, 0, 0}, tO = {0, 0, 0};
  RQ cq = RQ();
  if (clive && cross_take(cw)) {
    if (!other_zero(cw)) load_other(co, uO, tO);
    if constexpr (kStream) cq = load_rec5(rec5, cb);
  }
  PL_STAMP(1);
  __syncthreads();
  PL_STAMP(2);
  // record of a visit: palette form - all of it from LDS; streaming form - d from the LDS direction table, the rest from q
  auto record_of = [&](unsigned id, const RQ &q) -> Record {
    const double2 *p = ps2 + kTabChunks * id;
    Record r;
    if constexpr (kStream) {
      const double2 d0 = p[0];
      const double dz = reinterpret_cast<const double *>(p)[2];
      r.a = q.a; r.c = q.c; r.e1 = q.e1; r.e2 = q.e2; r.e3 = q.e3; r.dx = d0.x; r.dy = d0.y; r.dz = dz;
    } else {
      const double2 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
      r.a = r0.x; r.c = r0.y; r.e1 = r1.x; r.e2 = r1.y; r.e3 = r2.x; r.dx = r2.y; r.dy = r3.x; r.dz = r3.y;
    }
    return r;
  };
  auto interior = [&](unsigned w, const RQ &q) {
    const int la = (int)(w & ((1u << kVisRowBits) - 1)), lb = (int)((w >> kVisRowBits) & ((1u << kVisRowBits) - 1));
    bool takeA = true, takeB = true;
    if (ENDS != kEndsAll) {
      takeA = (((w >> kVisCendShift) & 1u) != 0) == kToCondensed;
      takeB = (((w >> kVisCendShift) & 2u) != 0) == kToCondensed;
    }
    if (takeA || takeB) {
      const Record r = record_of((w >> kVisPidShift) & 0xFFu, q);
      const double2 *pa = xs2 + 3 * la, *pb = xs2 + 3 * lb;
      const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
      const V3 uA = {a0.x, a0.y, a1.x}, tA = {a1.y, a2.x, a2.y}, uB = {b0.x, b0.y, b1.x}, tB = {b1.y, b2.x, b2.y};
      V3 F, M;
      tip_force(r, uA, tA, uB, tB, F, M);
      if (takeB) lds_add6(ys + lb, stride, F, M);
      if (takeA) {
        const V3 d = {r.dx, r.dy, r.dz};
        lds_add6(ys + la, stride, (-1.0) * F, (-1.0) * M - cross(d, F));
      }
    }
  };
  auto interior_visits = [&]() {
#pragma unroll
    for (int j = 0; j < kLdsPre; ++j) {
      const RQ qc = qn;
      if constexpr (kStream)
        if (j + 1 < kLdsPre && wv[j + 1] != kNoVisit)
          qn = load_rec5(rec5, td.h0 + (int64_t)threadIdx.x + (j + 1) * kLdsBlock);
      if (wv[j] != kNoVisit) interior(wv[j], qc);
      __builtin_amdgcn_sched_barrier(0);      // (visits interleaved by the scheduler cost 14 more registers: spills at 64)
    }
    for (int k = (int)threadIdx.x + kLdsPre * kLdsBlock; k < td.n_int; k += kLdsBlock) {      // large tiles
      RQ q = RQ();
      if constexpr (kStream) q = load_rec5(rec5, td.h0 + k);
      interior(vword[td.v0 + k], q);
    }
  };
  auto crossing_visits = [&]() {
    while (clive) {
      const int kn = kc + kLdsBlock;
      const bool live_n = kn < td.n_cross;
      unsigned cw_n = 0;
      int32_t co_n = 0;
      int64_t cb_n = 0;
      (void)cb_n;
      if (live_n) {
        cw_n = vword[vc0 + kn];
        co_n = vother[td.c0 + kn];
        if constexpr (kStream) cb_n = cross_strut(kn);
      }
      if (cross_take(cw)) {
        const int lo = (int)(cw & ((1u << kVisRowBits) - 1));
        const bool ownB = (cw >> kVisRowBits) & 1u;
        const Record r = record_of((cw >> kVisPidShift) & 0xFFu, cq);
        const double2 *po = xs2 + 3 * lo;
        const double2 a0 = po[0], a1 = po[1], a2 = po[2];
        const V3 uW = {a0.x, a0.y, a1.x}, tW = {a1.y, a2.x, a2.y};
        V3 F, M;
        if (ownB) {
          tip_force(r, uO, tO, uW, tW, F, M);
          lds_add6(ys + lo, stride, F, M);
        } else {
          tip_force(r, uW, tW, uO, tO, F, M);
          const V3 d = {r.dx, r.dy, r.dz};
          lds_add6(ys + lo, stride, (-1.0) * F, (-1.0) * M - cross(d, F));
        }
      }
      uO = {0, 0, 0};
      tO = {0, 0, 0};
      if (live_n && cross_take(cw_n)) {
        if (!other_zero(cw_n)) load_other(co_n, uO, tO);
        if constexpr (kStream) cq = load_rec5(rec5, cb_n);
      }
      kc = kn;
      cw = cw_n;
      clive = live_n;
    }
  };
  if constexpr (kStream || PL_LDS_CROSS_FIRST) {
    crossing_visits();
    interior_visits();
  } else {
    interior_visits();
    crossing_visits();
  }
  PL_STAMP(3);
  __syncthreads();
  PL_STAMP(4);
  if (ENDS == kEndsCondensedSolve) {
    for (int i = threadIdx.x; i < nn * 6; i += kLdsBlock) {
      const int node = i / 6, k = i - 6 * node;
      const int32_t b0 = sbase[node];
      if (b0 < 0) continue;
      const double *A = cs.inv + b0 + 6 * k;
      double vv = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) vv += A[j] * ys[j * stride + node];
      y[6 * (int64_t)(n0 + node) + k] = (VT)(-vv);
    }
    return;
  }
  double acc = 0.0;
  const int64_t pair0 = 3 * (int64_t)n0;
  for (int i = threadIdx.x; i < nn * 3; i += kLdsBlock) {
    const int node = i / 3, part = i - 3 * node;
    if (ENDS != kEndsAll && ((sflag[node] != 0) != (ENDS == kEndsCondensed))) continue;   // rows of the other kind
    double2 val = {ys[(2 * part) * stride + node], ys[(2 * part + 1) * stride + node]};
    if (MASK) {
      const unsigned fb = fixedbits[n0 + node] >> (2 * part);
      if (fb & 1u) val.x = 0.0;
      if (fb & 2u) val.y = 0.0;
    }
    store_pair(y, pair0 + i, val);
    if (DOT) {
      const double2 xv = xs2[i];
      acc += xv.x * val.x + xv.y * val.y;
    }
  }
  if (DOT) {
    double s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = 0.0;
    if (threadIdx.x == 0)
      for (int q = 0; q < kLdsBlock / kWave; ++q) s += red[q];
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
  }
  PL_STAMP(5);
}

// The palette form as its own function: the same text as k_spmv_tile_lds_t<.., kRecPalette, ..>, which the register allocator
// fits into 64 VGPRs here and spills from there (16 VGPRs of the crossing visit's row to scratch: 2 x 27 MB of traffic per
// launch at 50^3, K*p 33.7 -> 36.4 us) - the streaming form is held to 6 waves per SIMD instead.
template <bool MASK, bool DOT, typename VT, int ENDS = kEndsAll>
__global__ __launch_bounds__(kLdsBlock) __attribute__((amdgpu_waves_per_eu(PL_LDS_WAVES, 8))) void k_spmv_tile_lds(const TileDesc *__restrict__ tdesc,
                                                         const uint32_t *__restrict__ vword,
                                                         const int32_t *__restrict__ vother,
                                                         const Record *__restrict__ pal_dense, int n_pal,
                                                         const uint8_t *__restrict__ fixedbits,
                                                         const VT *__restrict__ x, VT *__restrict__ y,
                                                         double *__restrict__ dot_out, int stride,
                                                         const uint8_t *__restrict__ cflag = nullptr,
                                                         CondSolve cs = CondSolve(),
                                                         const int32_t *__restrict__ tile_list = nullptr) {
  constexpr bool kToCondensed = ENDS == kEndsCondensed || ENDS == kEndsCondensedSolve;
  extern __shared__ double ys[];                                        // [6][stride] accumulator, component-major
  double2 *xs2 = reinterpret_cast<double2 *>(ys + 6 * stride);          // [stride][3]: the tile's rows of x, node-major
  double2 *ps2 = xs2 + 3 * stride;                                      // [n_pal][kPalLdsChunks]: the record palette
  uint8_t *sflag = reinterpret_cast<uint8_t *>(ps2 + kPalLdsChunks * n_pal);   // [stride] condensed flags (ENDS != All)
  __shared__ double red[kLdsBlock / kWave];
  __shared__ int32_t sbase[ENDS == kEndsCondensedSolve ? kTileMaxNodes : 1];
  PL_STAMP(0);
  unsigned t = xcd_block(blockIdx.x, gridDim.x);
  if (tile_list) t = (unsigned)tile_list[t];
  const TileDesc td = tdesc[t];
  const int n0 = td.n0, nn = td.n1 - td.n0;
  // this thread's first kLdsPre interior visits and its first crossing visit: requested together with the rows of x, so
  // that the interior loop holds no load from global memory at all (a load inside it makes the compiler wait for
  // vmcnt(0) at the top of every iteration - for the word it has just requested AND for the crossing visit's row, which
  // is meant to stay in flight behind the loop)
  unsigned wv[kLdsPre];
#pragma unroll
  for (int j = 0; j < kLdsPre; ++j) {
    const int k = (int)threadIdx.x + j * kLdsBlock;
    wv[j] = k < td.n_int ? vword[td.v0 + k] : kNoVisit;
  }
  int kc = threadIdx.x;
  bool clive = kc < td.n_cross;
  unsigned cw = 0;
  int32_t co = 0;
  const int64_t vc0 = td.v0 + td.n_int;
  if (clive) {
    cw = vword[vc0 + kc];
    co = vother[td.c0 + kc];
  }
  for (int i = threadIdx.x; i < 6 * stride; i += kLdsBlock) ys[i] = 0.0;
  for (int i = threadIdx.x; i < 3 * nn; i += kLdsBlock) {               // own rows: 16 B per lane, contiguous
    uint8_t f = 0;
    if (ENDS != kEndsAll) {
      const int node = i / 3;
      f = cflag[n0 + node];
      if (i - 3 * node == 0) sflag[node] = f;
    }
    // (fused first pass of the condensed operator: a condensed node's row is being rewritten by its tile - it counts as
    // zero and is not read)
    double2 val = {0.0, 0.0};
    if (!(ENDS == kEndsCondensedSolve && f)) val = load_pair(x, 3 * (int64_t)n0 + i);
    xs2[i] = val;
  }
  for (int i = threadIdx.x; i < 4 * n_pal; i += kLdsBlock)
    ps2[(i >> 2) * kPalLdsChunks + (i & 3)] = reinterpret_cast<const double2 *>(pal_dense)[i];
  if (ENDS == kEndsCondensedSolve)
    for (int i = threadIdx.x; i < nn; i += kLdsBlock) sbase[i] = cs.base[n0 + i];
  // a crossing visit: is the tile's own end of the kind this pass accumulates, and does the other end's row count?
  auto cross_take = [&](unsigned cwv) -> bool {
    if (ENDS == kEndsAll) return true;
    const bool ownB = (cwv >> kVisRowBits) & 1u;
    return (((cwv >> (kVisCendShift + (ownB ? 1 : 0))) & 1u) != 0) == kToCondensed;
  };
  auto other_zero = [&](unsigned cwv) -> bool {        // (first pass: condensed rows count as zero)
    if (ENDS != kEndsCondensedSolve) return false;
    const bool ownB = (cwv >> kVisRowBits) & 1u;
    return ((cwv >> (kVisCendShift + (ownB ? 0 : 1))) & 1u) != 0;
  };
  // the first crossing visit's out-of-tile row: in flight during the barrier and the interior loop
  V3 uO = {0, 0, 0}, tO = {0, 0, 0};
  if (clive && cross_take(cw) && !other_zero(cw)) load6(x + 6 * (int64_t)co, uO, tO);
  PL_STAMP(1);
  __syncthreads();
  PL_STAMP(2);
  auto record_of = [&](unsigned id) -> Record {
    const double2 *q = ps2 + kPalLdsChunks * id;
    const double2 r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3];
    Record r;
    r.a = r0.x; r.c = r0.y; r.e1 = r1.x; r.e2 = r1.y; r.e3 = r2.x; r.dx = r2.y; r.dy = r3.x; r.dz = r3.y;
    return r;
  };
  auto interior = [&](unsigned w) {
    const int la = (int)(w & ((1u << kVisRowBits) - 1)), lb = (int)((w >> kVisRowBits) & ((1u << kVisRowBits) - 1));
    bool takeA = true, takeB = true;
    if (ENDS != kEndsAll) {
      takeA = (((w >> kVisCendShift) & 1u) != 0) == kToCondensed;
      takeB = (((w >> kVisCendShift) & 2u) != 0) == kToCondensed;
    }
    if (ENDS == kEndsCondensedSolve && takeA != takeB) {
      // first pass of the condensed operator, strut with ONE eliminated end (every strut of a bipartite lattice): that
      // end's row counts as zero - it is neither read nor carried through the arithmetic, and only its force is wanted.
      // (Visits are sorted by direction, so which end it is, is uniform over nearly every wave.)
      const Record r = record_of((w >> kVisPidShift) & 0xFFu);
      const V3 d = {r.dx, r.dy, r.dz};
      V3 F, M;
      if (takeA) {        // A eliminated: du = u_B, dth = th_B; force on A = -(F, M + d x F)
        const double2 *pb = xs2 + 3 * lb;
        const double2 b0 = pb[0], b1 = pb[1], b2 = pb[2];
        const V3 du = {b0.x, b0.y, b1.x}, dth = {b1.y, b2.x, b2.y};
        F = r.a * du + (r.e1 * dot(du, d)) * d + r.e2 * cross(d, dth);
        M = r.c * dth + (r.e3 * dot(dth, d)) * d - r.e2 * cross(d, du);
        lds_add6(ys + la, stride, (-1.0) * F, (-1.0) * M - cross(d, F));
      } else {            // B eliminated: du = -u_A + d x th_A, dth = -th_A; force on B = (F, M)
        const double2 *pa = xs2 + 3 * la;
        const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2];
        const V3 uA = {a0.x, a0.y, a1.x}, tA = {a1.y, a2.x, a2.y};
        const V3 du = cross(d, tA) - uA, dth = (-1.0) * tA;
        F = r.a * du + (r.e1 * dot(du, d)) * d + r.e2 * cross(d, dth);
        M = r.c * dth + (r.e3 * dot(dth, d)) * d - r.e2 * cross(d, du);
        lds_add6(ys + lb, stride, F, M);
      }
    } else if (takeA || takeB) {
      const Record r = record_of((w >> kVisPidShift) & 0xFFu);
      const double2 *pa = xs2 + 3 * la, *pb = xs2 + 3 * lb;
      const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
      const V3 uA = {a0.x, a0.y, a1.x}, tA = {a1.y, a2.x, a2.y}, uB = {b0.x, b0.y, b1.x}, tB = {b1.y, b2.x, b2.y};
      V3 F, M;
      tip_force(r, uA, tA, uB, tB, F, M);
      if (takeB) lds_add6(ys + lb, stride, F, M);
      if (takeA) {
        const V3 d = {r.dx, r.dy, r.dz};
        lds_add6(ys + la, stride, (-1.0) * F, (-1.0) * M - cross(d, F));
      }
    }
  };
#pragma unroll
  for (int j = 0; j < kLdsPre; ++j)
    if (wv[j] != kNoVisit) interior(wv[j]);
  for (int k = (int)threadIdx.x + kLdsPre * kLdsBlock; k < td.n_int; k += kLdsBlock) interior(vword[td.v0 + k]);   // large tiles
  while (clive) {
    const int kn = kc + kLdsBlock;
    const bool live_n = kn < td.n_cross;
    unsigned cw_n = 0;
    int32_t co_n = 0;
    if (live_n) {
      cw_n = vword[vc0 + kn];
      co_n = vother[td.c0 + kn];
    }
    if (cross_take(cw)) {
      const int lo = (int)(cw & ((1u << kVisRowBits) - 1));
      const bool ownB = (cw >> kVisRowBits) & 1u;
      const Record r = record_of((cw >> kVisPidShift) & 0xFFu);
      const double2 *po = xs2 + 3 * lo;
      const double2 a0 = po[0], a1 = po[1], a2 = po[2];
      const V3 uW = {a0.x, a0.y, a1.x}, tW = {a1.y, a2.x, a2.y};
      V3 F, M;
      if (ownB) {
        tip_force(r, uO, tO, uW, tW, F, M);
        lds_add6(ys + lo, stride, F, M);
      } else {
        tip_force(r, uW, tW, uO, tO, F, M);
        const V3 d = {r.dx, r.dy, r.dz};
        lds_add6(ys + lo, stride, (-1.0) * F, (-1.0) * M - cross(d, F));
      }
    }
    uO = {0, 0, 0};
    tO = {0, 0, 0};
    if (live_n && cross_take(cw_n) && !other_zero(cw_n)) load6(x + 6 * (int64_t)co_n, uO, tO);
    kc = kn;
    cw = cw_n;
    clive = live_n;
  }
  PL_STAMP(3);
  __syncthreads();
  PL_STAMP(4);
  if (ENDS == kEndsCondensedSolve) {
    for (int i = threadIdx.x; i < nn * 6; i += kLdsBlock) {
      const int node = i / 6, k = i - 6 * node;
      const int32_t b0 = sbase[node];
      if (b0 < 0) continue;
      const double *A = cs.inv + b0 + 6 * k;
      double vv = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) vv += A[j] * ys[j * stride + node];
      y[6 * (int64_t)(n0 + node) + k] = (VT)(-vv);
    }
    return;
  }
  double acc = 0.0;
  const int64_t pair0 = 3 * (int64_t)n0;
  for (int i = threadIdx.x; i < nn * 3; i += kLdsBlock) {
    const int node = i / 3, part = i - 3 * node;
    if (ENDS != kEndsAll && ((sflag[node] != 0) != (ENDS == kEndsCondensed))) continue;   // rows of the other kind
    double2 val = {ys[(2 * part) * stride + node], ys[(2 * part + 1) * stride + node]};
    if (MASK) {
      const unsigned fb = fixedbits[n0 + node] >> (2 * part);
      if (fb & 1u) val.x = 0.0;
      if (fb & 2u) val.y = 0.0;
    }
    store_pair(y, pair0 + i, val);
    if (DOT) {
      const double2 xv = xs2[i];
      acc += xv.x * val.x + xv.y * val.y;
    }
  }
  if (DOT) {
    double s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = 0.0;
    if (threadIdx.x == 0)
      for (int q = 0; q < kLdsBlock / kWave; ++q) s += red[q];
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), s);
  }
  PL_STAMP(5);
}


// false: this launch does not fit the LDS-resident kernel (the caller falls back to launch_tile_spmv).
// rec5 == nullptr: palette form (tab = dense record palette); else streaming form (tab = the plan's direction palette).
template <typename VT>
inline bool launch_tile_spmv_lds(const TilePlan &plan, const uint32_t *vword, const void *tab, int n_tab,
                                 const Rec5 *rec5, const uint8_t *fixedbits, const VT *x, VT *y, double *dot_dev,
                                 hipStream_t s, int ends = kEndsAll, const uint8_t *cflag = nullptr,
                                 CondSolve cs = CondSolve(), const int32_t *tile_list = nullptr, int64_t n_list = 0) {
  if (!plan.vis_ready || !vword || n_tab <= 0 || n_tab > kPalDenseMax) return false;
  const int stride = plan.max_nodes | 1;
  const size_t lds = (size_t)stride * 96 + (size_t)n_tab * 16 * (rec5 ? 2 : kPalLdsChunks) + (ends != kEndsAll ? (size_t)stride : 0);
  if (lds > 60 * 1024) return false;
  if (tile_list && n_list <= 0) return true;
  const int64_t n_units = tile_list ? n_list : plan.n_tiles;
  const dim3 g((unsigned)n_units), blk(kLdsBlock);
  const double2 *tab2 = static_cast<const double2 *>(tab);
#define PL_T(M, D, R, E)                                                                                              \
  do {                                                                                                                \
    if (R == kRecPalette)                                                                                             \
      hipLaunchKernelGGL((k_spmv_tile_lds<M, D, VT, E>), g, blk, lds, s, plan.tdesc.p, vword, plan.vother.p,          \
                         static_cast<const Record *>(tab), n_tab, fixedbits, x, y, dot_dev, stride, cflag, cs,        \
                         tile_list);                                                                                  \
    else                                                                                                              \
      hipLaunchKernelGGL((k_spmv_tile_lds_t<M, D, kRecCompact, VT, E>), g, blk, lds, s, plan.tdesc.p, vword,          \
                         plan.vother.p, tab2, n_tab, rec5, plan.foreign_idx.p, fixedbits, x, y, dot_dev, stride,      \
                         cflag, cs, tile_list);                                                                       \
  } while (0)
#define PL_TT(R, E)                                           \
  do {                                                        \
    if (fixedbits && dot_dev) PL_T(true, true, R, E);         \
    else if (fixedbits) PL_T(true, false, R, E);              \
    else if (dot_dev) PL_T(false, true, R, E);                \
    else PL_T(false, false, R, E);                            \
  } while (0)
#define PL_TE(R)                                                                       \
  do {                                                                                 \
    if (ends == kEndsCondensed) PL_TT(R, kEndsCondensed);                              \
    else if (ends == kEndsCondensedSolve) PL_T(false, false, R, kEndsCondensedSolve);  \
    else if (ends == kEndsOthers) PL_TT(R, kEndsOthers);                               \
    else PL_TT(R, kEndsAll);                                                           \
  } while (0)
  if (rec5) PL_TE(kRecCompact);
  else PL_TE(kRecPalette);
#undef PL_TE
#undef PL_TT
#undef PL_T
  return true;
}

// The same launch with the operand formed in the kernel (Defer, short form of the PCG iteration): fp64, all tiles, and only
// the two passes that start an operator application - ends = kEndsAll (masked, with the dot) or kEndsCondensedSolve.
inline bool launch_tile_spmv_lds_defer(const TilePlan &plan, const uint32_t *vword, const void *tab, int n_tab,
                                       const Rec5 *rec5, const uint8_t *fixedbits, double *y, double *dot_dev, hipStream_t s,
                                       int ends, const uint8_t *cflag, CondSolve cs, const Defer &df) {
  if (!plan.vis_ready || !vword || n_tab <= 0 || n_tab > kPalDenseMax) return false;
  if (ends != kEndsAll && ends != kEndsCondensedSolve) return false;
  const int stride = plan.max_nodes | 1;
  const size_t lds = (size_t)stride * 96 + (size_t)n_tab * 16 * (rec5 ? 2 : kPalLdsChunks) + (ends != kEndsAll ? (size_t)stride : 0);
  if (lds > 60 * 1024) return false;
  const dim3 g((unsigned)plan.n_tiles), blk(kLdsBlock);
  const double2 *tab2 = static_cast<const double2 *>(tab);
  const double *x = nullptr;
#define PL_TD(R)                                                                                                          \
  do {                                                                                                                    \
    if (ends == kEndsCondensedSolve)                                                                                      \
      hipLaunchKernelGGL((k_spmv_tile_lds_t<false, false, R, double, kEndsCondensedSolve, true>), g, blk, lds, s,         \
                         plan.tdesc.p, vword, plan.vother.p, tab2, n_tab, rec5, plan.foreign_idx.p, fixedbits, x, y,      \
                         dot_dev, stride, cflag, cs, (const int32_t *)nullptr, df);                                       \
    else                                                                                                                  \
      hipLaunchKernelGGL((k_spmv_tile_lds_t<true, true, R, double, kEndsAll, true>), g, blk, lds, s, plan.tdesc.p, vword, \
                         plan.vother.p, tab2, n_tab, rec5, plan.foreign_idx.p, fixedbits, x, y, dot_dev, stride, cflag,   \
                         cs, (const int32_t *)nullptr, df);                                                               \
  } while (0)
  if (rec5) PL_TD(kRecCompact);
  else PL_TD(kRecPalette);
#undef PL_TD
  return true;
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
