// Host-side lattice generation of libpylattice_hip (plain C++17, std::thread; compiled with -ffp-contract=off so that
// coordinates come out bit-identical to the numpy restatement in pylatticedso_amd/lattice_arrays.py).
//
// What the reference does with Python objects - Lattice.generate_lattice (lattice.py:421-483): for every cell, for every
// template strut, two end points frac * size + coordinate (cell.py:300-305), nodes and struts de-duplicated through
// coordinates rounded to 9 decimals, first creator wins (cell.py:312-368), nodes indexed in coordinate order and struts
// in (lower end, upper end) order (lattice.py:665-698) - is done here with flat arrays:
//   1. per axis, the sorted set of distinct rounded coordinates (a few hundred values: distinct (cell origin, cell size)
//      pairs x distinct template fractions), so a point is three small binary searches -> an integer triple;
//   2. a direct-address table over those triples keeps the FIRST creator of every node (atomic min on the creation
//      index); scanning the table in order numbers the nodes by (x, y, z) with no sort at all;
//   3. struts are bucketed by their lower end node (counting sort), each bucket (<= a few dozen entries) is ordered by
//      (upper end, creation index), which yields the de-duplicated struts already in their final order and the first
//      creator of each;
//   4. cell -> strut and cell -> node incidence by a small sort per cell.
// Every step is a parallel loop over cells, nodes or table slices.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <memory>
#include <new>
#include <vector>

#include "../../include/pylattice_hip.h"
#include "pl_parallel.h"

namespace {

using pl::parallel_for;
inline unsigned n_workers() { return pl::host_workers(); }

inline int64_t key9(double v) { return (int64_t)std::rint(v * 1e9); }   // np.round(v, 9), as an integer

struct Gen {
  int64_t C = 0;
  int32_t nb = 0, G = 0;
  const double *coord = nullptr, *size = nullptr, *radii = nullptr, *tmpl = nullptr;
  const int32_t *ttype = nullptr;
  inline double point(int64_t c, int32_t s, int e, int ax) const {
    return tmpl[6 * (int64_t)s + 3 * e + ax] * size[3 * c + ax] + coord[3 * c + ax];
  }
};

}  // namespace

struct pl_lattice {
  std::vector<double> node_xyz, beam_radius;
  std::vector<int32_t> beam_conn, beam_type, beam_cell0, pid, bid;
  std::vector<int64_t> cell_beam_ptr, cell_beam_idx, cell_node_ptr, cell_node_idx;
};

static int generate_impl(int64_t n_cells, const double *cell_coord, const double *cell_size, const double *cell_radii,
                         int32_t n_geom, int32_t n_tmpl, const double *tmpl, const int32_t *tmpl_type,
                         std::unique_ptr<pl_lattice> &L_out, pl_lattice_info_t *info) {
  if (!info || !info || n_cells <= 0 || n_tmpl <= 0 || n_geom <= 0 || !cell_coord || !cell_size || !cell_radii ||
      !tmpl || !tmpl_type)
    return PL_ERR_ARG;
  Gen g;
  g.C = n_cells;
  g.nb = n_tmpl;
  g.G = n_geom;
  g.coord = cell_coord;
  g.size = cell_size;
  g.radii = cell_radii;
  g.tmpl = tmpl;
  g.ttype = tmpl_type;
  const int64_t C = n_cells, nb = n_tmpl;
  const int64_t n_created = C * nb;
  if (n_created * 2 >= (1LL << 31)) return PL_ERR_ARG;   // creation indices are int32

  // ---- 1. distinct rounded coordinates per axis
  std::vector<int64_t> U[3];
  for (int ax = 0; ax < 3; ++ax) {
    std::vector<std::pair<double, double>> cs((size_t)C);
    for (int64_t c = 0; c < C; ++c) cs[c] = {cell_coord[3 * c + ax], cell_size[3 * c + ax]};
    std::sort(cs.begin(), cs.end());
    cs.erase(std::unique(cs.begin(), cs.end()), cs.end());
    std::vector<double> fr;
    for (int64_t s = 0; s < nb; ++s) {
      fr.push_back(tmpl[6 * s + ax]);
      fr.push_back(tmpl[6 * s + 3 + ax]);
    }
    std::sort(fr.begin(), fr.end());
    fr.erase(std::unique(fr.begin(), fr.end()), fr.end());
    auto &u = U[ax];
    u.reserve(cs.size() * fr.size());
    for (const auto &p : cs)
      for (double f : fr) u.push_back(key9(f * p.second + p.first));
    std::sort(u.begin(), u.end());
    u.erase(std::unique(u.begin(), u.end()), u.end());
  }
  const int64_t S0 = (int64_t)U[0].size(), S1 = (int64_t)U[1].size(), S2 = (int64_t)U[2].size();
  // The direct-address table costs 4 + 4 bytes per slot (creator + node id).  It pays while the lattice fills a fair share
  // of it: a regular lattice occupies ~1/20 of its slots at worst (Octet: 4 nodes per cell in a 2x2x2-per-cell grid of
  // distinct coordinates; graded cell sizes multiply the distinct coordinates).  Beyond 64 slots per created strut end
  // (or 3e9 slots) the table would be mostly empty - tens of GB for an irregular lattice: the caller takes the numpy path.
  if ((double)S0 * (double)S1 * (double)S2 > std::min(3.0e9, 64.0 * 2.0 * (double)n_created + 1.0e6)) return PL_ERR_STATE;
  const int64_t T = S0 * S1 * S2;
  auto rank_of = [&](int ax, double v) -> int64_t {
    const auto &u = U[ax];
    return (int64_t)(std::lower_bound(u.begin(), u.end(), key9(v)) - u.begin());
  };

  // ---- 2. first creator of every node
  std::vector<std::atomic<int32_t>> table((size_t)T);
  parallel_for(T, [&](int64_t b, int64_t e, unsigned) {
    for (int64_t i = b; i < e; ++i) table[i].store(INT32_MAX, std::memory_order_relaxed);
  });
  std::vector<int64_t> slot((size_t)n_created * 2);      // table slot of every created strut end
  parallel_for(C, [&](int64_t cb, int64_t ce, unsigned) {
    for (int64_t c = cb; c < ce; ++c)
      for (int32_t s = 0; s < nb; ++s)
        for (int e = 0; e < 2; ++e) {
          const int64_t k = (c * nb + s) * 2 + e;
          const int64_t t = (rank_of(0, g.point(c, s, e, 0)) * S1 + rank_of(1, g.point(c, s, e, 1))) * S2 +
                            rank_of(2, g.point(c, s, e, 2));
          slot[k] = t;
          int32_t cur = table[t].load(std::memory_order_relaxed);
          while ((int32_t)k < cur && !table[t].compare_exchange_weak(cur, (int32_t)k, std::memory_order_relaxed)) {
          }
        }
  });
  // node ids = rank of the occupied slots in table (= lexicographic) order
  const unsigned W = n_workers();
  std::vector<int64_t> part(W + 1, 0);
  parallel_for(W, [&](int64_t wb, int64_t we, unsigned) {
    for (int64_t w = wb; w < we; ++w) {
      int64_t cnt = 0;
      for (int64_t i = T * w / W; i < T * (w + 1) / W; ++i) cnt += table[i].load(std::memory_order_relaxed) != INT32_MAX;
      part[w + 1] = cnt;
    }
  }, 1);
  for (unsigned w = 0; w < W; ++w) part[w + 1] += part[w];
  const int64_t N = part[W];
  if (N >= (1LL << 31) - 64) return PL_ERR_ARG;
  L_out.reset(new pl_lattice());
  pl_lattice *L = L_out.get();
  L->node_xyz.resize((size_t)N * 3);
  std::vector<int32_t> node_of_slot((size_t)T);
  parallel_for(W, [&](int64_t wb, int64_t we, unsigned) {
    for (int64_t w = wb; w < we; ++w) {
      int64_t id = part[w];
      for (int64_t i = T * w / W; i < T * (w + 1) / W; ++i) {
        const int32_t k = table[i].load(std::memory_order_relaxed);
        if (k == INT32_MAX) continue;
        node_of_slot[i] = (int32_t)id;
        const int64_t cs = k / 2, c = cs / nb;
        const int32_t s = (int32_t)(cs % nb);
        for (int ax = 0; ax < 3; ++ax) L->node_xyz[3 * id + ax] = g.point(c, s, k & 1, ax);
        ++id;
      }
    }
  }, 1);
  L->pid.resize((size_t)n_created * 2);
  parallel_for(n_created * 2, [&](int64_t b, int64_t e, unsigned) {
    for (int64_t k = b; k < e; ++k) L->pid[k] = node_of_slot[slot[k]];
  });
  { std::vector<int64_t>().swap(slot); std::vector<int32_t>().swap(node_of_slot); }

  // ---- 3. struts bucketed by lower end node
  std::vector<int64_t> bptr((size_t)N + 1, 0);
  {
    std::vector<std::atomic<int32_t>> cnt((size_t)N);
    parallel_for(N, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) cnt[i].store(0, std::memory_order_relaxed);
    });
    parallel_for(n_created, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t k = b; k < e; ++k)
        cnt[std::min(L->pid[2 * k], L->pid[2 * k + 1])].fetch_add(1, std::memory_order_relaxed);
    });
    for (int64_t i = 0; i < N; ++i) bptr[i + 1] = bptr[i] + cnt[i].load(std::memory_order_relaxed);
    parallel_for(N, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) cnt[i].store(0, std::memory_order_relaxed);
    });
    struct Ent {
      int32_t hi, k;
    };
    std::vector<Ent> ent((size_t)n_created);
    parallel_for(n_created, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t k = b; k < e; ++k) {
        const int32_t p = L->pid[2 * k], q = L->pid[2 * k + 1];
        const int32_t lo = std::min(p, q);
        ent[bptr[lo] + cnt[lo].fetch_add(1, std::memory_order_relaxed)] = {std::max(p, q), (int32_t)k};
      }
    });
    std::vector<int64_t> ucnt((size_t)N + 1, 0);
    parallel_for(N, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) {
        Ent *x0 = ent.data() + bptr[i], *x1 = ent.data() + bptr[i + 1];
        std::sort(x0, x1, [](const Ent &l, const Ent &r) { return l.hi < r.hi || (l.hi == r.hi && l.k < r.k); });
        int64_t u = 0;
        for (Ent *x = x0; x < x1; ++x) u += (x == x0 || x->hi != (x - 1)->hi);
        ucnt[i + 1] = u;
      }
    });
    for (int64_t i = 0; i < N; ++i) ucnt[i + 1] += ucnt[i];
    const int64_t B = ucnt[N];
    if (B >= (1LL << 31) - 64) {
      delete L;
      return PL_ERR_ARG;
    }
    L->beam_conn.resize((size_t)B * 2);
    L->beam_radius.resize((size_t)B);
    L->beam_type.resize((size_t)B);
    L->beam_cell0.resize((size_t)B);
    L->bid.resize((size_t)n_created);
    parallel_for(N, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) {
        int64_t id = ucnt[i] - 1;
        for (int64_t q = bptr[i]; q < bptr[i + 1]; ++q) {
          const Ent &x = ent[q];
          if (q == bptr[i] || x.hi != ent[q - 1].hi) {   // first creator of this strut
            ++id;
            const int64_t k = x.k, c = k / nb;
            const int32_t s = (int32_t)(k % nb);
            L->beam_conn[2 * id] = L->pid[2 * k];
            L->beam_conn[2 * id + 1] = L->pid[2 * k + 1];
            L->beam_radius[id] = cell_radii[c * n_geom + tmpl_type[s]];
            L->beam_type[id] = tmpl_type[s];
            L->beam_cell0[id] = (int32_t)c;
          }
          L->bid[x.k] = (int32_t)id;
        }
      }
    });
  }

  // ---- 4. cell -> struts, cell -> nodes (sorted, unique)
  auto cell_csr = [&](const std::vector<int32_t> &items, int64_t per_cell, std::vector<int64_t> &ptr,
                      std::vector<int64_t> &idx) {
    ptr.assign((size_t)C + 1, 0);
    std::vector<int32_t> tmp(items);   // sorted in place per cell
    parallel_for(C, [&](int64_t cb, int64_t ce, unsigned) {
      for (int64_t c = cb; c < ce; ++c) {
        int32_t *x0 = tmp.data() + c * per_cell, *x1 = x0 + per_cell;
        std::sort(x0, x1);
        ptr[c + 1] = std::unique(x0, x1) - x0;
      }
    });
    for (int64_t c = 0; c < C; ++c) ptr[c + 1] += ptr[c];
    idx.resize((size_t)ptr[C]);
    parallel_for(C, [&](int64_t cb, int64_t ce, unsigned) {
      for (int64_t c = cb; c < ce; ++c) {
        const int32_t *x0 = tmp.data() + c * per_cell;
        for (int64_t q = 0; q < ptr[c + 1] - ptr[c]; ++q) idx[ptr[c] + q] = x0[q];
      }
    });
  };
  cell_csr(L->bid, nb, L->cell_beam_ptr, L->cell_beam_idx);
  cell_csr(L->pid, nb * 2, L->cell_node_ptr, L->cell_node_idx);

  info->n_nodes = N;
  info->n_beams = (int64_t)L->beam_radius.size();
  info->n_cell_beam = (int64_t)L->cell_beam_idx.size();
  info->n_cell_node = (int64_t)L->cell_node_idx.size();
  info->n_created = n_created;
  return PL_OK;
}

extern "C" {

int pl_generate_lattice(int64_t n_cells, const double *cell_coord, const double *cell_size, const double *cell_radii,
                        int32_t n_geom, int32_t n_tmpl, const double *tmpl, const int32_t *tmpl_type,
                        pl_lattice **out, pl_lattice_info_t *info) {
  if (!out) return PL_ERR_ARG;
  *out = nullptr;
  std::unique_ptr<pl_lattice> L;
  try {   // no C++ exception may cross the C ABI: out of host memory = "declined", the caller falls back to numpy
    const int rc = generate_impl(n_cells, cell_coord, cell_size, cell_radii, n_geom, n_tmpl, tmpl, tmpl_type, L, info);
    if (rc != PL_OK) return rc;
  } catch (const std::bad_alloc &) {
    return PL_ERR_STATE;
  } catch (...) {
    return PL_ERR_STATE;
  }
  *out = L.release();
  return PL_OK;
}

int pl_lattice_fetch(const pl_lattice *L, double *node_xyz, int32_t *beam_conn, double *beam_radius, int32_t *beam_type,
                     int32_t *beam_cell0, int64_t *cell_beam_ptr, int64_t *cell_beam_idx, int64_t *cell_node_ptr,
                     int64_t *cell_node_idx, int32_t *created_nodes, int32_t *created_beam) {
  if (!L) return PL_ERR_ARG;
  auto cp = [](auto *dst, const auto &v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cp(node_xyz, L->node_xyz);
  cp(beam_conn, L->beam_conn);
  cp(beam_radius, L->beam_radius);
  cp(beam_type, L->beam_type);
  cp(beam_cell0, L->beam_cell0);
  cp(cell_beam_ptr, L->cell_beam_ptr);
  cp(cell_beam_idx, L->cell_beam_idx);
  cp(cell_node_ptr, L->cell_node_ptr);
  cp(cell_node_idx, L->cell_node_idx);
  cp(created_nodes, L->pid);
  cp(created_beam, L->bid);
  return PL_OK;
}

void pl_lattice_free(pl_lattice *L) { delete L; }

// LatticeSim.set_penalized_beams (lattice_sim.py:245-308) + Beam.get_point_on_beam_at_distance (beam.py:279-326) +
// the gmsh subdivision count of every segment (lattice_generation.py:50-64), per strut, multi-threaded.  The new points
// sit at  end + (other - end) / round(length, 4) * L_zone  - the reference divides by Beam.length, which is rounded to
// 4 decimals with Python's round(), i.e. the correctly rounded decimal: printf("%.4f") gives the same digits.
static int boundary_index_impl(int64_t n_cells, const int64_t *cell_node_ptr, const int64_t *cell_node_idx, int64_t n_nodes,
                               const double *node_xyz, const double *cell_coord, const double *cell_size,
                               int64_t *index_boundary, int64_t *visit, int64_t *n_visit, bool by_coordinates);

int pl_boundary_index(int64_t n_cells, const int64_t *cell_node_ptr, const int64_t *cell_node_idx, int64_t n_nodes,
                      const double *node_xyz, const double *cell_coord, const double *cell_size, int64_t *index_boundary,
                      int64_t *visit, int64_t *n_visit) {
  return boundary_index_impl(n_cells, cell_node_ptr, cell_node_idx, n_nodes, node_xyz, cell_coord, cell_size, index_boundary,
                             visit, n_visit, false);
}
int pl_boundary_index_rows(int64_t n_cells, const int64_t *cell_node_ptr, const int64_t *cell_node_idx, int64_t n_nodes,
                           const double *node_xyz, const double *cell_coord, const double *cell_size,
                           int64_t *index_boundary, int64_t *visit, int64_t *n_visit) {
  return boundary_index_impl(n_cells, cell_node_ptr, cell_node_idx, n_nodes, node_xyz, cell_coord, cell_size, index_boundary,
                             visit, n_visit, true);
}

static int boundary_index_impl(int64_t n_cells, const int64_t *cell_node_ptr, const int64_t *cell_node_idx, int64_t n_nodes,
                               const double *node_xyz, const double *cell_coord, const double *cell_size,
                               int64_t *index_boundary, int64_t *visit, int64_t *n_visit, bool by_coordinates) {
  if (n_cells <= 0 || n_nodes <= 0 || !cell_node_ptr || !cell_node_idx || !node_xyz || !cell_coord || !cell_size ||
      !index_boundary || !visit || !n_visit)
    return PL_ERR_ARG;
  try {
    // by_coordinates (the reference's own rows, LatticeSim(reference_compat=True): design nodes and penalisation points of a
    // cell interleave): inside a cell the rows are visited in (round(x, 9), round(y, 9), round(z, 9), index) order
    // (_sorted_nodes, lattice_sim.py:193-199; numpy's round(x, 9) is rint(x 1e9) / 1e9, so rint(x 1e9) orders alike) -
    // every cell sorted on its own, all cells in parallel
    std::vector<int64_t> sorted_rows;
    if (by_coordinates) {
      sorted_rows.assign(cell_node_idx, cell_node_idx + cell_node_ptr[n_cells]);
      std::atomic<int> bad_idx{0};
      parallel_for(n_cells, [&](int64_t cb, int64_t ce, unsigned) {
        for (int64_t c = cb; c < ce; ++c) {
          int64_t *r0 = sorted_rows.data() + cell_node_ptr[c], *r1 = sorted_rows.data() + cell_node_ptr[c + 1];
          for (int64_t *q = r0; q < r1; ++q)
            if (*q < 0 || *q >= n_nodes) bad_idx = 1;
          if (bad_idx) return;
          std::sort(r0, r1, [&](int64_t a, int64_t b) {
            for (int k = 0; k < 3; ++k) {
              const double ka = std::nearbyint(node_xyz[3 * a + k] * 1e9), kb = std::nearbyint(node_xyz[3 * b + k] * 1e9);
              if (ka != kb) return ka < kb;
            }
            return a < b;
          });
        }
      }, 256);
      if (bad_idx) return PL_ERR_ARG;
      cell_node_idx = sorted_rows.data();
    }
    std::vector<std::atomic<uint8_t>> on_box((size_t)n_nodes);
    parallel_for(n_nodes, [&](int64_t b, int64_t e, unsigned) {
      for (int64_t i = b; i < e; ++i) {
        on_box[i].store(0, std::memory_order_relaxed);
        index_boundary[i] = -1;
      }
    }, 1 << 16);
    std::atomic<int> bad{0};
    parallel_for(n_cells, [&](int64_t cb, int64_t ce, unsigned) {
      for (int64_t c = cb; c < ce; ++c) {
        const double lo[3] = {cell_coord[3 * c], cell_coord[3 * c + 1], cell_coord[3 * c + 2]};
        const double hi[3] = {lo[0] + cell_size[3 * c], lo[1] + cell_size[3 * c + 1], lo[2] + cell_size[3 * c + 2]};
        for (int64_t q = cell_node_ptr[c]; q < cell_node_ptr[c + 1]; ++q) {
          const int64_t i = cell_node_idx[q];
          if (i < 0 || i >= n_nodes) { bad = 1; continue; }
          const double *x = node_xyz + 3 * i;
          if (x[0] == lo[0] || x[0] == hi[0] || x[1] == lo[1] || x[1] == hi[1] || x[2] == lo[2] || x[2] == hi[2])
            on_box[i].store(1, std::memory_order_relaxed);
        }
      }
    }, 256);
    if (bad) return PL_ERR_ARG;
    // visit order: cells in order, inside a cell by node index (= coordinate order), first visit counts
    int64_t nv = 0;
    std::vector<int64_t> row;
    for (int64_t c = 0; c < n_cells; ++c) {
      const int64_t q0 = cell_node_ptr[c], q1 = cell_node_ptr[c + 1];
      const int64_t *r = cell_node_idx + q0;
      if (!by_coordinates && !std::is_sorted(r, r + (q1 - q0))) {
        row.assign(r, r + (q1 - q0));
        std::sort(row.begin(), row.end());
        r = row.data();
      }
      for (int64_t q = 0; q < q1 - q0; ++q) {
        const int64_t i = r[q];
        if (on_box[i].load(std::memory_order_relaxed) && index_boundary[i] < 0) {
          index_boundary[i] = nv;
          visit[nv++] = i;
        }
      }
    }
    *n_visit = nv;
  } catch (...) {
    return PL_ERR_STATE;
  }
  return PL_OK;
}

int pl_penalize(int64_t n_beams, const double *node_xyz, const int32_t *beam_conn, const double *lzone /*may be null*/,
                double mesh_size, double *seg_len, int32_t *seg_nsub, double *pen_xyz) {
  if (n_beams < 0 || !node_xyz || !beam_conn || !seg_len || !seg_nsub || !pen_xyz || !(mesh_size > 0.0))
    return PL_ERR_ARG;
  const double nan = std::nan("");
  parallel_for(n_beams, [&](int64_t b0, int64_t b1, unsigned) {
    char buf[64];
    double last_len = -1.0, last_len4 = 0.0;
    for (int64_t b = b0; b < b1; ++b) {
      const double *pa = node_xyz + 3 * (int64_t)beam_conn[2 * b], *pb = node_xyz + 3 * (int64_t)beam_conn[2 * b + 1];
      const double d[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
      const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      if (len != last_len) {
        std::snprintf(buf, sizeof buf, "%.4f", len);
        last_len4 = std::strtod(buf, nullptr);
        last_len = len;
      }
      const double len4 = last_len4;
      const double L1 = lzone ? lzone[2 * b] : 0.0, L2 = lzone ? lzone[2 * b + 1] : 0.0;
      const bool has1 = L1 > 0.0, has2 = L2 > 0.0;
      double q1[3], q2[3], st[3], en[3];
      for (int k = 0; k < 3; ++k) {
        q1[k] = pa[k] + (d[k] / len4) * L1;
        q2[k] = pb[k] + ((-d[k]) / len4) * L2;
        st[k] = has1 ? q1[k] : pa[k];
        en[k] = has2 ? q2[k] : pb[k];
      }
      auto dist = [](const double *u, const double *v) {
        const double w0 = v[0] - u[0], w1 = v[1] - u[1], w2 = v[2] - u[2];
        return std::sqrt(w0 * w0 + w1 * w1 + w2 * w2);
      };
      const double sl[3] = {has1 ? dist(pa, q1) : 0.0, dist(st, en), has2 ? dist(q2, pb) : 0.0};
      for (int k = 0; k < 3; ++k) {
        seg_len[3 * b + k] = sl[k];
        int32_t n = 0;
        if (sl[k] > 0.0) n = std::max<int32_t>(1, (int32_t)std::floor(sl[k] / mesh_size + 0.99));
        seg_nsub[3 * b + k] = n;
        pen_xyz[6 * b + k] = has1 ? q1[k] : nan;
        pen_xyz[6 * b + 3 + k] = has2 ? q2[k] : nan;
      }
    }
  });
  return PL_OK;
}

}  // extern "C"
