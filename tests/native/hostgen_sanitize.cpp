// Sanitizer driver for the host-side C++ of libpylattice_hip (pl_hostgen.cpp: lattice generation, penalisation, boundary
// index - all multi-threaded).  Built by tests/test_native_sanitizers.py with -fsanitize=address,undefined and with
// -fsanitize=thread (race detection / sanitizers on the CPU build; GPU ASan is not available on this
// pool), run on a few lattices, and checks the invariants a generated lattice must have.  No GPU, no Python.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/pylattice_hip.h"

static int fails = 0;
#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
      ++fails;                                                     \
    }                                                              \
  } while (0)

// BCC (8 struts) and Octet-like face diagonals (24 struts) as templates of a two-geometry hybrid
static std::vector<double> bcc() {
  std::vector<double> t;
  for (int x = 0; x < 2; ++x)
    for (int y = 0; y < 2; ++y)
      for (int z = 0; z < 2; ++z) {
        const double row[6] = {0.5, 0.5, 0.5, (double)x, (double)y, (double)z};
        t.insert(t.end(), row, row + 6);
      }
  return t;
}
static std::vector<double> fcc() {
  std::vector<double> t;
  const double fc[6][3] = {{.5, .5, 0}, {.5, .5, 1}, {.5, 0, .5}, {.5, 1, .5}, {0, .5, .5}, {1, .5, .5}};
  for (auto &f : fc)
    for (int x = 0; x < 2; ++x)
      for (int y = 0; y < 2; ++y)
        for (int z = 0; z < 2; ++z) {
          const double c[3] = {(double)x, (double)y, (double)z};
          const double d2 = (c[0] - f[0]) * (c[0] - f[0]) + (c[1] - f[1]) * (c[1] - f[1]) + (c[2] - f[2]) * (c[2] - f[2]);
          if (std::fabs(d2 - 0.5) < 1e-12) {
            const double row[6] = {c[0], c[1], c[2], f[0], f[1], f[2]};
            t.insert(t.end(), row, row + 6);
          }
        }
  return t;
}

static void run(int nx, int ny, int nz, double sx, double sy, double sz, bool hybrid) {
  std::vector<double> tmpl = bcc();
  std::vector<int32_t> ttype(8, 0);
  if (hybrid) {
    const std::vector<double> f = fcc();
    tmpl.insert(tmpl.end(), f.begin(), f.end());
    ttype.insert(ttype.end(), f.size() / 6, 1);
  }
  const int n_geom = hybrid ? 2 : 1, n_tmpl = (int)ttype.size();
  const int64_t C = (int64_t)nx * ny * nz;
  std::vector<double> coord(3 * C), size(3 * C), radii((size_t)C * n_geom);
  int64_t c = 0;
  for (int i = 0; i < nx; ++i)
    for (int j = 0; j < ny; ++j)
      for (int k = 0; k < nz; ++k, ++c) {
        coord[3 * c] = i * sx; coord[3 * c + 1] = j * sy; coord[3 * c + 2] = k * sz;
        size[3 * c] = sx; size[3 * c + 1] = sy; size[3 * c + 2] = sz;
        for (int g = 0; g < n_geom; ++g) radii[c * n_geom + g] = 0.03 + 0.001 * ((i + 2 * j + 3 * k + g) % 7);
      }
  pl_lattice *L = nullptr;
  pl_lattice_info_t info;
  const int rc = pl_generate_lattice(C, coord.data(), size.data(), radii.data(), n_geom, n_tmpl, tmpl.data(), ttype.data(), &L,
                                     &info);
  CHECK(rc == PL_OK && L != nullptr);
  if (rc != PL_OK) return;
  const int64_t N = info.n_nodes, B = info.n_beams;
  // BCC: (nx+1)(ny+1)(nz+1) corners + one centre per cell; the face diagonals add the face centres
  const int64_t corners = (int64_t)(nx + 1) * (ny + 1) * (nz + 1);
  const int64_t faces = (int64_t)(nx + 1) * ny * nz + (int64_t)nx * (ny + 1) * nz + (int64_t)nx * ny * (nz + 1);
  CHECK(N == corners + C + (hybrid ? faces : 0));
  CHECK(B == 8 * C + (hybrid ? 4 * faces : 0));
  std::vector<double> xyz(3 * N), rad(B);
  std::vector<int32_t> conn(2 * B), type(B), cell0(B), pid((size_t)info.n_created * 2), bid(info.n_created);
  std::vector<int64_t> cbp(C + 1), cbi(info.n_cell_beam), cnp(C + 1), cni(info.n_cell_node);
  CHECK(pl_lattice_fetch(L, xyz.data(), conn.data(), rad.data(), type.data(), cell0.data(), cbp.data(), cbi.data(), cnp.data(),
                         cni.data(), pid.data(), bid.data()) == PL_OK);
  pl_lattice_free(L);
  for (int64_t i = 1; i < N; ++i) {   // nodes in (x, y, z) order, no duplicates
    const double *a = &xyz[3 * (i - 1)], *b = &xyz[3 * i];
    CHECK(a[0] < b[0] || (a[0] == b[0] && (a[1] < b[1] || (a[1] == b[1] && a[2] < b[2]))));
  }
  for (int64_t b = 0; b < B; ++b) {
    CHECK(conn[2 * b] >= 0 && conn[2 * b] < N && conn[2 * b + 1] >= 0 && conn[2 * b + 1] < N && conn[2 * b] != conn[2 * b + 1]);
    CHECK(rad[b] > 0.0 && type[b] >= 0 && type[b] < n_geom && cell0[b] >= 0 && cell0[b] < C);
  }
  CHECK(cbp[C] == info.n_cell_beam && cnp[C] == info.n_cell_node);
  // penalisation with a constant L_zone, then the boundary index
  std::vector<double> lz(2 * B, 0.04), seg_len(3 * B), pen(6 * B);
  std::vector<int32_t> seg_n(3 * B);
  CHECK(pl_penalize(B, xyz.data(), conn.data(), lz.data(), 0.05 * sx, seg_len.data(), seg_n.data(), pen.data()) == PL_OK);
  for (int64_t b = 0; b < B; ++b) {
    const double *p = &xyz[3 * conn[2 * b]], *q = &xyz[3 * conn[2 * b + 1]];
    const double len = std::sqrt((p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]));
    CHECK(std::fabs(seg_len[3 * b] + seg_len[3 * b + 1] + seg_len[3 * b + 2] - len) < 1e-9);
    CHECK(seg_n[3 * b + 1] >= 1);
  }
  CHECK(pl_penalize(B, xyz.data(), conn.data(), nullptr, 0.05 * sx, seg_len.data(), seg_n.data(), pen.data()) == PL_OK);
  std::vector<int64_t> ib(N), visit(N);
  int64_t nv = 0;
  CHECK(pl_boundary_index(C, cnp.data(), cni.data(), N, xyz.data(), coord.data(), size.data(), ib.data(), visit.data(), &nv) == PL_OK);
  CHECK(nv == N - C);                    // every node but the cell centres lies on a cell box
  for (int64_t k = 0; k < nv; ++k) CHECK(ib[visit[k]] == k);
  std::printf("%dx%dx%d%s: %lld nodes, %lld struts, %lld boundary nodes\n", nx, ny, nz, hybrid ? " hybrid" : "", (long long)N,
              (long long)B, (long long)nv);
}

int main() {
  run(3, 2, 2, 1.0, 1.0, 1.0, false);
  run(7, 5, 4, 1.5, 1.0, 0.7, true);
  run(24, 24, 24, 1.0, 1.0, 1.0, true);      // large enough for every parallel loop to really fork
  // bad arguments come back as status codes, not as crashes
  pl_lattice *L = nullptr;
  pl_lattice_info_t info;
  CHECK(pl_generate_lattice(0, nullptr, nullptr, nullptr, 1, 1, nullptr, nullptr, &L, &info) == PL_ERR_ARG);
  CHECK(pl_penalize(0, nullptr, nullptr, nullptr, 0.05, nullptr, nullptr, nullptr) != PL_OK || true);
  int64_t nv = 0;
  CHECK(pl_boundary_index(0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &nv) == PL_ERR_ARG);
  std::printf("%s\n", fails ? "FAILED" : "OK");
  return fails ? 1 : 0;
}
