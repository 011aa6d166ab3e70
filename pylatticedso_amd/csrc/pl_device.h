// Device-side element arithmetic shared by every kernel of libpylattice_hip (gfx950 only).
//
// One lattice strut = up to three colinear segments [pen(1.5 r) | r | pen(1.5 r)], each meshed by gmsh into n equal
// P1xP1 Timoshenko sub-elements with mid-point shear integration (reference: simulation_base.py:141-156,190-225,
// lattice_generation.py:50-101, lattice_sim.py:245-308).  A chain of n such sub-elements condenses exactly to a
// 2-node element with tip flexibility
//     axial   L/(ES)      torsion L/(GJ)
//     bending [[L/(kGS) + L^3/(3EI) (1 - 1/(4 n^2)),  L^2/(2EI)], [L^2/(2EI), L/(EI)]]
// and the three segments compose in series by transporting each segment's flexibility to end B
// (checked against the sub-meshed model by tests/test_oracle_golden.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pl {

struct Material {
  double E, G, kappa, pen;
};

// k identical struts in parallel between the same two nodes (the reference's per-cell copies of a strut shared by k cells,
// lattice_sim.py:250-303) are one strut of a k times stiffer material: every flexibility is proportional to 1/E, 1/G.
__device__ __forceinline__ Material scaled(Material m, double k) {
  m.E *= k;
  m.G *= k;
  return m;
}

// Condensed stiffness scalars of one strut.
struct Scalars {
  double ka, kt, a, b, c;
};

// Record the SpMV kernels read: 8 doubles = 64 B per strut.
//   F_B = a du + e1 (du.d) d + e2 (d x dth),  M_B = c dth + e3 (dth.d) d - e2 (d x du)
//   du = uB - uA + d x thA, dth = thB - thA, d = xB - xA
//   e1 = (ka - a)/L^2, e2 = b/L, e3 = (kt - c)/L^2
struct __attribute__((aligned(16))) Record {
  double a, c, e1, e2, e3, dx, dy, dz;
};

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Flexibility sums of the strut, split by their power of the radius:
//   shear/axial terms scale as r^-2, bending/torsion terms as r^-4 (used by the sensitivity kernel).
struct Flex {
  double fa;                 // axial            ~ r^-2
  double ft;                 // torsion          ~ r^-4
  double s11;                // shear part of f11 ~ r^-2
  double b11, b12, b22;      // bending parts     ~ r^-4
};

__device__ __forceinline__ Flex strut_flexibility(double r, const double *len, const int *nsub, const Material &m) {
  const double PI = 3.14159265358979323846;
  Flex f = {0, 0, 0, 0, 0, 0};
  const double L = len[0] + len[1] + len[2];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double l = len[i];
    if (l > 0.0) {
      const double R = (i == 1) ? r : m.pen * r;
      const double S = PI * R * R;
      const double I = 0.25 * PI * R * R * R * R;
      const double ES = m.E * S, GS = m.G * m.kappa * S, EI = m.E * I, GJ = m.G * 2.0 * I;
      const double n = (double)nsub[i];
      const double g11s = l / GS;
      const double g11b = l * l * l / (3.0 * EI) * (1.0 - 1.0 / (4.0 * n * n));
      const double g12 = l * l / (2.0 * EI);
      const double g22 = l / EI;
      const double d = L - (s + l);   // distance from this segment's end to end B
      f.fa += l / ES;
      f.ft += l / GJ;
      f.s11 += g11s;
      f.b11 += g11b + 2.0 * d * g12 + d * d * g22;
      f.b12 += g12 + d * g22;
      f.b22 += g22;
      s += l;
    }
  }
  return f;
}

__device__ __forceinline__ Scalars scalars_from_flex(const Flex &f) {
  const double f11 = f.s11 + f.b11, f12 = f.b12, f22 = f.b22;
  const double det = f11 * f22 - f12 * f12;
  return {1.0 / f.fa, 1.0 / f.ft, f22 / det, f12 / det, f11 / det};
}

// d(scalars)/dr at fixed segment geometry:  dF/dr = -(2/r) F_shear - (4/r) F_bend,  dK = -K dF K.
__device__ __forceinline__ Scalars dscalars_dr(const Flex &f, double r) {
  const Scalars k = scalars_from_flex(f);
  const double d11 = -(2.0 / r) * f.s11 - (4.0 / r) * f.b11;
  const double d12 = -(4.0 / r) * f.b12;
  const double d22 = -(4.0 / r) * f.b22;
  // K2 = [[a, -b], [-b, c]]; dK2 = -K2 dF K2
  const double m11 = k.a * d11 - k.b * d12, m12 = k.a * d12 - k.b * d22;      // (K2 dF) row 1
  const double m21 = -k.b * d11 + k.c * d12, m22 = -k.b * d12 + k.c * d22;    // (K2 dF) row 2
  const double n11 = -(m11 * k.a - m12 * k.b);
  const double n12 = -(-m11 * k.b + m12 * k.c);
  const double n22 = -(-m21 * k.b + m22 * k.c);
  (void)m21;
  Scalars d;
  d.ka = k.ka * 2.0 / r;
  d.kt = k.kt * 4.0 / r;
  d.a = n11;
  d.b = -n12;
  d.c = n22;
  return d;
}

__device__ __forceinline__ Record make_record(const Scalars &k, V3 d) {
  const double L2 = dot(d, d);
  const double L = sqrt(L2);
  Record r;
  r.a = k.a;
  r.c = k.c;
  r.e1 = (k.ka - k.a) / L2;
  r.e2 = k.b / L;
  r.e3 = (k.kt - k.c) / L2;
  r.dx = d.x;
  r.dy = d.y;
  r.dz = d.z;
  return r;
}

// The same strut seen from its other end (A becomes the "tip"): d -> -d, b -> aL - b, c -> c + aL^2 - 2bL.
__device__ __forceinline__ Record reversed(const Record &r) {
  const double L2 = r.dx * r.dx + r.dy * r.dy + r.dz * r.dz;
  const double g = r.a - 2.0 * r.e2;
  Record q;
  q.a = r.a;
  q.c = r.c + L2 * g;
  q.e1 = r.e1;
  q.e2 = r.a - r.e2;
  q.e3 = r.e3 - g;
  q.dx = -r.dx;
  q.dy = -r.dy;
  q.dz = -r.dz;
  return q;
}

// Force / moment the strut applies to its tip end B, for end values (uA, thA) and (uB, thB).
__device__ __forceinline__ void tip_force(const Record &r, V3 uA, V3 thA, V3 uB, V3 thB, V3 &F, V3 &M) {
  const V3 d = {r.dx, r.dy, r.dz};
  const V3 du = uB - uA + cross(d, thA);
  const V3 dth = thB - thA;
  F = r.a * du + (r.e1 * dot(du, d)) * d + r.e2 * cross(d, dth);
  M = r.c * dth + (r.e3 * dot(dth, d)) * d - r.e2 * cross(d, du);
}

// Diagonal of the tip-end 6x6 block.
__device__ __forceinline__ void tip_diag(const Record &r, double *dg) {
  dg[0] = r.a + r.e1 * r.dx * r.dx;
  dg[1] = r.a + r.e1 * r.dy * r.dy;
  dg[2] = r.a + r.e1 * r.dz * r.dz;
  dg[3] = r.c + r.e3 * r.dx * r.dx;
  dg[4] = r.c + r.e3 * r.dy * r.dy;
  dg[5] = r.c + r.e3 * r.dz * r.dz;
}

// 6x6 blocks of the tip row: Kss (tip,tip) and Kso (tip,other), row-major.
__device__ __forceinline__ void tip_blocks(const Record &r, double *Kss, double *Kso) {
  const double d[3] = {r.dx, r.dy, r.dz};
  // D = skew(d): D v = d x v
  const double D[3][3] = {{0, -d[2], d[1]}, {d[2], 0, -d[0]}, {-d[1], d[0], 0}};
  double Kuu[3][3], Ktt[3][3], Kut[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Kuu[i][j] = (i == j ? r.a : 0.0) + r.e1 * d[i] * d[j];
      Ktt[i][j] = (i == j ? r.c : 0.0) + r.e3 * d[i] * d[j];
      Kut[i][j] = r.e2 * D[i][j];          // F gets + e2 (d x dth);  M gets - e2 (d x du) = Kut^T du
    }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Kss[i * 6 + j] = Kuu[i][j];
      Kss[i * 6 + 3 + j] = Kut[i][j];
      Kss[(3 + i) * 6 + j] = -Kut[i][j];
      Kss[(3 + i) * 6 + 3 + j] = Ktt[i][j];
    }
  // Kso = -Kss R,  R = [[I, -D], [0, I]]   (rigid transport of the other end to the tip)
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 3; ++j) Kso[i * 6 + j] = -Kss[i * 6 + j];
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;   // (Kss[:, 0:3] * (-D))[i][j]
      for (int k = 0; k < 3; ++k) acc += Kss[i * 6 + k] * (-D[k][j]);
      Kso[i * 6 + 3 + j] = -(acc + Kss[i * 6 + 3 + j]);
    }
  }
}

}  // namespace pl
