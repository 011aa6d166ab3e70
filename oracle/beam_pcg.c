/*
 * oracle/beam_pcg.c — plain-C CPU restatement (TEST INFRASTRUCTURE / cpu_baseline ONLY; never linked or called
 * by the product path).  Same algorithm as oracle/timoshenko_oracle.py:
 *   - condensed strut scalars from the reference's sub-meshed Timoshenko model
 *     (simulation_base.py:141-156,190-225; lattice_generation.py:50-101; lattice_sim.py:245-308),
 *   - matrix-free K*x as the sum of per-strut products,
 *   - Dirichlet handling with dolfinx semantics (simulation_base.py:465-514),
 *   - Jacobi-preconditioned CG in the textbook form (x0 = 0, stop on ||r|| <= rtol ||b||).
 * Pinned against the numpy oracle (itself pinned by the reference's dolfinx Schur goldens) in
 * tests/test_oracle_c.py.  oracle_pcg is single-threaded (bitwise reproducible: the checker of smoke() and of the
 * tests); oracle_pcg_mt is the same Jacobi-PCG on all host cores (OpenMP; per-node gather over the incident struts
 * instead of the scatter, so no two threads write one row) - the "cores": C leg of bench.py's cpu_baseline.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI 3.14159265358979323846

void oracle_condense(double r, const double *len, const int32_t *nsub, double E, double nu, double kappa, double pen,
                     double *out5) {
  const double G = E / (2.0 * (1.0 + nu));
  const double L = len[0] + len[1] + len[2];
  double fa = 0, ft = 0, f11 = 0, f12 = 0, f22 = 0, s = 0;
  for (int i = 0; i < 3; ++i) {
    const double l = len[i];
    if (l <= 0.0) continue;
    const double R = (i == 1) ? r : pen * r;
    const double S = PI * R * R, I = 0.25 * PI * R * R * R * R;
    const double ES = E * S, GS = G * kappa * S, EI = E * I, GJ = G * 2.0 * I;
    const double n = (double)nsub[i];
    const double g11 = l / GS + l * l * l / (3.0 * EI) * (1.0 - 1.0 / (4.0 * n * n));
    const double g12 = l * l / (2.0 * EI), g22 = l / EI;
    const double d = L - (s + l);
    fa += l / ES;
    ft += l / GJ;
    f11 += g11 + 2.0 * d * g12 + d * d * g22;
    f12 += g12 + d * g22;
    f22 += g22;
    s += l;
  }
  const double det = f11 * f22 - f12 * f12;
  out5[0] = 1.0 / fa;
  out5[1] = 1.0 / ft;
  out5[2] = f22 / det;
  out5[3] = f12 / det;
  out5[4] = f11 / det;
}

static inline void cross3(const double *a, const double *b, double *c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* y += K x (y must be zeroed by the caller) */
void oracle_spmv_add(int64_t B, const double *xyz, const int32_t *conn, const double *sc, const double *x, double *y) {
  for (int64_t b = 0; b < B; ++b) {
    const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
    const double ka = sc[5 * b], kt = sc[5 * b + 1], a = sc[5 * b + 2], bb = sc[5 * b + 3], c = sc[5 * b + 4];
    double d[3] = {xyz[3 * ib] - xyz[3 * ia], xyz[3 * ib + 1] - xyz[3 * ia + 1], xyz[3 * ib + 2] - xyz[3 * ia + 2]};
    const double L = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double t[3] = {d[0] / L, d[1] / L, d[2] / L};
    const double *xa = x + 6 * ia, *xb = x + 6 * ib;
    double dxt[3], du[3], dth[3], tc[3], tu[3], F[3], M[3], dF[3];
    cross3(d, xa + 3, dxt);
    for (int k = 0; k < 3; ++k) {
      du[k] = xb[k] - xa[k] + dxt[k];
      dth[k] = xb[3 + k] - xa[3 + k];
    }
    const double dut = du[0] * t[0] + du[1] * t[1] + du[2] * t[2];
    const double dtt = dth[0] * t[0] + dth[1] * t[1] + dth[2] * t[2];
    cross3(t, dth, tc);
    cross3(t, du, tu);
    for (int k = 0; k < 3; ++k) {
      F[k] = (ka - a) * dut * t[k] + a * du[k] + bb * tc[k];
      M[k] = (kt - c) * dtt * t[k] + c * dth[k] - bb * tu[k];
    }
    cross3(d, F, dF);
    double *ya = y + 6 * ia, *yb = y + 6 * ib;
    for (int k = 0; k < 3; ++k) {
      yb[k] += F[k];
      yb[3 + k] += M[k];
      ya[k] -= F[k];
      ya[3 + k] += -M[k] - dF[k];
    }
  }
}

/* Jacobi diagonal of K */
static void oracle_diag(int64_t N, int64_t B, const double *xyz, const int32_t *conn, const double *sc, double *dg) {
  memset(dg, 0, sizeof(double) * 6 * N);
  for (int64_t b = 0; b < B; ++b) {
    const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
    const double ka = sc[5 * b], kt = sc[5 * b + 1], a = sc[5 * b + 2], bb = sc[5 * b + 3], c = sc[5 * b + 4];
    double d[3] = {xyz[3 * ib] - xyz[3 * ia], xyz[3 * ib + 1] - xyz[3 * ia + 1], xyz[3 * ib + 2] - xyz[3 * ia + 2]};
    const double L2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], L = sqrt(L2);
    const double cA = c + a * L2 - 2.0 * bb * L;
    for (int k = 0; k < 3; ++k) {
      const double t2 = d[k] * d[k] / L2;
      dg[6 * ib + k] += ka * t2 + a * (1.0 - t2);
      dg[6 * ia + k] += ka * t2 + a * (1.0 - t2);
      dg[6 * ib + 3 + k] += kt * t2 + c * (1.0 - t2);
      dg[6 * ia + 3 + k] += kt * t2 + cA * (1.0 - t2);
    }
  }
}

/* Solve K u = f with Dirichlet data; returns iterations (negative if not converged). */
int oracle_pcg(int64_t N, int64_t B, const double *xyz, const int32_t *conn, const double *sc, const uint8_t *fixed,
               const double *ubar, const double *f, double rtol, int maxit, double *u, double *relres_out) {
  const int64_t n = 6 * N;
  double *buf = (double *)calloc((size_t)n * 7, sizeof(double));
  if (!buf) return -1000000;
  double *x = buf, *r = buf + n, *z = buf + 2 * n, *p = buf + 3 * n, *Ap = buf + 4 * n, *dinv = buf + 5 * n,
         *ub = buf + 6 * n;
  oracle_diag(N, B, xyz, conn, sc, dinv);
  for (int64_t i = 0; i < n; ++i) {
    dinv[i] = (fixed[i] || dinv[i] == 0.0) ? 0.0 : 1.0 / dinv[i];
    ub[i] = fixed[i] ? ubar[i] : 0.0;
  }
  oracle_spmv_add(B, xyz, conn, sc, ub, Ap);   /* lifting */
  double bb = 0.0, rz = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    r[i] = fixed[i] ? 0.0 : f[i] - Ap[i];
    z[i] = dinv[i] * r[i];
    p[i] = z[i];
    bb += r[i] * r[i];
    rz += r[i] * z[i];
  }
  int it = 0, conv = 0;
  double rr = bb;
  if (bb > 0.0) {
    for (it = 1; it <= maxit; ++it) {
      memset(Ap, 0, sizeof(double) * n);
      oracle_spmv_add(B, xyz, conn, sc, p, Ap);
      double pAp = 0.0;
      for (int64_t i = 0; i < n; ++i) {
        if (fixed[i]) Ap[i] = 0.0;
        pAp += p[i] * Ap[i];
      }
      const double alpha = rz / pAp;
      double rz_new = 0.0;
      rr = 0.0;
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha * p[i];
        r[i] -= alpha * Ap[i];
        z[i] = dinv[i] * r[i];
        rz_new += r[i] * z[i];
        rr += r[i] * r[i];
      }
      if (rr <= rtol * rtol * bb) { conv = 1; break; }
      const double beta = rz_new / rz;
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
      rz = rz_new;
    }
  } else {
    conv = 1;
  }
  for (int64_t i = 0; i < n; ++i) u[i] = fixed[i] ? ubar[i] : x[i];
  if (relres_out) *relres_out = bb > 0.0 ? sqrt(rr / bb) : 0.0;
  free(buf);
  return conv ? it : -it;
}


/* the condensation of every strut ("assembly" of the matrix-free operator) on all cores */
void oracle_condense_all(int64_t B, const double *radius, const double *seg_len, const int32_t *seg_nsub, double E, double nu,
                         double kappa, double pen, double *out5) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) oracle_condense(radius[b], seg_len + 3 * b, seg_nsub + 3 * b, E, nu, kappa, pen, out5 + 5 * b);
}

/* ---- all-cores variant ------------------------------------------------------------------------------------ */
void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* force / moment strut b applies to its end `tip` (0: point1, 1: point2) for the field x */
static inline void strut_end_force(int64_t b, int tip, const double *xyz, const int32_t *conn, const double *sc,
                                   const double *x, double *out6) {
  const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
  const double ka = sc[5 * b], kt = sc[5 * b + 1], a = sc[5 * b + 2], bb = sc[5 * b + 3], c = sc[5 * b + 4];
  double d[3] = {xyz[3 * ib] - xyz[3 * ia], xyz[3 * ib + 1] - xyz[3 * ia + 1], xyz[3 * ib + 2] - xyz[3 * ia + 2]};
  const double L = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const double t[3] = {d[0] / L, d[1] / L, d[2] / L};
  const double *xa = x + 6 * ia, *xb = x + 6 * ib;
  double dxt[3], du[3], dth[3], tc[3], tu[3], F[3], M[3], dF[3];
  cross3(d, xa + 3, dxt);
  for (int k = 0; k < 3; ++k) {
    du[k] = xb[k] - xa[k] + dxt[k];
    dth[k] = xb[3 + k] - xa[3 + k];
  }
  const double dut = du[0] * t[0] + du[1] * t[1] + du[2] * t[2];
  const double dtt = dth[0] * t[0] + dth[1] * t[1] + dth[2] * t[2];
  cross3(t, dth, tc);
  cross3(t, du, tu);
  for (int k = 0; k < 3; ++k) {
    F[k] = (ka - a) * dut * t[k] + a * du[k] + bb * tc[k];
    M[k] = (kt - c) * dtt * t[k] + c * dth[k] - bb * tu[k];
  }
  if (tip) {
    for (int k = 0; k < 3; ++k) { out6[k] = F[k]; out6[3 + k] = M[k]; }
  } else {
    cross3(d, F, dF);
    for (int k = 0; k < 3; ++k) { out6[k] = -F[k]; out6[3 + k] = -M[k] - dF[k]; }
  }
}

/* y = mask .* (K x) by rows: ptr/inc = CSR node -> (strut << 1 | end) */
static void spmv_gather(int64_t N, const int64_t *ptr, const int64_t *inc, const double *xyz, const int32_t *conn,
                        const double *sc, const uint8_t *fixed, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    double acc[6] = {0, 0, 0, 0, 0, 0}, f6[6];
    for (int64_t q = ptr[i]; q < ptr[i + 1]; ++q) {
      strut_end_force(inc[q] >> 1, (int)(inc[q] & 1), xyz, conn, sc, x, f6);
      for (int k = 0; k < 6; ++k) acc[k] += f6[k];
    }
    for (int k = 0; k < 6; ++k) y[6 * i + k] = (fixed && fixed[6 * i + k]) ? 0.0 : acc[k];
  }
}

int oracle_pcg_mt(int64_t N, int64_t B, const double *xyz, const int32_t *conn, const double *sc,
                  const uint8_t *fixed, const double *ubar, const double *f, double rtol, int maxit, double *u,
                  double *relres_out) {
  const int64_t n = 6 * N;
  /* malloc + parallel first touch: with calloc the master thread would place every page on its own NUMA node and the
   * 128 threads of the GPU box's host would all pull from one memory controller (round 2: 1.5 x over one thread) */
  double *buf = (double *)malloc((size_t)n * 7 * sizeof(double));
  int64_t *ptr = (int64_t *)calloc((size_t)N + 1, sizeof(int64_t));
  int64_t *inc = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)B);
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
  if (!buf || !ptr || !inc || !fill) return -1000000;
  double *x = buf, *r = buf + n, *z = buf + 2 * n, *p = buf + 3 * n, *Ap = buf + 4 * n, *dinv = buf + 5 * n,
         *ub = buf + 6 * n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) x[i] = r[i] = z[i] = p[i] = Ap[i] = dinv[i] = ub[i] = 0.0;
  for (int64_t b = 0; b < B; ++b) { ptr[conn[2 * b] + 1]++; ptr[conn[2 * b + 1] + 1]++; }
  for (int64_t i = 0; i < N; ++i) { ptr[i + 1] += ptr[i]; fill[i] = ptr[i]; }
  for (int64_t b = 0; b < B; ++b) { inc[fill[conn[2 * b]]++] = b << 1; inc[fill[conn[2 * b + 1]]++] = (b << 1) | 1; }
  /* Jacobi diagonal by rows (same numbers as oracle_diag, summed per node instead of scattered per strut) */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    double dg[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t q = ptr[i]; q < ptr[i + 1]; ++q) {
      const int64_t b = inc[q] >> 1;
      const int tip = (int)(inc[q] & 1);
      const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
      const double ka = sc[5 * b], kt = sc[5 * b + 1], a = sc[5 * b + 2], bb2 = sc[5 * b + 3], c = sc[5 * b + 4];
      const double d[3] = {xyz[3 * ib] - xyz[3 * ia], xyz[3 * ib + 1] - xyz[3 * ia + 1], xyz[3 * ib + 2] - xyz[3 * ia + 2]};
      const double L2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], L = sqrt(L2);
      const double cA = c + a * L2 - 2.0 * bb2 * L;
      for (int k = 0; k < 3; ++k) {
        const double t2 = d[k] * d[k] / L2;
        dg[k] += ka * t2 + a * (1.0 - t2);
        dg[3 + k] += kt * t2 + (tip ? c : cA) * (1.0 - t2);
      }
    }
    for (int k = 0; k < 6; ++k) dinv[6 * i + k] = dg[k];
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    dinv[i] = (fixed[i] || dinv[i] == 0.0) ? 0.0 : 1.0 / dinv[i];
    ub[i] = fixed[i] ? ubar[i] : 0.0;
  }
  spmv_gather(N, ptr, inc, xyz, conn, sc, NULL, ub, Ap);
  double bb = 0.0, rz = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : bb, rz)
  for (int64_t i = 0; i < n; ++i) {
    r[i] = fixed[i] ? 0.0 : f[i] - Ap[i];
    z[i] = dinv[i] * r[i];
    p[i] = z[i];
    bb += r[i] * r[i];
    rz += r[i] * z[i];
  }
  int it = 0, conv = 0;
  double rr = bb;
  if (bb > 0.0) {
    for (it = 1; it <= maxit; ++it) {
      spmv_gather(N, ptr, inc, xyz, conn, sc, fixed, p, Ap);
      double pAp = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : pAp)
      for (int64_t i = 0; i < n; ++i) pAp += p[i] * Ap[i];
      const double alpha = rz / pAp;
      double rz_new = 0.0;
      rr = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : rz_new, rr)
      for (int64_t i = 0; i < n; ++i) {
        x[i] += alpha * p[i];
        r[i] -= alpha * Ap[i];
        z[i] = dinv[i] * r[i];
        rz_new += r[i] * z[i];
        rr += r[i] * r[i];
      }
      if (rr <= rtol * rtol * bb) { conv = 1; break; }
      const double beta = rz_new / rz;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
      rz = rz_new;
    }
  } else {
    conv = 1;
  }
  for (int64_t i = 0; i < n; ++i) u[i] = fixed[i] ? ubar[i] : x[i];
  if (relres_out) *relres_out = bb > 0.0 ? sqrt(rr / bb) : 0.0;
  free(buf); free(ptr); free(inc); free(fill);
  return conv ? it : -it;
}
