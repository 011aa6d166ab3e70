// The whole PCG loop of a small lattice as ONE persistent launch (pl_opts_t.short_iteration = 2; experimental, round 5).
//
// One 512-thread workgroup per K*p tile, all co-resident (n_tiles <= the CU count), single-reduction CG (Chronopoulos - Gear,
// pl_cg1.h) WITHOUT node elimination - the form with the fewest exchanges per iteration:
//     u = M^-1 r (local: Jacobi + tile level + the tile's own rows of A_c^-1 r_c; r_c is kept by recurrence in every workgroup)
//     publish u                                                    -> hand-off 1: the rows of u other tiles' crossing visits read
//     w = K u (own rows; LDS-resident visits as k_spmv_tile_lds_t), delta_t = u.w, gamma_t = r.u, rr_t = r.r, Z^T w of the tile
//     publish [Z^T w_t | delta_t | gamma_t | rr_t]                 -> hand-off 2: all-gather of 15 doubles per tile
//     beta, alpha;  Z^T s = Z^T w + beta Z^T s;  r_c -= alpha Z^T s;  p = u + beta p;  s = w + beta s;  x += alpha p;  r -= alpha s
// x, r, p, s, the Jacobi weights, the lever arms and the tile's rows of A_c^-1 live in REGISTERS for the whole solve (thread i
// owns node i of the tile), u and K u in LDS, r_c and Z^T s (ncp doubles each) in LDS, replicated per workgroup.
// Hand-offs (MI355X_MICROARCH.md, inter-workgroup visibility): payload by agent-scope relaxed stores (global_store sc1,
// write-through), s_waitcnt vmcnt(0) in every storing wave, workgroup barrier, ONE flag store per workgroup; consumers poll the
// flags with agent-scope relaxed loads (one wave, s_sleep between polls, bounded: a flag that never comes sets the error
// word and every workgroup leaves), then read the payload with agent-scope loads.  Every polled word is zeroed by the host
// before the launch; epochs count iterations inside the launch.
// Sums that decide alpha / beta / the stopping test are formed in a FIXED order, so every workgroup takes the same decisions.
#pragma once
#include "pl_small.h"
#include "pl_tile.h"

namespace pl {

constexpr int kPersistBlock = 512;
constexpr int kPersistRed = 16;          // doubles a tile publishes per iteration: 12 restriction sums, delta, gamma, rr, spare
constexpr int kPersistMaxCols = 4;       // columns of A_c^-1 per thread: ncp <= 4 x 512

struct PersistArgs {
  // K*p plan (pl_tile.h)
  const TileDesc *tdesc;
  const uint32_t *vword;
  const int32_t *vother;
  const double2 *tab;
  int n_tab;
  const Rec5 *rec5;                // null: palette form
  const int32_t *foreign_idx;
  const uint8_t *fixedbits;
  int stride;
  // levels
  const int32_t *agg_of_tile;
  const double *cen, *xyz;
  const float *dinv32;
  const double *Bt_inv;            // [n_tiles][TM x TM] or null
  const float *Ainv;               // [ncp][ncp]
  const int32_t *agg_tile_ptr, *agg_tile_idx;   // aggregate -> its tiles (CSR), n_agg + 1 / n_tiles
  int ncp, cm, n_agg;
  // vectors (in: x0, r0; out: x, r)
  double *x, *r;
  // exchange
  double *Ug;                      // [N][6] published rows of u
  double *red;                     // [n_tiles][kPersistRed]
  unsigned *flagU, *flagR;         // [n_tiles] epochs
  unsigned *err;                   // != 0: a hand-off timed out
  // control / results
  double *hist;                    // ||r_k||^2 per iteration
  int max_iter;
  double thresh;
  int *iters_out;                  // [0] iterations run, [1] converged
  unsigned long long *dbg;         // [8] ticks of the 100 MHz clock workgroup 0 spent per phase (summed over the iterations)
};

#define PL_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// every flag of `flags[0 .. n)` >= epoch?  One wave polls (bounded); returns false on time-out or when another workgroup gave up.
__device__ __forceinline__ bool persist_wait_all(const unsigned *flags, int n, unsigned epoch, unsigned *err) {
  const int lane = threadIdx.x & 63;
  for (unsigned spins = 0;; ++spins) {
    bool ok = true;
    for (int q = lane; q < n; q += 64) ok = ok && (__hip_atomic_load(flags + q, PL_RLX) >= epoch);
    if (__all(ok)) return true;
    if (spins > (1u << 20) || __hip_atomic_load(err, PL_RLX) != 0u) {
      if (lane == 0) __hip_atomic_store(err, 1u, PL_RLX);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

template <int REC, int TM, int NC>
__global__ __launch_bounds__(kPersistBlock) void k_persist_cg1(PersistArgs a) {
  constexpr bool kStream = REC == kRecCompact;
  constexpr int kSrcChunks = kStream ? 2 : 4, kTabChunks = kStream ? 2 : kPalLdsChunks;
  extern __shared__ double lds[];
  const int stride = a.stride, ncp = a.ncp, cm = a.cm;
  double *ys = lds;                                                     // [6][stride] K u accumulator
  double2 *us2 = reinterpret_cast<double2 *>(ys + 6 * stride);           // [stride][3] rows of u
  double2 *ps2 = us2 + 3 * stride;                                       // record / direction palette
  double *rcs = reinterpret_cast<double *>(ps2 + kTabChunks * a.n_tab);  // [ncp] r_c
  double *scs = rcs + ncp;                                               // [ncp] Z^T s
  double *stage = scs + ncp;                                             // [gridDim.x][kPersistRed] gathered partial sums
  int32_t *agp = reinterpret_cast<int32_t *>(stage + (size_t)gridDim.x * kPersistRed);   // [n_agg + 1] aggregate -> tiles (CSR)
  int32_t *agi = agp + a.n_agg + 1;                                                       // [gridDim.x]
  __shared__ double redw[24][kPersistBlock / kWave];
  __shared__ double tot[24];
  __shared__ double yct[16];
  __shared__ int ok_s;
  const int t = blockIdx.x, G = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int nw = kPersistBlock / kWave;
  const TileDesc td = a.tdesc[t];
  const int n0 = td.n0, nn = td.n1 - td.n0;
  const bool own = tid < nn;                       // this thread owns node n0 + tid
  const int64_t node = n0 + tid;
  const int ag = a.agg_of_tile[t];
  // ---- static per-node data
  double rel[3] = {0, 0, 0};
  float dv[6] = {0, 0, 0, 0, 0, 0};
  unsigned fb = 0x3f;
  double xr[6], rr_[6], pr[6], sr[6], ur[6], wr[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) xr[k] = rr_[k] = pr[k] = sr[k] = ur[k] = wr[k] = 0.0;
  if (own) {
#pragma unroll
    for (int k = 0; k < 3; ++k) rel[k] = a.xyz[3 * node + k] - a.cen[3 * ag + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      dv[k] = a.dinv32[6 * node + k];
      xr[k] = a.x[6 * node + k];
      rr_[k] = a.r[6 * node + k];
    }
    fb = a.fixedbits[node];
  }
  // rows cm ag .. of A_c^-1: thread tid holds columns tid + 512 q
  float ar[12][NC];
#pragma unroll
  for (int m = 0; m < 12; ++m)
#pragma unroll
    for (int q = 0; q < NC; ++q) {
      const int j = tid + kPersistBlock * q;
      ar[m][q] = (m < cm && j < ncp) ? a.Ainv[(size_t)(cm * ag + m) * ncp + j] : 0.f;
    }
  // tile level: B_t^-1 of this tile, parked in LDS (only sixteen lanes ever use it)
  __shared__ double sbi[TM][TM];
  if (tid < TM * TM) sbi[tid / TM][tid % TM] = a.Bt_inv ? a.Bt_inv[(size_t)t * (TM * TM) + tid] : 0.0;
  // visit words / records of this thread (static for the whole solve)
  unsigned wvw[kLdsPre];
#pragma unroll
  for (int j = 0; j < kLdsPre; ++j) {
    const int k = tid + j * kPersistBlock;
    wvw[j] = k < td.n_int ? a.vword[td.v0 + k] : kNoVisit;
  }
  Rec5 q5[kLdsPre];
  if constexpr (kStream) {
#pragma unroll
    for (int j = 0; j < kLdsPre; ++j)
      if (wvw[j] != kNoVisit) q5[j] = a.rec5[td.h0 + tid + (int64_t)j * kPersistBlock];
  }
  for (int i = tid; i < kSrcChunks * a.n_tab; i += kPersistBlock)
    ps2[(i / kSrcChunks) * kTabChunks + (i % kSrcChunks)] = a.tab[i];
  for (int e = tid; e < ncp; e += kPersistBlock) scs[e] = 0.0;
  for (int e = tid; e <= a.n_agg; e += kPersistBlock) agp[e] = a.agg_tile_ptr[e];
  for (int e = tid; e < (int)gridDim.x; e += kPersistBlock) agi[e] = a.agg_tile_idx[e];
  // ---- r_c of the initial residual: restriction summed over ALL tiles through one all-gather (epoch 1 of flagR)
  auto restrict12 = [&](const double v[6], double acc[12]) {
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2];
    acc[3] += v[3] + (rel[1] * v[2] - rel[2] * v[1]);
    acc[4] += v[4] + (rel[2] * v[0] - rel[0] * v[2]);
    acc[5] += v[5] + (rel[0] * v[1] - rel[1] * v[0]);
    acc[6] += rel[0] * v[0]; acc[7] += rel[1] * v[1]; acc[8] += rel[2] * v[2];
    acc[9] += 0.5 * (rel[1] * v[0] + rel[0] * v[1]);
    acc[10] += 0.5 * (rel[2] * v[1] + rel[1] * v[2]);
    acc[11] += 0.5 * (rel[2] * v[0] + rel[0] * v[2]);
  };
  // block sums of NQ values per thread -> tot[0 .. NQ)   (NQ a compile-time constant: the values stay in registers)
  auto block_sums = [&](const double *vals, auto nq_tag) {
    constexpr int NQ = decltype(nq_tag)::value;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const double s = wave_sum(vals[q]);
      if (lane == 0) redw[q][wv] = s;
    }
    __syncthreads();
    if (tid < NQ) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < nw; ++w) s += redw[tid][w];
      tot[tid] = s;
    }
    __syncthreads();
  };
  // publish this tile's kPersistRed doubles (tot[0 .. 16)) and set its flag; gather everybody's; returns false on time-out
  auto all_gather = [&](unsigned epoch) -> bool {
    if (tid < kPersistRed) __hip_atomic_store(a.red + (size_t)t * kPersistRed + tid, tot[tid], PL_RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flagR + t, epoch, PL_RLX);
    if (wv == 0) {
      const bool good = persist_wait_all(a.flagR, G, epoch, a.err);
      if (lane == 0) ok_s = good ? 1 : 0;
    }
    __syncthreads();
    if (!ok_s) return false;
    for (int e = tid; e < G * kPersistRed; e += kPersistBlock) stage[e] = __hip_atomic_load(a.red + e, PL_RLX);
    __syncthreads();
    return true;
  };
  {
    double acc[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) acc[m] = 0.0;
    if (own) restrict12(rr_, acc);
    block_sums(acc, std::integral_constant<int, 12>());
    if (tid >= 12 && tid < kPersistRed) tot[tid] = 0.0;
    __syncthreads();
    if (!all_gather(1u)) return;
    for (int e = tid; e < ncp; e += kPersistBlock) {
      const int g = e / cm, m = e - cm * g;
      double s = 0.0;
      if (g < a.n_agg)
        for (int q = agp[g]; q < agp[g + 1]; ++q) s += stage[agi[q] * kPersistRed + m];
      rcs[e] = s;
    }
    __syncthreads();
  }
  double gamma_old = 0.0, alpha_old = 0.0;
  int k = 0, converged = 0;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
#define PL_PH(i)                                \
  do {                                          \
    const unsigned long long now_ = wall_clock64(); \
    ph[i] += now_ - tlast;                      \
    tlast = now_;                               \
  } while (0)
  for (; k < a.max_iter; ++k) {
    const unsigned eU = (unsigned)k + 1u, eR = (unsigned)k + 2u;
    // ---- u = D^-1 r + P Z (y_c + y_t)
    double part[24];
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < NC; ++q) {
        const int j = tid + kPersistBlock * q;
        if (j < ncp) s += (double)ar[m][q] * rcs[j];
      }
      part[m] = s;
      part[12 + m] = 0.0;
    }
    if (own) restrict12(rr_, part + 12);
    block_sums(part, std::integral_constant<int, 24>());
    if (tid < 16) {                          // tile solve: y_t = B_t^-1 (Z_t^T r); C = y_c + y_t in yct[0 .. 12)
      double y = 0.0;
      if (tid < TM) {
#pragma unroll
        for (int j = 0; j < TM; ++j) y += sbi[tid][j] * tot[12 + j];
      }
      yct[tid] = (tid < cm ? tot[tid] : 0.0) + (tid < TM && a.Bt_inv ? y : 0.0);
    }
    __syncthreads();
    double gam = 0.0;
    if (own) {
      double C[12];
#pragma unroll
      for (int m = 0; m < 12; ++m) C[m] = yct[m];
      double zc[6] = {C[0] + (C[4] * rel[2] - C[5] * rel[1]), C[1] + (C[5] * rel[0] - C[3] * rel[2]),
                      C[2] + (C[3] * rel[1] - C[4] * rel[0]), C[3], C[4], C[5]};
      zc[0] += C[6] * rel[0] + 0.5 * (C[9] * rel[1] + C[11] * rel[2]);
      zc[1] += C[7] * rel[1] + 0.5 * (C[9] * rel[0] + C[10] * rel[2]);
      zc[2] += C[8] * rel[2] + 0.5 * (C[10] * rel[1] + C[11] * rel[0]);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        ur[q] = (double)dv[q] * rr_[q] + (((fb >> q) & 1u) ? 0.0 : zc[q]);
        gam += rr_[q] * ur[q];
        __hip_atomic_store(a.Ug + 6 * node + q, ur[q], PL_RLX);
      }
      us2[3 * tid] = double2{ur[0], ur[1]};
      us2[3 * tid + 1] = double2{ur[2], ur[3]};
      us2[3 * tid + 2] = double2{ur[4], ur[5]};
    }
    for (int i = tid; i < 6 * stride; i += kPersistBlock) ys[i] = 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(a.flagU + t, eU, PL_RLX);
    PL_PH(0);
    if (wv == 0) {
      const bool good = persist_wait_all(a.flagU, G, eU, a.err);
      if (lane == 0) ok_s = good ? 1 : 0;
    }
    __syncthreads();
    PL_PH(1);
    if (!ok_s) break;
    // ---- w = K u on the tile's rows
    auto record_of = [&](unsigned id, const Rec5 &q) -> Record {
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
    auto interior = [&](unsigned w, const Rec5 &q) {
      const int la = (int)(w & ((1u << kVisRowBits) - 1)), lb = (int)((w >> kVisRowBits) & ((1u << kVisRowBits) - 1));
      const Record r = record_of((w >> kVisPidShift) & 0xFFu, q);
      const double2 *pa = us2 + 3 * la, *pb = us2 + 3 * lb;
      const double2 a0 = pa[0], a1 = pa[1], a2 = pa[2], b0 = pb[0], b1 = pb[1], b2 = pb[2];
      const V3 uA = {a0.x, a0.y, a1.x}, tA = {a1.y, a2.x, a2.y}, uB = {b0.x, b0.y, b1.x}, tB = {b1.y, b2.x, b2.y};
      V3 F, M;
      tip_force(r, uA, tA, uB, tB, F, M);
      lds_add6(ys + lb, stride, F, M);
      const V3 d = {r.dx, r.dy, r.dz};
      lds_add6(ys + la, stride, (-1.0) * F, (-1.0) * M - cross(d, F));
    };
#pragma unroll
    for (int j = 0; j < kLdsPre; ++j)
      if (wvw[j] != kNoVisit) interior(wvw[j], q5[j]);
    for (int kk = tid + kLdsPre * kPersistBlock; kk < td.n_int; kk += kPersistBlock) {      // large tiles
      Rec5 q = Rec5();
      if constexpr (kStream) q = a.rec5[td.h0 + kk];
      interior(a.vword[td.v0 + kk], q);
    }
    for (int kc = tid; kc < td.n_cross; kc += kPersistBlock) {
      const unsigned cw = a.vword[td.v0 + td.n_int + kc];
      const int32_t co = a.vother[td.c0 + kc];
      Rec5 q = Rec5();
      if constexpr (kStream) q = a.rec5[kc < td.n_ch ? td.h0 + td.n_int + kc : (int64_t)a.foreign_idx[td.f0 + (kc - td.n_ch)]];
      double o[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) o[c] = __hip_atomic_load(a.Ug + 6 * (int64_t)co + c, PL_RLX);
      const V3 uO = {o[0], o[1], o[2]}, tO = {o[3], o[4], o[5]};
      const int lo = (int)(cw & ((1u << kVisRowBits) - 1));
      const bool ownB = (cw >> kVisRowBits) & 1u;
      const Record r = record_of((cw >> kVisPidShift) & 0xFFu, q);
      const double2 *po = us2 + 3 * lo;
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
    __syncthreads();
    PL_PH(2);
    // ---- partial sums of this tile: Z^T w, delta, gamma, rr
    double pr15[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) pr15[m] = 0.0;
    if (own) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        wr[q] = ((fb >> q) & 1u) ? 0.0 : ys[q * stride + tid];
        pr15[12] += ur[q] * wr[q];
        pr15[14] += rr_[q] * rr_[q];
      }
      pr15[13] = gam;
      restrict12(wr, pr15);
    }
    block_sums(pr15, std::integral_constant<int, 15>());
    if (tid == 15) tot[15] = 0.0;
    __syncthreads();
    PL_PH(3);
    if (!all_gather(eR)) break;
    PL_PH(4);
    // ---- scalars in a fixed order (every workgroup alike), coarse recurrences
    if (wv < 3) {                          // wave w sums column 12 + w over the tiles: lanes take g = lane, lane + 64, ... in
      double s = 0.0;                      // order, then a DPP tree - the same order in every workgroup
      for (int g = lane; g < G; g += 64) s += stage[g * kPersistRed + 12 + wv];
      s = wave_sum(s);
      if (lane == 0) tot[16 + wv] = s;
    }
    __syncthreads();
    const double delta = tot[16], gamma = tot[17], rr_now = tot[18];
    if (t == 0 && tid == 0) a.hist[k] = rr_now;
    if (!(rr_now > a.thresh)) {          // (also leaves on NaN)
      converged = rr_now <= a.thresh ? 1 : 0;
      break;
    }
    double beta = 0.0, den = delta;
    if (k > 0 && gamma_old != 0.0 && alpha_old != 0.0) {
      beta = gamma / gamma_old;
      den = delta - beta * gamma / alpha_old;
    }
    const double alpha = den != 0.0 ? gamma / den : 0.0;
    gamma_old = gamma;
    alpha_old = alpha;
    for (int e = tid; e < ncp; e += kPersistBlock) {
      const int g = e / cm, m = e - cm * g;
      double zw = 0.0;
      if (g < a.n_agg)
        for (int q = agp[g]; q < agp[g + 1]; ++q) zw += stage[agi[q] * kPersistRed + m];
      const double sv = zw + beta * scs[e];
      scs[e] = sv;
      rcs[e] -= alpha * sv;
    }
    if (own) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        pr[q] = ur[q] + beta * pr[q];
        sr[q] = wr[q] + beta * sr[q];
        xr[q] += alpha * pr[q];
        rr_[q] -= alpha * sr[q];
      }
    }
    __syncthreads();
    PL_PH(5);
  }
#undef PL_PH
  if (t == 0 && tid == 0 && a.dbg)
    for (int i = 0; i < 8; ++i) a.dbg[i] = ph[i];
  if (own) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      a.x[6 * node + q] = xr[q];
      a.r[6 * node + q] = rr_[q];
    }
  }
  if (t == 0 && tid == 0) {
    a.iters_out[0] = k;
    a.iters_out[1] = converged;
  }
}

#undef PL_RLX

}  // namespace pl
