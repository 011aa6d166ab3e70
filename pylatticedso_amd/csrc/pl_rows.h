// K*p by ROWS, LDS-resident (round 4 experiment, opt-in with PL_ROWS=1): one lane per node, no scatter.
//
// The tile kernels of pl_tile.h visit every strut once and scatter both end forces into an LDS accumulator with
// ds_add_f64 - twelve atomics per visit; SQ_WAIT_INST_LDS is 30 % of that kernel's wave cycles.  By rows, every strut is
// evaluated TWICE - once for each end, by the lane that owns that end's row - and nothing is scattered:
//
//   * a tile's own rows of x AND the rows of its out-of-tile neighbours (the "halo": one list per tile, gathered once)
//     are staged in LDS; a half-visit is one 32-bit word = LDS row of the other end | orientation | palette id;
//   * the palette holds every record in both orientations (pl_device.h::reversed), so the owning node is always the tip
//     end B of the strut it looks at: one code path, F and M accumulate in registers by fused multiply-adds;
//   * the lanes of a wave are 64 nodes of (mostly) one kind walking their struts in the same order, so slot by slot a wave
//     looks at ONE palette entry: it comes through the scalar cache into scalar registers (s_load_dwordx16; the fused
//     multiply-adds take it from there), lanes with another entry follow in a waterfall loop;
//   * no atomics, no accumulator, one barrier; the result is bitwise reproducible (fixed summation order);
//   * node elimination (pl_coarse.h) becomes a row selection: each pass evaluates only the rows of its kind.
//
// MEASURED (50^3 Octet, profiles/r04_e_rows_sq.json against r04_e_tile_sq.json): 51 us against 36 us of k_spmv_tile_lds.
// The LDS stalls are gone (SQ_WAIT_INST_LDS 0.1 M against 36.7 M wave cycles, LDS instructions 0.74 M against 1.35 M, bank
// conflicts 1.4 M against 3.9 M), but the vector ALU does more, not less: the element's ~55 fp64 operations are issued
// twice per strut, and SQ_THREAD_CYCLES_VALU comes out EQUAL (516 M against 511 M lane-instructions: 85 per half-visit
// against 127 per visit at 1.33 visits per strut) - the unpacking the scatter form pays is what the second evaluation
// costs.  What then decides is lane utilisation: 52 % here (tiles of 108 / 144 / 192 / 256 nodes on 64-lane waves, 13 % of
// the wave-slots walked twice by the waterfall, lattice faces) against 82 %, i.e. 15.4 M issued vector instructions against
// 9.8 M.  configs[2] (both passes of the condensed operator by rows): 300 against 283 us per iteration.  It would take
// tiles cut to whole waves of one node kind to turn this around; the bricks are the preconditioner's blocks and are cut
// for it.  Kept as an opt-in and as the bitwise-reproducible form; tests/test_gpu_parity.py holds it to the oracle.
//
// Only the palette form (a strut's record is a palette entry): graded lattices keep the streaming tile kernel, which reads
// every 40-byte record once - by rows it would be read twice, and that kernel is bound by those bytes.
#pragma once
#include "pl_tile.h"

namespace pl {

struct RowDesc {
  int32_t n0, nn;        // first node, own rows
  int32_t nh, S;         // halo rows, slots (largest number of struts at one node of the tile)
  int32_t pitch, pad;    // words per slot
  int64_t w0, h0;        // first word (slot s of node i: w0 + s * pitch + i), first halo entry
};

struct RowPlan {
  bool ready = false;
  int64_t n_tiles = 0, n_words = 0, n_halo = 0;
  int max_rows = 0, max_S = 0;
  TBuf<RowDesc> rdesc;
  TBuf<uint32_t> rloc;      // static bits of the words: LDS row of the other end (bits 0..9), orientation (bit 10)
  TBuf<int32_t> rstrut;     // strut of a word, -1 = padding
  TBuf<int32_t> halo;       // node id of every halo row
};

constexpr int kRowBits = 10;             // LDS rows per tile (own + halo) < 1024
constexpr int kRowOrientBit = 10;        // 1: the owning node is end A of the strut -> reversed record
constexpr int kRowPidShift = 21;         // dense palette id (8 bits), as in the visit words of pl_tile.h
#ifndef PL_ROW_BLOCK
#define PL_ROW_BLOCK 256
#endif
#ifndef PL_ROW_WAVES
#define PL_ROW_WAVES 6
#endif
#ifndef PL_ROW_CHUNK
#define PL_ROW_CHUNK 8
#endif
constexpr int kRowBlock = PL_ROW_BLOCK;       // one lane per node: a tile's <= 256 nodes in one pass
constexpr int kRowChunk = PL_ROW_CHUNK;       // words per lane in flight (one chunk evaluated, the next requested)

// conn in the device numbering (struts by home tile - only the node numbering matters here).
inline int build_row_plan(RowPlan &plan, const std::vector<int32_t> &conn, int64_t N, int64_t B,
                          const std::vector<int32_t> &tile_start, const double *xyz) {
  plan.ready = false;
  const int64_t T = (int64_t)tile_start.size() - 1;
  if (!xyz || T <= 0) return 0;
  // node -> half-visits (strut, end)
  std::vector<int64_t> ptr((size_t)N + 1, 0);
  for (int64_t k = 0; k < 2 * B; ++k) ptr[(size_t)conn[k] + 1]++;
  for (int64_t i = 0; i < N; ++i) ptr[i + 1] += ptr[i];
  std::vector<int32_t> inc((size_t)2 * B);                   // 2 * strut + end
  {
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t k = 0; k < 2 * B; ++k) inc[fill[conn[k]]++] = (int32_t)k;
  }
  std::vector<RowDesc> rd((size_t)T);
  std::vector<std::vector<int32_t>> halos((size_t)T);
  bool too_big = false;
  parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
    for (int64_t t = t0; t < t1; ++t) {
      const int32_t n0 = tile_start[t], n1 = tile_start[t + 1];
      int S = 0;
      std::vector<int32_t> &h = halos[t];
      for (int32_t i = n0; i < n1; ++i) {
        S = std::max<int>(S, (int)(ptr[i + 1] - ptr[i]));
        for (int64_t q = ptr[i]; q < ptr[i + 1]; ++q) {
          const int32_t o = conn[inc[q] ^ 1];
          if (o < n0 || o >= n1) h.push_back(o);
        }
      }
      std::sort(h.begin(), h.end());
      h.erase(std::unique(h.begin(), h.end()), h.end());
      const int nn = n1 - n0;
      rd[t] = {n0, nn, (int32_t)h.size(), S, (nn + 15) & ~15, 0, 0, 0};
      if (nn + (int)h.size() >= (1 << kRowBits)) too_big = true;
    }
  }, 16);
  if (too_big) return 0;                                      // (a tile with 1024+ rows: the tile kernels stay)
  int64_t w = 0, hh = 0;
  int max_rows = 0, max_S = 0;
  for (int64_t t = 0; t < T; ++t) {
    rd[t].w0 = w;
    rd[t].h0 = hh;
    w += (int64_t)rd[t].S * rd[t].pitch;
    hh += rd[t].nh;
    max_rows = std::max(max_rows, rd[t].nn + rd[t].nh);
    max_S = std::max(max_S, rd[t].S);
  }
  std::vector<uint32_t> rloc((size_t)std::max<int64_t>(1, w), kNoVisit);
  std::vector<int32_t> rstrut((size_t)std::max<int64_t>(1, w), -1), halo((size_t)std::max<int64_t>(1, hh));
  parallel_for(T, [&](int64_t t0, int64_t t1, unsigned) {
    std::vector<std::pair<uint64_t, int32_t>> run;
    for (int64_t t = t0; t < t1; ++t) {
      const RowDesc &d = rd[t];
      const std::vector<int32_t> &h = halos[t];
      std::copy(h.begin(), h.end(), halo.begin() + d.h0);
      for (int32_t i = 0; i < d.nn; ++i) {
        const int32_t node = d.n0 + i;
        run.clear();
        for (int64_t q = ptr[node]; q < ptr[node + 1]; ++q) {
          const int32_t o = conn[inc[q] ^ 1];
          // slots in the order of the signed direction own -> other: nodes of one kind (the tile's nodes are sorted by
          // kind) then have the same strut in the same slot, and a wave reads one or two palette entries per slot
          run.push_back({strut_dir_code(xyz, node, o), inc[q]});
        }
        std::sort(run.begin(), run.end());
        for (size_t s = 0; s < run.size(); ++s) {
          const int32_t k = run[s].second, o = conn[k ^ 1];
          uint32_t row;
          if (o >= d.n0 && o < d.n0 + d.nn) row = (uint32_t)(o - d.n0);
          else row = (uint32_t)(d.nn + (std::lower_bound(h.begin(), h.end(), o) - h.begin()));
          const uint32_t own_is_A = (k & 1) ? 0u : 1u;
          const size_t at = (size_t)(d.w0 + (int64_t)s * d.pitch + i);
          rloc[at] = row | (own_is_A << kRowOrientBit);
          rstrut[at] = k >> 1;
        }
      }
    }
  }, 16);
  if (std::getenv("PL_ROWS_STATS")) {
    // how many distinct (direction, orientation) classes a wave of 64 nodes meets per slot, halo sizes, tile sizes
    double waves = 0, iters = 0, live = 0, lanes = 0;
    int64_t hsum = 0, nsum = 0, smax = 0;
    for (int64_t t = 0; t < T; ++t) {
      const RowDesc &d = rd[t];
      hsum += d.nh; nsum += d.nn; smax = std::max<int64_t>(smax, d.S);
      for (int w0 = 0; w0 < d.nn; w0 += 64)
        for (int sl = 0; sl < d.S; ++sl) {
          std::vector<uint64_t> keys;
          int nl = 0;
          for (int i = w0; i < std::min(d.nn, w0 + 64); ++i) {
            const size_t at = (size_t)(d.w0 + (int64_t)sl * d.pitch + i);
            if (rstrut[at] < 0) continue;
            ++nl;
            const int32_t b = rstrut[at];
            const bool ownA = (rloc[at] >> kRowOrientBit) & 1u;
            keys.push_back(strut_dir_code(xyz, conn[2 * b + (ownA ? 0 : 1)], conn[2 * b + (ownA ? 1 : 0)]));
          }
          std::sort(keys.begin(), keys.end());
          keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
          waves += 1; iters += (double)keys.size(); live += nl; lanes += 64;
        }
    }
    std::fprintf(stderr, "[rows] tiles %lld nodes/tile %.1f halo/tile %.1f max S %lld | wave-slots %.0f, distinct classes per "
                 "wave-slot %.2f, live lanes %.1f %%, max rows %d\n", (long long)T, (double)nsum / T, (double)hsum / T, (long long)smax,
                 waves, iters / waves, 100.0 * live / lanes, max_rows);
  }
  if (plan.rdesc.upload(rd) != hipSuccess || plan.rloc.upload(rloc) != hipSuccess ||
      plan.rstrut.upload(rstrut) != hipSuccess || plan.halo.upload(halo) != hipSuccess)
    return 3;
  plan.n_tiles = T;
  plan.n_words = w;
  plan.n_halo = hh;
  plan.max_rows = max_rows;
  plan.max_S = max_S;
  plan.ready = true;
  return 0;
}

// Final words: static bits | dense palette id of the strut.  Behind every palette build (ids change with the radii).
__global__ __launch_bounds__(kBlock) void k_row_words(int64_t n_words, const uint32_t *__restrict__ rloc,
                                                      const int32_t *__restrict__ rstrut,
                                                      const uint16_t *__restrict__ pal,
                                                      const int *__restrict__ dense_of_slot,
                                                      uint32_t *__restrict__ rword) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (v >= n_words) return;
  const int32_t b = rstrut[v];
  unsigned w = kNoVisit;
  if (b >= 0) w = rloc[v] | (((unsigned)dense_of_slot[pal[b]] & 0xFFu) << kRowPidShift);
  rword[v] = w;
}

// The dense palette in both orientations: entry 2 p = record p, entry 2 p + 1 = the same strut seen from its other end.
__global__ void k_pal_orient(int n_pal_max, const Record *__restrict__ pal_dense, Record *__restrict__ pal2) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pal_max) return;
  const Record r = pal_dense[p];
  pal2[2 * p] = r;
  pal2[2 * p + 1] = reversed(r);
}

template <bool MASK, bool DOT, typename VT, int ENDS>
__global__ __launch_bounds__(kRowBlock) __attribute__((amdgpu_waves_per_eu(PL_ROW_WAVES, 8)))
void k_spmv_rows(const RowDesc *__restrict__ rdesc, const uint32_t *__restrict__ rword, const int32_t *__restrict__ halo,
                 const Record *__restrict__ pal2, const uint8_t *__restrict__ fixedbits,
                 const VT *__restrict__ x, VT *__restrict__ y, double *__restrict__ dot_out,
                 const uint8_t *__restrict__ cflag, CondSolve cs, const int32_t *__restrict__ tile_list) {
  constexpr bool kToCondensed = ENDS == kEndsCondensed || ENDS == kEndsCondensedSolve;
  extern __shared__ double2 xs2[];                                  // [nn + nh][3]: own rows, then halo rows
  PL_STAMP(0);
  unsigned t = xcd_block(blockIdx.x, gridDim.x);
  if (tile_list) t = (unsigned)tile_list[t];
  const RowDesc rd = rdesc[t];
  const int n0 = rd.n0, nn = rd.nn, S = rd.S;
  // this lane's node (first pass over the tile) and the first chunk of its words: requested with the rows, before the barrier
  const int i0 = (int)threadIdx.x;
  unsigned cur[kRowChunk];
#pragma unroll
  for (int j = 0; j < kRowChunk; ++j) cur[j] = (i0 < nn && j < S) ? rword[rd.w0 + (int64_t)j * rd.pitch + i0] : kNoVisit;
  for (int i = threadIdx.x; i < 3 * rd.nh; i += kRowBlock) {        // halo rows: id, then the row (two hops)
    const int h = i / 3;
    const int32_t node = halo[rd.h0 + h];
    double2 val = {0.0, 0.0};
    if (!(ENDS == kEndsCondensedSolve && cflag[node])) val = load_pair(x, 3 * (int64_t)node + (i - 3 * h));
    xs2[3 * nn + i] = val;
  }
  for (int i = threadIdx.x; i < 3 * nn; i += kRowBlock) {           // own rows: 16 B per lane, contiguous
    double2 val = {0.0, 0.0};
    // (fused first pass of the condensed operator: the rows being rewritten count as zero and are not read)
    if (!(ENDS == kEndsCondensedSolve && cflag[n0 + i / 3])) val = load_pair(x, 3 * (int64_t)n0 + i);
    xs2[i] = val;
  }
  PL_STAMP(1);
  __syncthreads();
  PL_STAMP(2);
  // a wave without a node has nothing left to do (a tile of 150 nodes fills three of the four waves; in the passes of the
  // condensed operator the eliminated / kept nodes are whole waves at the end / start of a tile)
  if ((int)(threadIdx.x & ~63u) >= nn) return;
  if (ENDS != kEndsAll && nn <= kRowBlock) {
    // a pass over one kind of rows: kept and eliminated nodes are whole runs of a tile, so a wave is usually all of one kind
    const bool mine = i0 < nn && ((cflag[n0 + i0] != 0) == kToCondensed);
    if (__builtin_amdgcn_ballot_w64(mine) == 0) return;
  }
  double acc = 0.0;
  for (int base = 0; base < nn; base += kRowBlock) {                // (one pass: a tile has <= 256 nodes by default)
    const int i = base + i0;
    bool live = i < nn;
    if (ENDS != kEndsAll && live) live = ((cflag[n0 + i] != 0) == kToCondensed);
    V3 uW = {0, 0, 0}, tW = {0, 0, 0}, F = {0, 0, 0}, M = {0, 0, 0};
    if (live) {
      const double2 *po = xs2 + 3 * i;
      const double2 a0 = po[0], a1 = po[1], a2 = po[2];
      uW = {a0.x, a0.y, a1.x};
      tW = {a1.y, a2.x, a2.y};
    }
    // One half-visit: the owning node is the tip end B of palette entry `key` (both orientations are in the table), the
    // other end A sits in LDS row `row`.  The lanes of a wave are 64 nodes of (mostly) one kind and walk their struts in
    // the same order, so slot by slot the whole wave looks at ONE entry: it is read through the scalar cache into scalar
    // registers (the fused multiply-adds take it from there - no LDS read, no vector register for the record).  A wave
    // that holds several entries (kind boundaries, lattice faces: 13 % of the wave-slots at 50^3 Octet) takes them one
    // after the other ("waterfall").
    auto half_visit = [&](unsigned w) {
      unsigned todo = (w != kNoVisit) ? (((w >> kRowPidShift) & 0xFFu) * 2u + ((w >> kRowOrientBit) & 1u)) : 0xFFFFFFFFu;
      while (todo != 0xFFFFFFFFu) {
        const unsigned k0 = __builtin_amdgcn_readfirstlane(todo);
        // (the comparison below goes through an opaque copy: from `todo == k0` the optimiser would conclude that inside
        // the branch the VECTOR register `todo` can stand for k0, and turn the record's scalar loads into per-lane ones)
        unsigned kc = k0;
        asm volatile("" : "+v"(kc));
        typedef const __attribute__((address_space(4))) double *cptr;
        cptr q = (cptr)(reinterpret_cast<const double *>(pal2) + 8 * (size_t)k0);
        const double ra = q[0], rc = q[1], e1 = q[2], e2 = q[3], e3 = q[4];
        const V3 d = {q[5], q[6], q[7]};
        if (todo == kc) {
          const double2 *pa = xs2 + 3 * (int)(w & ((1u << kRowBits) - 1));
          const double2 b0 = pa[0], b1 = pa[1], b2 = pa[2];
          const V3 uA = {b0.x, b0.y, b1.x}, tA = {b1.y, b2.x, b2.y};
          const V3 du = uW - uA + cross(d, tA);
          const V3 dth = tW - tA;
          const V3 cf = cross(d, dth), cm = cross(d, du);
          const double sf = e1 * dot(du, d), sm = e3 * dot(dth, d);
          F.x = fma(ra, du.x, F.x); F.y = fma(ra, du.y, F.y); F.z = fma(ra, du.z, F.z);              // a du
          F.x = fma(sf, d.x, F.x); F.y = fma(sf, d.y, F.y); F.z = fma(sf, d.z, F.z);                 // e1 (du.d) d
          F.x = fma(e2, cf.x, F.x); F.y = fma(e2, cf.y, F.y); F.z = fma(e2, cf.z, F.z);              // e2 (d x dth)
          M.x = fma(rc, dth.x, M.x); M.y = fma(rc, dth.y, M.y); M.z = fma(rc, dth.z, M.z);           // c dth
          M.x = fma(sm, d.x, M.x); M.y = fma(sm, d.y, M.y); M.z = fma(sm, d.z, M.z);                 // e3 (dth.d) d
          M.x = fma(-e2, cm.x, M.x); M.y = fma(-e2, cm.y, M.y); M.z = fma(-e2, cm.z, M.z);           // - e2 (d x du)
          todo = 0xFFFFFFFFu;
        }
      }
    };
    for (int c0 = 0; c0 < S; c0 += kRowChunk) {
      unsigned nxt[kRowChunk];
      if (c0 + kRowChunk < S || base + kRowBlock < nn) {
        // the next chunk of this node - or the first chunk of the node of the next pass - while this one is evaluated
        const bool more = c0 + kRowChunk < S;
        const int ib = more ? i : i + kRowBlock, sb = more ? c0 + kRowChunk : 0;
#pragma unroll
        for (int j = 0; j < kRowChunk; ++j)
          nxt[j] = (ib < nn && sb + j < S) ? rword[rd.w0 + (int64_t)(sb + j) * rd.pitch + ib] : kNoVisit;
      }
#pragma unroll
      for (int j = 0; j < kRowChunk; ++j)
        if (c0 + j < S) half_visit(live ? cur[j] : kNoVisit);
#pragma unroll
      for (int j = 0; j < kRowChunk; ++j) cur[j] = nxt[j];
    }
    PL_STAMP(3);
    if (!live) continue;
    const int64_t node = (int64_t)n0 + i;
    if (ENDS == kEndsCondensedSolve) {
      // v = -K_cc^-1 (accumulated row): the equilibrium position of the eliminated node under the others' x
      const int32_t b0 = cs.base[node];
      if (b0 >= 0) {
        const double v6[6] = {F.x, F.y, F.z, M.x, M.y, M.z};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double *A = cs.inv + b0 + 6 * k;
          double vv = 0.0;
#pragma unroll
          for (int j = 0; j < 6; ++j) vv += A[j] * v6[j];
          y[6 * node + k] = (VT)(-vv);
        }
      }
      continue;
    }
    double2 v0 = {F.x, F.y}, v1 = {F.z, M.x}, v2 = {M.y, M.z};
    if (MASK) {
      const unsigned fb = fixedbits[node];
      if (fb & 1u) v0.x = 0.0;
      if (fb & 2u) v0.y = 0.0;
      if (fb & 4u) v1.x = 0.0;
      if (fb & 8u) v1.y = 0.0;
      if (fb & 16u) v2.x = 0.0;
      if (fb & 32u) v2.y = 0.0;
    }
    store_pair(y, 3 * node, v0);
    store_pair(y, 3 * node + 1, v1);
    store_pair(y, 3 * node + 2, v2);
    if (DOT) acc += uW.x * v0.x + uW.y * v0.y + uW.z * v1.x + tW.x * v1.y + tW.y * v2.x + tW.z * v2.y;
  }
  if (DOT) {       // one atomic per wave (no second barrier: the waves of a tile finish on their own)
    const double s = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(dot_out + ((blockIdx.x * (kRowBlock / kWave) + (threadIdx.x >> 6)) & (kSlots - 1)), s);
  }
  PL_STAMP(4);
  PL_STAMP(5);
}

// false: this launch does not fit the row kernel (the caller falls back to the tile kernels).
template <typename VT>
inline bool launch_spmv_rows(const RowPlan &plan, const uint32_t *rword, const Record *pal2, int n_pal,
                             const uint8_t *fixedbits, const VT *x, VT *y, double *dot_dev, hipStream_t s,
                             int ends = kEndsAll, const uint8_t *cflag = nullptr, CondSolve cs = CondSolve(),
                             const int32_t *tile_list = nullptr, int64_t n_list = 0) {
  if (!plan.ready || !rword || !pal2 || n_pal <= 0 || n_pal > kPalDenseMax) return false;
  const size_t lds = (size_t)plan.max_rows * 48;
  if (lds > 64 * 1024 - 256) return false;
  if (tile_list && n_list <= 0) return true;
  const dim3 g((unsigned)(tile_list ? n_list : plan.n_tiles)), blk(kRowBlock);
#define PL_R(M, D, E)                                                                                               \
  hipLaunchKernelGGL((k_spmv_rows<M, D, VT, E>), g, blk, lds, s, plan.rdesc.p, rword, plan.halo.p, pal2, fixedbits, \
                     x, y, dot_dev, cflag, cs, tile_list)
#define PL_RR(E)                                          \
  do {                                                    \
    if (fixedbits && dot_dev) PL_R(true, true, E);        \
    else if (fixedbits) PL_R(true, false, E);             \
    else if (dot_dev) PL_R(false, true, E);               \
    else PL_R(false, false, E);                           \
  } while (0)
  if (ends == kEndsCondensed) PL_RR(kEndsCondensed);
  else if (ends == kEndsCondensedSolve) PL_R(false, false, kEndsCondensedSolve);
  else if (ends == kEndsOthers) PL_RR(kEndsOthers);
  else PL_RR(kEndsAll);
#undef PL_RR
#undef PL_R
  return true;
}

}  // namespace pl
