// HIP kernels of libpylattice_hip (gfx950 / MI355X only; wave = 64).
#pragma once
#include "pl_device.h"

namespace pl {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kDefaultLPN = 4;          // lanes per node in the gather kernels (64 / LPN nodes per wave / ELL slice)
constexpr int kSlots = 64;              // every device-side reduction scalar is spread over 64 atomics targets:
                                        // 2000+ blocks adding into ONE address serialise at the memory side

// blockIdx -> logical block so that each of the 8 XCDs (blocks are dealt round-robin, b and b+8 share an XCD and
// its private 4 MiB L2) walks one CONTIGUOUS eighth of the node/strut range: struts and their end nodes are then
// served from one L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nblk) {
  const unsigned xcd = b & 7u, q = nblk >> 3, r = nblk & 7u;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}

// Sum over the 64 lanes of a wave, returned in EVERY lane; all lanes must be active.  Data-parallel-primitive moves on
// the vector ALU (row_shr 1, 2, 4, 8: an inclusive scan inside each row of 16 lanes, zeros shifted in), then the four
// row totals through scalar registers.  The shuffle form (__shfl_down: two ds_bpermute_b32 per step and double) goes
// through the LDS crossbar of the CU - with ten workgroups per CU each reducing eight values at the end of
// k_pcg_update_tile that pipe, not HBM, set the kernel's time (25.5 us, 18.2 us with the reductions cut out).
__device__ __forceinline__ double dpp_row_shr_add(double v, const int ctrl_tag) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (ctrl_tag) {   // the control word must be an immediate
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true); break;
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xf, 0xf, true); break;
    case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xf, 0xf, true); break;
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xf, 0xf, true); break;
  }
  return v + __hiloint2double(hi, lo);
}
// The first half of wave_sum alone: afterwards lanes 15, 31, 47, 63 hold the totals of their rows of 16 lanes (callers
// that go through LDS anyway store those four instead of paying the v_readlane round)
__device__ __forceinline__ double row_sums(double v) {
  v = dpp_row_shr_add(v, 1);
  v = dpp_row_shr_add(v, 2);
  v = dpp_row_shr_add(v, 4);
  return dpp_row_shr_add(v, 8);
}
// value of lane l (a compile-time constant after unrolling) of the calling wave, for every lane: two v_readlane_b32
__device__ __forceinline__ double lane_value(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// ... of a lane that differs from lane to lane (ds_bpermute)
__device__ __forceinline__ double lane_value_dyn(double v, int l) {
  return __hiloint2double(__shfl(__double2hiint(v), l), __shfl(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_row_shr_add(v, 1);
  v = dpp_row_shr_add(v, 2);
  v = dpp_row_shr_add(v, 4);
  v = dpp_row_shr_add(v, 8);          // lane 15 of every row now holds the row's total
  const int lo = __double2loint(v), hi = __double2hiint(v);
  double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
  t += __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
  t += __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
  t += __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
  return t;
}

// Reduction scalars live as kSlots partial sums; slot chosen by block so concurrent blocks rarely collide.
__device__ __forceinline__ void scalar_add(double *scal, int which, double v) {
  unsafeAtomicAdd(scal + which * kSlots + (blockIdx.x & (kSlots - 1)), v);
}
// Every lane of the calling wave gets the total.
__device__ __forceinline__ double scalar_read(const double *scal, int which) {
  const int lane = threadIdx.x & 63;
  double v = 0.0;
#pragma unroll
  for (int s = lane; s < kSlots; s += 64) v += scal[which * kSlots + s];
  return wave_sum(v);
}

// Two / three scalars at once: their slot values are requested together and reduced afterwards - one memory round trip,
// where consecutive scalar_read calls wait for each other's load.
__device__ __forceinline__ void scalar_read2(const double *scal, int a, int b, double &va, double &vb) {
  static_assert(kSlots == kWave, "one slot per lane");
  const int lane = threadIdx.x & 63;
  const double xa = scal[a * kSlots + lane], xb = scal[b * kSlots + lane];
  va = wave_sum(xa);
  vb = wave_sum(xb);
}
__device__ __forceinline__ void scalar_read3(const double *scal, int a, int b, int c, double &va, double &vb, double &vc) {
  static_assert(kSlots == kWave, "one slot per lane");
  const int lane = threadIdx.x & 63;
  const double xa = scal[a * kSlots + lane], xb = scal[b * kSlots + lane], xc = scal[c * kSlots + lane];
  va = wave_sum(xa);
  vb = wave_sum(xb);
  vc = wave_sum(xc);
}

// Sum over the block, result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *smem /*[kBlock/kWave]*/) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < kBlock / kWave; ++i) t += smem[i];
  }
  return t;
}

__device__ __forceinline__ void load6(const double *__restrict__ p, V3 &u, V3 &t) {
  const double2 *q = reinterpret_cast<const double2 *>(p);
  const double2 a = q[0], b = q[1], c = q[2];
  u = {a.x, a.y, b.x};
  t = {b.y, c.x, c.y};
}

// fp32-stored vectors (opts.precision = 1 / 2): rows are 24 B, read as three float2; all arithmetic stays fp64.
__device__ __forceinline__ void load6(const float *__restrict__ p, V3 &u, V3 &t) {
  const float2 *q = reinterpret_cast<const float2 *>(p);
  const float2 a = q[0], b = q[1], c = q[2];
  u = {(double)a.x, (double)a.y, (double)b.x};
  t = {(double)b.y, (double)c.x, (double)c.y};
}
// third `part` (0..2) of node row `node` of a node-major [N][6] vector, as two doubles
__device__ __forceinline__ double2 load_pair(const double *v, int64_t pair) {
  return reinterpret_cast<const double2 *>(v)[pair];
}
__device__ __forceinline__ double2 load_pair(const float *v, int64_t pair) {
  const float2 f = reinterpret_cast<const float2 *>(v)[pair];
  return {(double)f.x, (double)f.y};
}
__device__ __forceinline__ void store_pair(double *v, int64_t pair, double2 a) {
  reinterpret_cast<double2 *>(v)[pair] = a;
}
__device__ __forceinline__ void store_pair(float *v, int64_t pair, double2 a) {
  reinterpret_cast<float2 *>(v)[pair] = {(float)a.x, (float)a.y};
}

__device__ __forceinline__ Record load_record(const Record *__restrict__ rec, int64_t i) {
  const double2 *q = reinterpret_cast<const double2 *>(rec + i);
  const double2 a = q[0], b = q[1], c = q[2], d = q[3];
  Record r;
  r.a = a.x; r.c = a.y; r.e1 = b.x; r.e2 = b.y; r.e3 = c.x; r.dx = c.y; r.dy = d.x; r.dz = d.y;
  return r;
}

// ---------------------------------------------------------------------------------------------------------
// Record build: one thread per strut ("local stiffness build").
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_build_records(int64_t B, const double *__restrict__ xyz,
                                                          const int32_t *__restrict__ conn,
                                                          const double *__restrict__ radius,
                                                          const double *__restrict__ seg_len,
                                                          const int32_t *__restrict__ seg_nsub,
                                                          const double *__restrict__ mult, Material m,
                                                          Record *__restrict__ rec, double *__restrict__ rec5) {
  const int64_t b = (int64_t)xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  if (mult) m = scaled(m, mult[b]);   // k identical chains in parallel (pl_set_multiplicity)
  const int ia = conn[2 * b], ib = conn[2 * b + 1];
  const V3 d = {xyz[3 * (int64_t)ib] - xyz[3 * (int64_t)ia], xyz[3 * (int64_t)ib + 1] - xyz[3 * (int64_t)ia + 1],
                xyz[3 * (int64_t)ib + 2] - xyz[3 * (int64_t)ia + 2]};
  const double len[3] = {seg_len[3 * b], seg_len[3 * b + 1], seg_len[3 * b + 2]};
  const int ns[3] = {seg_nsub[3 * b], seg_nsub[3 * b + 1], seg_nsub[3 * b + 2]};
  const Flex f = strut_flexibility(radius[b], len, ns, m);
  const Record r = make_record(scalars_from_flex(f), d);
  rec[b] = r;
  if (rec5) {   // compact copy for the streaming K*p (pl_tile.h, kRecCompact)
    double *q = rec5 + 5 * b;
    q[0] = r.a; q[1] = r.c; q[2] = r.e1; q[3] = r.e2; q[4] = r.e3;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K*x, variant 1: one thread per strut, scatter with f64 global atomics (y must be zeroed).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_spmv_atomic(int64_t B, const int32_t *__restrict__ conn,
                                                        const Record *__restrict__ rec, const double *__restrict__ x,
                                                        double *__restrict__ y) {
  const int64_t b = (int64_t)xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
  if (b >= B) return;
  const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
  const Record r = load_record(rec, b);
  V3 uA, tA, uB, tB, F, M;
  load6(x + 6 * ia, uA, tA);
  load6(x + 6 * ib, uB, tB);
  tip_force(r, uA, tA, uB, tB, F, M);
  const V3 d = {r.dx, r.dy, r.dz};
  const V3 MA = (-1.0) * M - cross(d, F);
  double *ya = y + 6 * ia, *yb = y + 6 * ib;
  unsafeAtomicAdd(yb + 0, F.x); unsafeAtomicAdd(yb + 1, F.y); unsafeAtomicAdd(yb + 2, F.z);
  unsafeAtomicAdd(yb + 3, M.x); unsafeAtomicAdd(yb + 4, M.y); unsafeAtomicAdd(yb + 5, M.z);
  unsafeAtomicAdd(ya + 0, -F.x); unsafeAtomicAdd(ya + 1, -F.y); unsafeAtomicAdd(ya + 2, -F.z);
  unsafeAtomicAdd(ya + 3, MA.x); unsafeAtomicAdd(ya + 4, MA.y); unsafeAtomicAdd(ya + 5, MA.z);
}

// y = mask .* y (atomic variant post-pass), optionally accumulating dot(x, y) into dot_out[kSlots].
__global__ __launch_bounds__(kBlock) void k_mask_dot(int64_t n6, const uint8_t *__restrict__ fixed,
                                                     const double *__restrict__ x, double *__restrict__ y,
                                                     double *__restrict__ dot_out) {
  __shared__ double red[kBlock / kWave];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * kBlock) {
    double v = y[i];
    if (fixed && fixed[i]) { v = 0.0; y[i] = 0.0; }
    acc += x[i] * v;
  }
  if (dot_out) {
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), t);
  }
}

// ---------------------------------------------------------------------------------------------------------
// K*x, variant 2: per-node gather over the incident struts.  kLPN = 4 lanes share one node (16 nodes per wave), so a
// 12-valent Octet node is three loop trips per lane and the launch has 4x the waves of a lane-per-node mapping.
// Sliced ELL, one slice = 16 nodes = one wave:  ent[slice_ptr[s] + j*16 + n] = j-th strut of node 16 s + n as
// (other node, strut | end<<31), other < 0 = padding; slice width is a multiple of 4 so that lane = (j%4)*16 + n
// reads 64 consecutive entries per trip.  No atomics, bitwise reproducible.
// fixedbits[node] holds the 6 Dirichlet flags; MASK=true gives y = P K x (x is assumed to be 0 on fixed dofs).
// ---------------------------------------------------------------------------------------------------------
template <int LPN>
__device__ __forceinline__ double lpn_sum(double v) {
#pragma unroll
  for (int o = kWave / LPN; o < kWave; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPN, bool MASK, bool DOT>
__global__ __launch_bounds__(kBlock) void k_spmv_gather(int64_t N, const int64_t *__restrict__ slice_ptr,
                                                        const int2 *__restrict__ ent, const Record *__restrict__ rec,
                                                        const uint8_t *__restrict__ fixedbits,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        double *__restrict__ dot_out) {
  __shared__ double red[kBlock / kWave];
  const unsigned blk = xcd_block(blockIdx.x, gridDim.x);
  constexpr int kSliceNodes = kWave / LPN;
  const int lane = threadIdx.x & 63, sub = lane / kSliceNodes;
  const int64_t slice = (int64_t)blk * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t i = slice * kSliceNodes + (lane & (kSliceNodes - 1));
  double acc = 0.0;
  if (slice * kSliceNodes < N) {          // wave-uniform
    const bool live = i < N;
    V3 us = {0, 0, 0}, ts = {0, 0, 0};
    if (live) load6(x + 6 * i, us, ts);
    V3 F = {0, 0, 0}, M = {0, 0, 0};
    const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
#pragma unroll 2
    for (int64_t p = p0 + lane; p < p1; p += 64) {
      const int2 e = ent[p];
      if (e.x >= 0) {
        Record r = load_record(rec, e.y & 0x7fffffff);
        V3 uo, to, f, m;
        load6(x + 6 * (int64_t)e.x, uo, to);
        if (e.y < 0) r = reversed(r);   // this node is the strut's point1
        tip_force(r, uo, to, us, ts, f, m);
        F = F + f;
        M = M + m;
      }
    }
    double out[6] = {lpn_sum<LPN>(F.x), lpn_sum<LPN>(F.y), lpn_sum<LPN>(F.z),
                     lpn_sum<LPN>(M.x), lpn_sum<LPN>(M.y), lpn_sum<LPN>(M.z)};
    if (live) {
      if (MASK) {
        const unsigned fb = fixedbits[i];
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (fb & (1u << k)) out[k] = 0.0;
      }
      double2 *q = reinterpret_cast<double2 *>(y + 6 * i);
      if (LPN >= 4) {   // lanes sub = 0,1,2 of a node store one 16-byte third of its row each
        if (sub == 0) q[0] = {out[0], out[1]};
        else if (sub == 1) q[1] = {out[2], out[3]};
        else if (sub == 2) q[2] = {out[4], out[5]};
      } else if (sub == 0) {
        q[0] = {out[0], out[1]};
        q[1] = {out[2], out[3]};
        q[2] = {out[4], out[5]};
      }
      if (DOT && sub == 0)
        acc = us.x * out[0] + us.y * out[1] + us.z * out[2] + ts.x * out[3] + ts.y * out[4] + ts.z * out[5];
    }
  }
  if (DOT) {
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), t);
  }
}

// Jacobi diagonal (and its inverse on free dofs) by the same gather.  dinv = 1/diag on free dofs, 0 on fixed.
template <int LPN>
__global__ __launch_bounds__(kBlock) void k_diag_gather(int64_t N, const int64_t *__restrict__ slice_ptr,
                                                        const int2 *__restrict__ ent, const Record *__restrict__ rec,
                                                        const uint8_t *__restrict__ fixedbits,
                                                        double *__restrict__ diag, double *__restrict__ dinv) {
  constexpr int kSliceNodes = kWave / LPN;
  const int lane = threadIdx.x & 63, sub = lane / kSliceNodes;
  const int64_t slice = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  const int64_t i = slice * kSliceNodes + (lane & (kSliceNodes - 1));
  if (slice * kSliceNodes >= N) return;
  double dg[6] = {0, 0, 0, 0, 0, 0};
  const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
  for (int64_t p = p0 + lane; p < p1; p += 64) {
    const int2 e = ent[p];
    if (e.x >= 0) {
      Record r = load_record(rec, e.y & 0x7fffffff);
      if (e.y < 0) r = reversed(r);
      double t[6];
      tip_diag(r, t);
#pragma unroll
      for (int k = 0; k < 6; ++k) dg[k] += t[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) dg[k] = lpn_sum<LPN>(dg[k]);
  if (i < N && sub == 0) {
    const unsigned fb = fixedbits ? fixedbits[i] : 0u;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      diag[6 * i + k] = dg[k];
      if (dinv) dinv[6 * i + k] = (((fb >> k) & 1u) || dg[k] == 0.0) ? 0.0 : 1.0 / dg[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// BSR(6x6) numeric fill, same lane mapping: every lane writes the off-diagonal blocks of its entries, the diagonal
// block is the lane-group sum of the Kss blocks.  Row i holds [diag block] + one block per incident strut, columns
// ascending (slots fixed at setup).  with_bc: dolfinx Dirichlet treatment (rows/cols zeroed, unit diagonal).
// ---------------------------------------------------------------------------------------------------------
// Stores: a lane's 6 x 6 block is 288 contiguous bytes, but the 64 lanes of a wave own 64 blocks scattered over the rows
// of 16 nodes - written lane by lane that is 36 store instructions of 64 x 8 bytes with a 288-byte stride (2.4 TB/s).
// Every block is therefore staged in LDS (one row of kBsrPitch doubles per lane) and the wave writes three whole blocks
// per instruction: lanes 0-17 / 18-35 / 36-53 each store the 18 double2 of one block.
constexpr int kBsrPitch = 38;          // doubles per staged block (36 + 2: rows stay 16-byte aligned)
constexpr int kBsrBlock = 128;         // two waves per workgroup: 2 x 19 KB of staging, four workgroups per CU
__device__ __forceinline__ void bsr_store_staged(double *stage /* this wave's [64][kBsrPitch] */, int lane,
                                                 int64_t my_block /* -1: nothing staged by this lane */,
                                                 double *__restrict__ vals) {
  const int g = lane / 18, l = lane - 18 * g;        // g = 3: idle lanes 54..63
#pragma unroll 2
  for (int j = 0; j < 66; j += 3) {
    const int src = j + g;
    const int64_t blk = __shfl(my_block, src < 64 ? src : 0, 64);
    if (g < 3 && src < 64 && blk >= 0) {
      const double2 v = *reinterpret_cast<const double2 *>(stage + src * kBsrPitch + 2 * l);
      *reinterpret_cast<double2 *>(vals + 36 * blk + 2 * l) = v;
    }
  }
}

template <int LPN>
__global__ __launch_bounds__(kBsrBlock) void k_bsr_fill(int64_t N, const int64_t *__restrict__ slice_ptr,
                                                     const int2 *__restrict__ ent, const Record *__restrict__ rec,
                                                     const int64_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ ent_slot,
                                                     const int32_t *__restrict__ diag_slot,
                                                     const uint8_t *__restrict__ fixedbits, int with_bc,
                                                     double *__restrict__ vals, int64_t slice0 = 0,
                                                     int64_t slice1 = (int64_t)1 << 62) {
  extern __shared__ double bsr_lds[];                   // [kBsrBlock / kWave][64][kBsrPitch]
  constexpr int kSliceNodes = kWave / LPN;
  const int lane = threadIdx.x & 63, sub = lane / kSliceNodes, wv = threadIdx.x >> 6;
  double *stage = bsr_lds + (size_t)wv * 64 * kBsrPitch;
  const int64_t slice = slice0 + (int64_t)blockIdx.x * (kBsrBlock / kWave) + wv;     // (a launch covers slices [slice0, slice1))
  const int64_t i = slice * kSliceNodes + (lane & (kSliceNodes - 1));
  if (slice >= slice1 || slice * kSliceNodes >= N) return;                 // wave-uniform
  const bool live = i < N;
  double Kd[36];
#pragma unroll
  for (int k = 0; k < 36; ++k) Kd[k] = 0.0;
  const unsigned fi = (live && with_bc && fixedbits) ? fixedbits[i] : 0u;
  const int64_t p0 = slice_ptr[slice], p1 = slice_ptr[slice + 1];
  const int64_t row0 = live ? rowptr[i] : 0;
  // A trip is a chain entry -> record -> blocks -> stores, and a wave makes 3-4 of them with 8 waves per CU (LDS): the NEXT
  // trip's entry and slot are requested right after this trip's record (vector loads return in order: requested before
  // it they would stand between the record and the arithmetic), so only the first trip pays both hops.
  int2 e_next = {-1, 0};
  int32_t slot_next = 0;
  if (p0 + lane < p1) {
    e_next = ent[p0 + lane];
    slot_next = ent_slot[p0 + lane];
  }
  for (int64_t pb = p0; pb < p1; pb += 64) {            // whole wave takes every trip (the staged stores are collective)
    const int64_t p = pb + lane;
    int64_t my_block = -1;
    const int2 e = e_next;
    const int32_t slot = slot_next;
    Record r;
    unsigned fo = 0u;
    const bool have = p < p1 && e.x >= 0;
    if (have) {
      r = load_record(rec, e.y & 0x7fffffff);
      if (with_bc && fixedbits) fo = fixedbits[e.x];
    }
    e_next = int2{-1, 0};
    if (p + 64 < p1) {
      e_next = ent[p + 64];
      slot_next = ent_slot[p + 64];
    }
    if (p < p1) {
      if (e.x >= 0) {
        if (e.y < 0) r = reversed(r);
        double Kss[36], Kso[36];
        tip_blocks(r, Kss, Kso);
        my_block = row0 + slot;
        double *dst = stage + lane * kBsrPitch;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            Kd[a * 6 + b] += Kss[a * 6 + b];
            const bool z = ((fi >> a) & 1u) || ((fo >> b) & 1u);
            dst[a * 6 + b] = z ? 0.0 : Kso[a * 6 + b];
          }
      }
    }
    bsr_store_staged(stage, lane, my_block, vals);       // (LDS accesses of one wave are ordered: no barrier needed)
  }
#pragma unroll
  for (int k = 0; k < 36; ++k) Kd[k] = lpn_sum<LPN>(Kd[k]);
  int64_t my_block = -1;
  if (live && sub == 0) {
    my_block = row0 + diag_slot[i];
    double *dd = stage + lane * kBsrPitch;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const bool fa = (fi >> a) & 1u, fb = (fi >> b) & 1u;
        dd[a * 6 + b] = (fa || fb) ? ((a == b) ? 1.0 : 0.0) : Kd[a * 6 + b];
      }
  }
  bsr_store_staged(stage, lane, my_block, vals);
}

// y = A x for the assembled BSR matrix: one thread per block row (cross-check path).
__global__ __launch_bounds__(kBlock) void k_bsr_spmv(int64_t N, const int64_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ colidx,
                                                     const double *__restrict__ vals, const double *__restrict__ x,
                                                     double *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
    const double *xv = x + 6 * (int64_t)colidx[p];
    const double *A = vals + 36 * p;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[a] += A[a * 6 + b] * xv[b];
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) y[6 * i + a] = acc[a];
}

// ---------------------------------------------------------------------------------------------------------
// PCG vector kernels.  Reduction scalars live on the device as kSlots partial sums each: scal[which*kSlots + slot].
// ---------------------------------------------------------------------------------------------------------
enum { S_RZ_OLD = 0, S_PAP = 1, S_RZ_NEW = 2, S_RR = 3, S_BB = 4, S_AUX = 5, S_PP = 5, S_XX = 6, S_ALPHA = 7, S_COUNT = 8 };

// x += alpha p ; r -= alpha Ap ; z = dinv r ; rz_new += r.z ; rr += r.r      (alpha = rz_old / pAp)
// alpha_max > 0 clamps the step like the reference's conjugate_gradient_solver.py:79 (used by the DDM solve).
// REF: the extra bookkeeping of the reference's hand-written CG (conjugate_gradient_solver.py:96-109): ||x||^2 and the
// "direction norm" ||pn||^2 (pn = the search direction, or the vector a restart replaces it with; null: the restart
// source is the updated residual itself, its norm is r.r) into the S_XX / S_PP slots, the step length into S_ALPHA.
template <bool REF>
__global__ __launch_bounds__(kBlock) void k_pcg_update(int64_t n6, const double *__restrict__ p,
                                                       const double *__restrict__ Ap,
                                                       const double *__restrict__ dinv, double *__restrict__ x,
                                                       double *__restrict__ r, double *__restrict__ z,
                                                       double *__restrict__ scal, double alpha_max,
                                                       const double *__restrict__ pn,
                                                       const int *__restrict__ stop = nullptr /* see k_pcg_direction */) {
  __shared__ double red[4][kBlock / kWave];
  if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // the iterate is final
  const double pap = scalar_read(scal, S_PAP);
  double alpha = (pap != 0.0) ? scalar_read(scal, S_RZ_OLD) / pap : 0.0;
  if (alpha_max > 0.0 && alpha > alpha_max) alpha = alpha_max;
  double rz = 0.0, rr = 0.0, xx = 0.0, pp = 0.0;
  if (REF && blockIdx.x == 0 && threadIdx.x == 0) scal[S_ALPHA * kSlots] = alpha;
  const int64_t n2 = n6 >> 1;   // n6 is even (6 per node)
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    const double2 pv = reinterpret_cast<const double2 *>(p)[i];
    const double2 av = reinterpret_cast<const double2 *>(Ap)[i];
    const double2 dv = reinterpret_cast<const double2 *>(dinv)[i];
    double2 xv = reinterpret_cast<double2 *>(x)[i];
    double2 rv = reinterpret_cast<double2 *>(r)[i];
    xv.x += alpha * pv.x; xv.y += alpha * pv.y;
    rv.x -= alpha * av.x; rv.y -= alpha * av.y;
    const double2 zv = {dv.x * rv.x, dv.y * rv.y};
    reinterpret_cast<double2 *>(x)[i] = xv;
    reinterpret_cast<double2 *>(r)[i] = rv;
    reinterpret_cast<double2 *>(z)[i] = zv;
    rz += rv.x * zv.x + rv.y * zv.y;
    rr += rv.x * rv.x + rv.y * rv.y;
    if (REF) {
      xx += xv.x * xv.x + xv.y * xv.y;
      if (pn) {
        const double2 q = reinterpret_cast<const double2 *>(pn)[i];
        pp += q.x * q.x + q.y * q.y;
      }
    }
  }
  rz = wave_sum(rz);
  rr = wave_sum(rr);
  if (REF) {
    xx = wave_sum(xx);
    pp = wave_sum(pp);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = rz; red[1][w] = rr; red[2][w] = xx; red[3][w] = pp; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0, c = 0, d = 0;
    for (int k = 0; k < kBlock / kWave; ++k) { a += red[0][k]; b += red[1][k]; c += red[2][k]; d += red[3][k]; }
    scalar_add(scal, S_RZ_NEW, a);
    scalar_add(scal, S_RR, b);
    if (REF) {
      scalar_add(scal, S_XX, c);
      scalar_add(scal, S_PP, pn ? d : b);
    }
  }
}

// p = z + beta p   (beta = rz_new / rz_old), and the end-of-iteration scalar bookkeeping: the reduction scalars are
// double-buffered by iteration parity, so block 0 can record ||r||^2 and prepare the NEXT iteration's set
// (rz_old <- rz_new, accumulators zeroed) while the other blocks still read the current one.
// psrc (may be null = p): the vector the new direction is built on, p = z + beta psrc - on a restart iteration of the
// reference CG that is the previous z (with a preconditioner) or the updated residual (without: z aliases r there).
// hist_cap > 0: also record ||psrc||^2, ||x||^2 and the step length (slots filled by k_pcg_update<true>) behind the
// residual history, at hist[hist_cap + k], hist[2 hist_cap + k], hist[3 hist_cap + k].
// stop != null (DDM handles, round 5): the stopping rules of conjugate_gradient_solver.py:96-109 are evaluated HERE, on the
// numbers the host would look at - ||r||^2 <= thresh, or (mintol > 0) ||p|| < mintol (||x|| + 1e-12) - and the first iteration
// that meets one sets *stop = k + 1; from then on this kernel and k_pcg_update return at once, so the iterate stays exactly
// where the reference's loop would have left it however many iterations the host had queued (it used to drain the stream
// after EVERY iteration for that: ~190 us per iteration against 25 - 50 us of work).
__global__ __launch_bounds__(kBlock) void k_pcg_direction(int64_t n6, const double *__restrict__ z,
                                                          double *p, const double *__restrict__ scal,
                                                          double *__restrict__ scal_next, double *__restrict__ hist,
                                                          int k, const double *psrc, int hist_cap,
                                                          int *__restrict__ stop = nullptr, double thresh = 0.0,
                                                          double mintol = 0.0) {
  if (stop && __hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
  const double old = scalar_read(scal, S_RZ_OLD);
  const double beta = (old != 0.0) ? scalar_read(scal, S_RZ_NEW) / old : 0.0;
  if (blockIdx.x == 0 && threadIdx.x < kWave) {
    const double rr = scalar_read(scal, S_RR);
    double pp = 0.0, xx = 0.0;
    if (hist_cap > 0) {
      pp = scalar_read(scal, S_PP);
      xx = scalar_read(scal, S_XX);
    }
    const int s = threadIdx.x;
    if (s == 0) {
      hist[k] = rr;
      if (hist_cap > 0) {
        hist[hist_cap + k] = pp;
        hist[2 * hist_cap + k] = xx;
        hist[3 * hist_cap + k] = scal[S_ALPHA * kSlots];
      }
      if (stop && (rr <= thresh || (hist_cap > 0 && mintol > 0.0 && sqrt(pp) < mintol * (sqrt(xx) + 1e-12))))
        __hip_atomic_store(stop, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int q = s; q < kSlots; q += kWave) {
      scal_next[S_RZ_OLD * kSlots + q] = scal[S_RZ_NEW * kSlots + q];
      scal_next[S_RZ_NEW * kSlots + q] = 0.0;
      scal_next[S_RR * kSlots + q] = 0.0;
      scal_next[S_PAP * kSlots + q] = 0.0;
      if (hist_cap > 0) {
        scal_next[S_PP * kSlots + q] = 0.0;
        scal_next[S_XX * kSlots + q] = 0.0;
      }
    }
  }
  if (!psrc) psrc = p;
  const int64_t n2 = n6 >> 1;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    const double2 zv = reinterpret_cast<const double2 *>(z)[i];
    double2 pv = reinterpret_cast<const double2 *>(psrc)[i];
    pv.x = zv.x + beta * pv.x;
    pv.y = zv.y + beta * pv.y;
    reinterpret_cast<double2 *>(p)[i] = pv;
  }
}

// r = mask.*(f - y) ; z = dinv r ; p = z ; x = 0 ; rz_old = r.z ; bb = r.r
__global__ __launch_bounds__(kBlock) void k_pcg_init(int64_t n6, const double *__restrict__ f,
                                                     const double *__restrict__ Kubar,
                                                     const uint8_t *__restrict__ fixed,
                                                     const double *__restrict__ dinv, double *__restrict__ x,
                                                     double *__restrict__ r, double *__restrict__ z,
                                                     double *__restrict__ p, double *__restrict__ scal) {
  __shared__ double red[2][kBlock / kWave];
  double rz = 0.0, rr = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * kBlock) {
    const double rv = fixed[i] ? 0.0 : (f[i] - Kubar[i]);
    const double zv = dinv[i] * rv;
    x[i] = 0.0;
    r[i] = rv;
    z[i] = zv;
    p[i] = zv;
    rz += rv * zv;
    rr += rv * rv;
  }
  rz = wave_sum(rz);
  rr = wave_sum(rr);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = rz; red[1][w] = rr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int k = 0; k < kBlock / kWave; ++k) { a += red[0][k]; b += red[1][k]; }
    scalar_add(scal, S_RZ_OLD, a);
    scalar_add(scal, S_BB, b);
  }
}

// u = fixed ? ubar : x
__global__ __launch_bounds__(kBlock) void k_compose_solution(int64_t n6, const uint8_t *__restrict__ fixed,
                                                             const double *__restrict__ ubar,
                                                             const double *__restrict__ x, double *__restrict__ u) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * kBlock)
    u[i] = fixed[i] ? ubar[i] : x[i];
}

// ---------------------------------------------------------------------------------------------------------
// Per-strut sensitivity s_b = lam_e^T (dK_e/dr) u_e and strain energy (one thread per strut, no scatter).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_sens(int64_t B, const double *__restrict__ xyz,
                                                 const int32_t *__restrict__ conn, const double *__restrict__ radius,
                                                 const double *__restrict__ seg_len,
                                                 const int32_t *__restrict__ seg_nsub,
                                                 const double *__restrict__ mult, Material m,
                                                 const double *__restrict__ u, const double *__restrict__ lam,
                                                 double *__restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  if (mult) m = scaled(m, mult[b]);
  const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
  const V3 d = {xyz[3 * ib] - xyz[3 * ia], xyz[3 * ib + 1] - xyz[3 * ia + 1], xyz[3 * ib + 2] - xyz[3 * ia + 2]};
  const double len[3] = {seg_len[3 * b], seg_len[3 * b + 1], seg_len[3 * b + 2]};
  const int ns[3] = {seg_nsub[3 * b], seg_nsub[3 * b + 1], seg_nsub[3 * b + 2]};
  const double r = radius[b];
  const Flex f = strut_flexibility(r, len, ns, m);
  const Record dr = make_record(dscalars_dr(f, r), d);
  V3 uA, tA, uB, tB, F, M;
  load6(u + 6 * ia, uA, tA);
  load6(u + 6 * ib, uB, tB);
  tip_force(dr, uA, tA, uB, tB, F, M);
  // energy-conjugate pairing: lam_e . (dK u)_e = F.(dlu) + M.(dlth) with the SAME relative deformations of lam
  V3 lA, mA, lB, mB;
  load6(lam + 6 * ia, lA, mA);
  load6(lam + 6 * ib, lB, mB);
  const V3 dlu = lB - lA + cross(d, mA);
  const V3 dlt = mB - mA;
  out[b] = dot(F, dlu) + dot(M, dlt);
}

// ---------------------------------------------------------------------------------------------------------
// Back-substitution of the penalisation points ("node_mod" points of LatticeSim.set_penalized_beams,
// lattice_sim.py:245-308): the strut carries no load between its ends, so the section force F is constant along it and
// the moment about a point Q is M_B + (x_B - x_Q) x F; the junctions q1 (end of the penalised segment at point1) and q2
// (start of the one at point2) follow by integrating the segment flexibilities from end A:
//     u_Q = u_P + th_P x (x_Q - x_P) + du,   th_Q = th_P + dth
//     du  = fa (F.t) t + f11 F_perp + f12 (M_Q x t),   dth = ft (M_Q.t) t + f12 (t x F) + f22 M_Q_perp
// with the closed-form chain flexibilities of pl_device.h.  out[b] = [u(q1) th(q1) u(q2) th(q2)] (12 doubles); an absent
// segment returns the end node's own values.  One thread per strut, no scatter.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void segment_step(double l, int nsub, double R, const Material &m, V3 t, V3 F, V3 MQ,
                                             V3 &u, V3 &th) {
  const double PI = 3.14159265358979323846;
  const double S = PI * R * R, I = 0.25 * PI * R * R * R * R;
  const double ES = m.E * S, GS = m.G * m.kappa * S, EI = m.E * I, GJ = m.G * 2.0 * I;
  const double n = (double)nsub;
  const double fa = l / ES, ft = l / GJ;
  const double f11 = l / GS + l * l * l / (3.0 * EI) * (1.0 - 1.0 / (4.0 * n * n));
  const double f12 = l * l / (2.0 * EI), f22 = l / EI;
  const double Ft = dot(F, t), Mt = dot(MQ, t);
  const V3 Fp = F - Ft * t, Mp = MQ - Mt * t;
  const V3 du = (fa * Ft) * t + f11 * Fp + f12 * cross(MQ, t);
  const V3 dth = (ft * Mt) * t + f12 * cross(t, F) + f22 * Mp;
  u = u + l * cross(th, t) + du;      // rigid part uses the rotation at the segment's start
  th = th + dth;
}

__global__ __launch_bounds__(kBlock) void k_node_mod(int64_t B, const double *__restrict__ xyz,
                                                     const int32_t *__restrict__ conn,
                                                     const double *__restrict__ radius,
                                                     const double *__restrict__ seg_len,
                                                     const int32_t *__restrict__ seg_nsub,
                                                     const double *__restrict__ mult, Material m,
                                                     const Record *__restrict__ rec, const double *__restrict__ u,
                                                     double *__restrict__ out) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  if (mult) m = scaled(m, mult[b]);   // the record's force is shared by the parallel chains: flexibility / k
  const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
  const Record r = load_record(rec, b);
  V3 uA, tA, uB, tB, F, M;
  load6(u + 6 * ia, uA, tA);
  load6(u + 6 * ib, uB, tB);
  tip_force(r, uA, tA, uB, tB, F, M);                      // load at end B that holds this deformation
  const V3 d = {r.dx, r.dy, r.dz};
  const double L = sqrt(dot(d, d));
  const V3 t = (1.0 / L) * d;
  const double l1 = seg_len[3 * b], l2 = seg_len[3 * b + 1], l3 = seg_len[3 * b + 2];
  (void)l3;
  const double rr = radius[b];
  V3 uq = uA, tq = tA;
  double s = 0.0;
  double *o = out + 12 * b;
  if (l1 > 0.0) {
    s = l1;
    const V3 MQ = M + (L - s) * cross(t, F);               // moment about q1
    segment_step(l1, seg_nsub[3 * b], m.pen * rr, m, t, F, MQ, uq, tq);
  }
  o[0] = uq.x; o[1] = uq.y; o[2] = uq.z; o[3] = tq.x; o[4] = tq.y; o[5] = tq.z;
  if (seg_len[3 * b + 2] > 0.0) {
    s += l2;
    const V3 MQ = M + (L - s) * cross(t, F);               // moment about q2
    if (l2 > 0.0) segment_step(l2, seg_nsub[3 * b + 1], rr, m, t, F, MQ, uq, tq);
    o[6] = uq.x; o[7] = uq.y; o[8] = uq.z; o[9] = tq.x; o[10] = tq.y; o[11] = tq.z;
  } else {
    o[6] = uB.x; o[7] = uB.y; o[8] = uB.z; o[9] = tB.x; o[10] = tB.y; o[11] = tB.z;
  }
}

// partial strain energies 1/2 e^T K e per strut, block-reduced and atomically added.
__global__ __launch_bounds__(kBlock) void k_energy(int64_t B, const int32_t *__restrict__ conn,
                                                   const Record *__restrict__ rec, const double *__restrict__ u,
                                                   double *__restrict__ out) {
  __shared__ double red[kBlock / kWave];
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  double e = 0.0;
  if (b < B) {
    const int64_t ia = conn[2 * b], ib = conn[2 * b + 1];
    const Record r = load_record(rec, b);
    V3 uA, tA, uB, tB, F, M;
    load6(u + 6 * ia, uA, tA);
    load6(u + 6 * ib, uB, tB);
    tip_force(r, uA, tA, uB, tB, F, M);
    const V3 d = {r.dx, r.dy, r.dz};
    e = 0.5 * (dot(F, uB - uA + cross(d, tA)) + dot(M, tB - tA));
  }
  const double t = block_sum(e, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out + (blockIdx.x & (kSlots - 1)), t);
}

// Periodic constraints (pl_set_periodic): v <- Q v, the average over every group of nodes that share their dofs, written
// back to all members.  One thread per (group, component).
__global__ __launch_bounds__(kBlock) void k_periodic_average(int64_t n_groups, const int32_t *__restrict__ gptr,
                                                             const int32_t *__restrict__ gnodes, double *__restrict__ v) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= 6 * n_groups) return;
  const int64_t g = t / 6;
  const int k = (int)(t - 6 * g);
  const int32_t b = gptr[g], e = gptr[g + 1];
  double s = 0.0;
  for (int32_t q = b; q < e; ++q) s += v[6 * (int64_t)gnodes[q] + k];
  s /= (double)(e - b);
  for (int32_t q = b; q < e; ++q) v[6 * (int64_t)gnodes[q] + k] = s;
}

}  // namespace pl
