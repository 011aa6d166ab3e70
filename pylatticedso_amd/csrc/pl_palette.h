// Record palette: periodic lattices repeat a handful of distinct strut records millions of times.  After the record
// build, struts are hashed (record quantised to 40 mantissa bits) into a 65 536-slot open-addressing table; if every
// strut finds a slot within a few probes and no two different records share one, K*p reads a 2-byte palette id per
// strut instead of its 64-byte record (the table itself, 4 MiB, stays in L2).  Graded / optimised lattices with more
// distinct records than the table holds simply keep the plain path (ok flag = 0).
#pragma once
#include <hip/hip_runtime.h>

#include "pl_kernels.h"

namespace pl {

constexpr int kPalBits = 16;
constexpr int kPalSize = 1 << kPalBits;
constexpr int kPalProbes = 32;
constexpr unsigned long long kPalEmpty = ~0ull;

__device__ __forceinline__ unsigned long long pal_quant(double v) {
  return (unsigned long long)__double_as_longlong(v + 0.0) & ~0xFFFull;       // drop 12 of 52 mantissa bits
}
__device__ __forceinline__ unsigned long long pal_hash(const Record &r) {
  const double f[8] = {r.a, r.c, r.e1, r.e2, r.e3, r.dx, r.dy, r.dz};
  unsigned long long h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    h ^= pal_quant(f[k]) + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
  }
  return h == kPalEmpty ? 0 : h;
}

// pass 1: claim a slot per distinct hash; owner[slot] = smallest strut id that uses it; flags[0] = 1 on overflow.
// Millions of struts share a few hundred slots.  One lane per wave and distinct hash (the lowest: strut ids ascend with
// the lane) probes the table and settles the owner, the others take its answer - otherwise the 64 lanes of every wave
// hammer the same few addresses of one L2 channel (and, the vector L1 being incoherent, keep seeing a slot as empty
// after another CU claimed it, so that nearly every strut issued the compare-and-swap: 0.27 ms at 3 M struts).
// A wave with many distinct hashes (graded lattices) leaves the grouping after kPalGroups rounds and lets the
// remaining lanes probe for themselves, in parallel.
constexpr int kPalGroups = 12;
__device__ __forceinline__ int pal_probe(unsigned long long h, unsigned long long *__restrict__ keys,
                                         int *__restrict__ owner, int b) {
  unsigned slot = (unsigned)(h >> (64 - kPalBits));
  for (int probe = 0; probe < kPalProbes; ++probe) {
    unsigned long long cur = __hip_atomic_load(keys + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kPalEmpty) cur = atomicCAS(keys + slot, kPalEmpty, h);
    if (cur == kPalEmpty || cur == h) {
      // the owner only decreases: a read that is already smaller spares the atomic
      if (b < __hip_atomic_load(owner + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(owner + slot, b);
      return (int)slot;
    }
    slot = (slot + 1) & (kPalSize - 1);
  }
  return -1;
}
__global__ __launch_bounds__(kBlock) void k_pal_insert(int64_t B, const Record *__restrict__ rec,
                                                       unsigned long long *__restrict__ keys,
                                                       int *__restrict__ owner, uint16_t *__restrict__ pal,
                                                       int *__restrict__ flags) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool valid = b < B;
  const unsigned long long h = valid ? pal_hash(load_record(rec, b)) : 0ull;
  const int lane = threadIdx.x & 63;
  int found = -2;                                   // -2: not settled yet, -1: table full
  unsigned long long todo = __ballot(valid);
  for (int round = 0; round < kPalGroups && todo; ++round) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned long long hl =
        ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(h >> 32), leader) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(h & 0xFFFFFFFFull), leader);
    const unsigned long long same = __ballot(valid && found == -2 && h == hl);
    int sl = 0;
    if (lane == leader) sl = pal_probe(h, keys, owner, (int)b);
    sl = __builtin_amdgcn_readlane(sl, leader);
    if ((same >> lane) & 1ull) found = sl;
    todo &= ~same;
  }
  if (valid && found == -2) found = pal_probe(h, keys, owner, (int)b);
  if (valid) {
    if (found < 0) flags[0] = 1;
    else pal[b] = (uint16_t)found;
  }
}
// pass 2: the owner publishes its record
__global__ __launch_bounds__(kBlock) void k_pal_publish(int64_t B, const Record *__restrict__ rec,
                                                        const int *__restrict__ owner,
                                                        const uint16_t *__restrict__ pal,
                                                        Record *__restrict__ palette) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const unsigned slot = pal[b];
  if (owner[slot] == (int)b) palette[slot] = rec[b];
}
// pass 3: every strut checks that its slot really holds its (quantised) record; the owners count the distinct entries and
// number them densely (dense_of_slot[slot] = 0, 1, ... in arrival order; the first kDense of them copied to `dense`: the
// table the LDS-resident K*p keeps in LDS, pl_tile.h)
__global__ __launch_bounds__(kBlock) void k_pal_verify(int64_t B, const Record *__restrict__ rec,
                                                       const uint16_t *__restrict__ pal,
                                                       const Record *__restrict__ palette,
                                                       const int *__restrict__ owner, int *__restrict__ flags,
                                                       int *__restrict__ dense_of_slot, Record *__restrict__ dense,
                                                       int n_dense_max) {
  const int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (b >= B) return;
  const Record r = load_record(rec, b), q = load_record(palette, pal[b]);
  const bool same = pal_quant(r.a) == pal_quant(q.a) && pal_quant(r.c) == pal_quant(q.c) &&
                    pal_quant(r.e1) == pal_quant(q.e1) && pal_quant(r.e2) == pal_quant(q.e2) &&
                    pal_quant(r.e3) == pal_quant(q.e3) && pal_quant(r.dx) == pal_quant(q.dx) &&
                    pal_quant(r.dy) == pal_quant(q.dy) && pal_quant(r.dz) == pal_quant(q.dz);
  if (!same) flags[0] = 1;
  if (owner[pal[b]] == (int)b) {
    const int d = atomicAdd(flags + 1, 1);
    dense_of_slot[pal[b]] = d;
    if (d < n_dense_max) dense[d] = q;
  }
}
}  // namespace pl
