// Multi-GPU support of libpylattice_hip: one handle per rank/GPU, slab-partitioned lattice, RCCL over xGMI.
//
// Per PCG iteration the only exchanges are (i) the sum of the partial nodal forces on the interface nodes shared
// by neighbouring slabs and (ii) the scalar dot products.  Interface forces are packed into one dense vector
// indexed by a global interface id (zero where this rank does not touch the node), all-reduced with RCCL, and
// unpacked; dot products weight every dof by 1/multiplicity so shared dofs count once.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "pl_kernels.h"

namespace pl {

template <typename T>
struct DBuf {
  T *p = nullptr;
  ~DBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) {
    if (p) (void)hipFree(p);
    p = nullptr;
    return n ? hipMalloc(reinterpret_cast<void **>(&p), n * sizeof(T)) : hipSuccess;
  }
};

// One neighbouring rank of a slab partition: the nodes shared with it (both sides list them in the order of their
// global interface ids) and the staging buffers of the point-to-point exchange.
struct Peer {
  int rank = -1;
  int32_t n = 0;
  DBuf<int32_t> loc;
  DBuf<double> send, recv;
};

struct Dist {
  bool active = false;
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  // interface rows by grouped ncclSend / ncclRecv with the (at most two) neighbouring slabs instead of an all-reduce
  // over ALL interface planes (pl_dist_set_peers): at N ranks the all-reduce carries N - 1 planes to everyone, each
  // rank needs two
  bool p2p = false;
  std::vector<Peer *> peers;
  std::vector<int32_t> h_loc, h_glob;
  ~Dist() { for (Peer *q : peers) delete q; }
  int32_t n_shared = 0, n_shared_global = 0;
  DBuf<int32_t> local_idx, global_idx;   // [n_shared]
  DBuf<int32_t> slot2loc;                // [n_shared_global] local node of a global interface slot, -1 if not here
  DBuf<double> pack;                     // [6*n_shared_global]
  DBuf<double> weight;                   // [6N] 1/multiplicity
};

// One kernel fills the whole message: interface rows this rank holds (others zero: the all-reduce sums ranks), then
// the optional scalar tail.  slot2loc[g] = local node of global interface slot g, or -1.
template <typename VT>
__global__ void k_pack_message(int64_t nrow, const int32_t *__restrict__ slot2loc, const VT *__restrict__ y,
                               const double *__restrict__ scal, int nscal, double *__restrict__ pack) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrow) {
    const int64_t g = i / 6, k = i - 6 * g;
    const int32_t l = slot2loc[g];
    pack[i] = l >= 0 ? (double)y[6 * (int64_t)l + k] : 0.0;
  } else if (i < nrow + nscal) {
    pack[i] = scal[i - nrow];
  }
}
// ... and one kernel takes it apart again.
template <typename VT>
__global__ void k_unpack_message(int32_t n, const int32_t *__restrict__ loc, const int32_t *__restrict__ glob,
                                 const double *__restrict__ pack, VT *__restrict__ y, int64_t nrow,
                                 double *__restrict__ scal, int nscal) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n6 = (int64_t)n * 6;
  if (i < n6) {
    const int64_t s = i / 6, k = i - 6 * s;
    y[6 * (int64_t)loc[s] + k] = (VT)pack[6 * (int64_t)glob[s] + k];
  } else if (i < n6 + nscal) {
    scal[i - n6] = pack[nrow + (i - n6)];
  }
}
__global__ void k_fill(int64_t n, double v, double *__restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}
__global__ void k_recip(int64_t n, double *__restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 1.0 / x[i];
}

inline int dist_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }
inline int dist_unique_id(void *out) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return 1;
  std::memcpy(out, &id, sizeof(id));
  return 0;
}

template <typename VT>
__global__ void k_pack_rows(int32_t n, const int32_t *__restrict__ loc, const VT *__restrict__ y,
                            double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)n * 6) out[i] = (double)y[6 * (int64_t)loc[i / 6] + i % 6];
}
template <typename VT>
__global__ void k_add_rows(int32_t n, const int32_t *__restrict__ loc, const double *__restrict__ in,
                           VT *__restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)n * 6) {
    VT *q = y + 6 * (int64_t)loc[i / 6] + i % 6;
    *q = (VT)((double)*q + in[i]);       // mine + theirs on both sides: the same bits on both ranks
  }
}

// Neighbour exchange: every rank sends its partial rows of each shared plane to the rank on the other side and adds
// what it receives.  All sends and receives of one call form ONE RCCL group (they progress concurrently).
template <typename VT>
inline int dist_exchange_p2p(Dist &d, VT *y, hipStream_t s) {
  for (Peer *q : d.peers)
    if (q->n > 0)
      hipLaunchKernelGGL(k_pack_rows<VT>, dim3((unsigned)((q->n * 6 + 255) / 256)), dim3(256), 0, s, q->n, q->loc.p, y,
                         q->send.p);
  if (ncclGroupStart() != ncclSuccess) return 2;
  for (Peer *q : d.peers)
    if (q->n > 0) {
      if (ncclSend(q->send.p, (size_t)q->n * 6, ncclDouble, q->rank, d.comm, s) != ncclSuccess) return 2;
      if (ncclRecv(q->recv.p, (size_t)q->n * 6, ncclDouble, q->rank, d.comm, s) != ncclSuccess) return 2;
    }
  if (ncclGroupEnd() != ncclSuccess) return 2;
  for (Peer *q : d.peers)
    if (q->n > 0)
      hipLaunchKernelGGL(k_add_rows<VT>, dim3((unsigned)((q->n * 6 + 255) / 256)), dim3(256), 0, s, q->n, q->loc.p,
                         (const double *)q->recv.p, y);
  return 0;
}

// y[shared] <- sum over ranks of y[shared]; optionally `nscal` device scalars (e.g. the 32 slots of a LOCAL partial
// dot product) ride in the tail of the same message and are summed over ranks too: ONE collective, one kernel
// before it and one after.
// (VT = float: the fp32 solver modes; the message itself stays fp64 - it is small and latency-bound)
template <typename VT>
inline int dist_sum_shared(Dist &d, VT *y, hipStream_t s, double *scal = nullptr, int nscal = 0) {
  if (!d.active) return 0;
  if (d.p2p) {   // rows with the neighbours; the scalar tail needs a (tiny) all-reduce of its own
    if (dist_exchange_p2p<VT>(d, y, s)) return 2;
    if (scal && nscal > 0 && ncclAllReduce(scal, scal, (size_t)nscal, ncclDouble, ncclSum, d.comm, s) != ncclSuccess)
      return 2;
    return 0;
  }
  const int64_t nrow = (int64_t)d.n_shared_global * 6;
  if (!scal) nscal = 0;
  const int64_t n = nrow + nscal;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pack_message<VT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nrow, d.slot2loc.p, y, scal,
                     nscal, d.pack.p);
  if (ncclAllReduce(d.pack.p, d.pack.p, (size_t)n, ncclDouble, ncclSum, d.comm, s) != ncclSuccess) return 2;
  const int64_t m = (int64_t)d.n_shared * 6 + nscal;
  if (m > 0)
    hipLaunchKernelGGL(k_unpack_message<VT>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d.n_shared,
                       d.local_idx.p, d.global_idx.p, d.pack.p, y, nrow, scal, nscal);
  return 0;
}

inline int dist_sum_scalars(Dist &d, double *dev, int count, hipStream_t s) {
  if (!d.active) return 0;
  return ncclAllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, d.comm, s) == ncclSuccess ? 0 : 1;
}

inline int dist_init(Dist &d, int rank, int world, const void *uid, const int32_t *loc, const int32_t *glob,
                     int32_t n_shared, int32_t n_shared_global, int64_t N, hipStream_t s) {
  ncclUniqueId id;
  std::memcpy(&id, uid, sizeof(id));
  if (ncclCommInitRank(&d.comm, world, id, rank) != ncclSuccess) return 1;
  d.rank = rank;
  d.world = world;
  d.n_shared = n_shared;
  d.n_shared_global = n_shared_global;
  if (d.local_idx.alloc(std::max(1, n_shared)) != hipSuccess) return 2;
  if (d.global_idx.alloc(std::max(1, n_shared)) != hipSuccess) return 2;
  if (d.pack.alloc((size_t)n_shared_global * 6 + 4 * kSlots) != hipSuccess) return 2;   // rows + scalar tail
  if (d.weight.alloc((size_t)N * 6) != hipSuccess) return 2;
  if (n_shared > 0) {
    if (hipMemcpy(d.local_idx.p, loc, n_shared * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return 3;
    if (hipMemcpy(d.global_idx.p, glob, n_shared * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return 3;
  }
  {
    std::vector<int32_t> s2l((size_t)std::max(1, n_shared_global), -1);
    for (int32_t i = 0; i < n_shared; ++i) s2l[glob[i]] = loc[i];
    if (d.slot2loc.alloc(s2l.size()) != hipSuccess) return 2;
    if (hipMemcpy(d.slot2loc.p, s2l.data(), s2l.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return 3;
  }
  d.h_loc.assign(loc, loc + n_shared);
  d.h_glob.assign(glob, glob + n_shared);
  d.active = true;
  // multiplicity = all-reduce of ones on the shared nodes; weight = 1/multiplicity
  const int64_t n6 = N * 6;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n6 + 255) / 256)), dim3(256), 0, s, n6, 1.0, d.weight.p);
  if (dist_sum_shared(d, d.weight.p, s)) return 4;
  hipLaunchKernelGGL(k_recip, dim3((unsigned)((n6 + 255) / 256)), dim3(256), 0, s, n6, d.weight.p);
  if (hipStreamSynchronize(s) != hipSuccess) return 5;
  return 0;
}

// peer[i] = rank on the other side of shared entry i (as passed to dist_init).  Both ranks of a pair order the common
// nodes by global interface id, so send and receive buffers line up without any further exchange.
inline int dist_set_peers(Dist &d, const int32_t *peer) {
  if (!d.active) return 1;
  for (Peer *q : d.peers) delete q;
  d.peers.clear();
  std::vector<int32_t> ranks(peer, peer + d.n_shared);
  std::sort(ranks.begin(), ranks.end());
  ranks.erase(std::unique(ranks.begin(), ranks.end()), ranks.end());
  for (int32_t pr : ranks) {
    if (pr < 0 || pr >= d.world) return 2;
    std::vector<std::pair<int32_t, int32_t>> rows;   // (global id, local node)
    for (int32_t i = 0; i < d.n_shared; ++i)
      if (peer[i] == pr) rows.push_back({d.h_glob[i], d.h_loc[i]});
    std::sort(rows.begin(), rows.end());
    std::vector<int32_t> loc(rows.size());
    for (size_t k = 0; k < rows.size(); ++k) loc[k] = rows[k].second;
    Peer *q = new Peer();
    q->rank = pr;
    q->n = (int32_t)rows.size();
    d.peers.push_back(q);
    if (q->loc.alloc(std::max<size_t>(1, loc.size())) != hipSuccess) return 3;
    if (q->send.alloc(std::max<size_t>(1, loc.size() * 6)) != hipSuccess) return 3;
    if (q->recv.alloc(std::max<size_t>(1, loc.size() * 6)) != hipSuccess) return 3;
    if (!loc.empty() &&
        hipMemcpy(q->loc.p, loc.data(), loc.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
      return 3;
  }
  d.p2p = true;
  return 0;
}

inline void dist_destroy(Dist &d) {
  if (d.comm) (void)ncclCommDestroy(d.comm);
  d.comm = nullptr;
  d.active = false;
}

// ---- weighted variants of the reduction kernels (shared dofs count once) ---------------------------------------
__global__ __launch_bounds__(kBlock) void k_mask_dot_w(int64_t n6, const uint8_t *__restrict__ fixed,
                                                       const double *__restrict__ w, const double *__restrict__ x,
                                                       double *__restrict__ y, double *__restrict__ dot_out) {
  __shared__ double red[kBlock / kWave];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * kBlock) {
    double v = y[i];
    if (fixed && fixed[i]) { v = 0.0; y[i] = 0.0; }
    acc += w[i] * x[i] * v;
  }
  if (dot_out) {
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) unsafeAtomicAdd(dot_out + (blockIdx.x & (kSlots - 1)), t);
  }
}

__global__ __launch_bounds__(kBlock) void k_pcg_update_w(int64_t n6, const double *__restrict__ p,
                                                         const double *__restrict__ Ap,
                                                         const double *__restrict__ dinv,
                                                         const double *__restrict__ w, double *__restrict__ x,
                                                         double *__restrict__ r, double *__restrict__ z,
                                                         double *__restrict__ scal) {
  __shared__ double red[2][kBlock / kWave];
  const double pap = scalar_read(scal, S_PAP);
  const double alpha = (pap != 0.0) ? scalar_read(scal, S_RZ_OLD) / pap : 0.0;
  double rz = 0.0, rr = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n6; i += (int64_t)gridDim.x * kBlock) {
    const double xv = x[i] + alpha * p[i];
    const double rv = r[i] - alpha * Ap[i];
    const double zv = dinv[i] * rv;
    x[i] = xv;
    r[i] = rv;
    z[i] = zv;
    rz += w[i] * rv * zv;
    rr += w[i] * rv * rv;
  }
  rz = wave_sum(rz);
  rr = wave_sum(rr);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = rz; red[1][wv] = rr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int k = 0; k < kBlock / kWave; ++k) { a += red[0][k]; b += red[1][k]; }
    scalar_add(scal, S_RZ_NEW, a);
    scalar_add(scal, S_RR, b);
  }
}

__global__ __launch_bounds__(kBlock) void k_pcg_init_w(int64_t n6, const double *__restrict__ f,
                                                       const double *__restrict__ Kubar,
                                                       const uint8_t *__restrict__ fixed,
                                                       const double *__restrict__ dinv,
                                                       const double *__restrict__ w, double *__restrict__ x,
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
    rz += w[i] * rv * zv;
    rr += w[i] * rv * rv;
  }
  rz = wave_sum(rz);
  rr = wave_sum(rr);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = rz; red[1][wv] = rr; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, b = 0;
    for (int k = 0; k < kBlock / kWave; ++k) { a += red[0][k]; b += red[1][k]; }
    scalar_add(scal, S_RZ_OLD, a);
    scalar_add(scal, S_BB, b);
  }
}

__global__ void k_invert_diag(int64_t n6, const double *__restrict__ diag, const uint8_t *__restrict__ fixed,
                              double *__restrict__ dinv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n6) return;
  const double d = diag[i];
  dinv[i] = ((fixed && fixed[i]) || d == 0.0) ? 0.0 : 1.0 / d;
}

inline unsigned stream_grid(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
inline void launch_mask_dot_weighted(int64_t n6, const uint8_t *fixed, const double *w, const double *x, double *y,
                                     double *dot_dev, hipStream_t s) {
  hipLaunchKernelGGL(k_mask_dot_w, dim3(stream_grid(n6)), dim3(kBlock), 0, s, n6, fixed, w, x, y, dot_dev);
}
inline void launch_pcg_update_weighted(int64_t n6, const double *p, const double *Ap, const double *dinv,
                                       const double *w, double *x, double *r, double *z, double *scal,
                                       hipStream_t s) {
  hipLaunchKernelGGL(k_pcg_update_w, dim3(stream_grid(n6)), dim3(kBlock), 0, s, n6, p, Ap, dinv, w, x, r, z, scal);
}
inline void launch_pcg_init_weighted(int64_t n6, const double *f, const double *Kubar, const uint8_t *fixed,
                                     const double *dinv, const double *w, double *x, double *r, double *z, double *p,
                                     double *scal, hipStream_t s) {
  hipLaunchKernelGGL(k_pcg_init_w, dim3(stream_grid(n6)), dim3(kBlock), 0, s, n6, f, Kubar, fixed, dinv, w, x, r, z,
                     p, scal);
}
inline void launch_fill(int64_t n, double v, double *x, hipStream_t s) {
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, v, x);
}
inline void launch_invert_diag(int64_t n6, const double *diag, const uint8_t *fixed, double *dinv, hipStream_t s) {
  hipLaunchKernelGGL(k_invert_diag, dim3((unsigned)((n6 + 255) / 256)), dim3(256), 0, s, n6, diag, fixed, dinv);
}

}  // namespace pl
