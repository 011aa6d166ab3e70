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
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "pl_kernels.h"

namespace pl {

// ---- loopback transport -----------------------------------------------------------------------------------------
// R handles of ONE process on ONE device form a communicator without RCCL: every collective of pl_dist.h has a second
// implementation in which the ranks' contributions meet in device buffers.  The solver code above it is the same -
// pack kernels, weights, coarse band, single-reduction form, node elimination - so the whole multi-rank path runs (and is
// tested) with world = 2, 4, 8 on a one-GPU box, each rank driven by its own host thread exactly as each rank of an
// RCCL run is driven by its own process.  (It is also a way to run several sub-domains on one GPU.)
//
// One collective, seen from rank r (n = this rank's count of collectives so far, p = n & 1):
//   wait (stream) for done[p][q] of every rank q      - buffers of parity p were last read in collective n - 2
//   publish: write own contribution into own buffer of parity p, record ready[p][r]
//   HOST barrier of the R threads                     - every ready[p][q] of THIS collective is now recorded
//   wait (stream) for ready[p][q] of every q, combine (sum in rank order: the same bits on every rank), record done[p][r]
// Parity double-buffering makes one barrier per collective enough: a rank can only be one collective ahead of the
// slowest one, so an event or buffer of parity p is never re-used before everybody has queued its reads of it.
constexpr int kLoopMaxWorld = 16;
struct LoopPtrs {
  const double *p[kLoopMaxWorld];
};
__global__ void k_loop_sum(int world, LoopPtrs src, int64_t n, double *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = src.p[0][i];
  for (int q = 1; q < world; ++q) acc += src.p[q][i];
  out[i] = acc;
}

struct LoopGroup {
  int world = 0, device = -1, attached = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;
  hipEvent_t ready[2][kLoopMaxWorld] = {}, done[2][kLoopMaxWorld] = {};
  const double *stage[2][kLoopMaxWorld] = {};                  // published all-reduce contributions
  const double *send[2][kLoopMaxWorld][kLoopMaxWorld] = {};    // published neighbour messages [parity][from][to]
  // all ranks arrive or the group is declared broken (a rank that failed elsewhere never arrives: no hang, an error)
  bool barrier(double timeout_s = 120.0) {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; });
    if (!ok || broken) {
      broken = true;
      cv.notify_all();
      return false;
    }
    return true;
  }
};
inline std::mutex &loop_registry_mutex() {
  static std::mutex m;
  return m;
}
inline std::map<uint64_t, LoopGroup *> &loop_registry() {
  static std::map<uint64_t, LoopGroup *> r;
  return r;
}
constexpr char kLoopMagic[8] = {'P', 'L', 'L', 'O', 'O', 'P', 'v', '1'};

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
  DBuf<double> send2;      // loopback transport: the message buffer of odd collectives
};

struct Dist {
  bool active = false;
  int rank = 0, world = 1;
  int comm_count = 0, comm_user_rank = -1;   // as the communicator itself reports them (ncclCommCount / ncclCommUserRank)
  ncclComm_t comm = nullptr;
  // loopback transport (see above): non-null instead of comm
  LoopGroup *loop = nullptr;
  uint64_t loop_key = 0;
  uint64_t loop_ops = 0;                   // collectives this rank has queued (all ranks count alike)
  double *loop_stage[2] = {nullptr, nullptr};
  size_t loop_cap[2] = {0, 0};
  std::vector<double *> loop_garbage;      // outgrown staging buffers: other ranks may still read them, freed at the end
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

// A loopback "unique id": magic + serial number; pl_dist_init recognises it and attaches the handle to the in-process
// group of that number instead of an RCCL communicator.
inline void dist_loopback_id(void *out) {
  static uint64_t serial = 0;
  std::memset(out, 0, sizeof(ncclUniqueId));
  std::memcpy(out, kLoopMagic, 8);
  uint64_t key;
  {
    std::lock_guard<std::mutex> lk(loop_registry_mutex());
    key = ++serial;
  }
  std::memcpy(static_cast<char *>(out) + 8, &key, sizeof(key));
}
inline bool dist_is_loopback_id(const void *uid) { return std::memcmp(uid, kLoopMagic, 8) == 0; }

// ---- loopback collectives ------------------------------------------------------------------------------------------
// begin: the stream waits until every rank has finished reading the parity-p buffers of two collectives ago
inline int loop_begin(Dist &d, hipStream_t s, int &p) {
  LoopGroup &g = *d.loop;
  p = (int)(d.loop_ops & 1);
  if (d.loop_ops >= 2)
    for (int q = 0; q < g.world; ++q)
      if (hipStreamWaitEvent(s, g.done[p][q], 0) != hipSuccess) return 1;
  return 0;
}
// middle: own contribution is queued -> ready; meet the other ranks; wait for theirs
inline int loop_meet(Dist &d, hipStream_t s, int p) {
  LoopGroup &g = *d.loop;
  if (hipEventRecord(g.ready[p][d.rank], s) != hipSuccess) return 1;
  if (!g.barrier()) return 3;
  for (int q = 0; q < g.world; ++q)
    if (q != d.rank && hipStreamWaitEvent(s, g.ready[p][q], 0) != hipSuccess) return 1;
  return 0;
}
inline int loop_end(Dist &d, hipStream_t s, int p) {
  if (hipEventRecord(d.loop->done[p][d.rank], s) != hipSuccess) return 1;
  ++d.loop_ops;
  return 0;
}
inline int loop_allreduce(Dist &d, double *buf, size_t count, hipStream_t s) {
  LoopGroup &g = *d.loop;
  int p = 0;
  if (loop_begin(d, s, p)) return 1;
  if (d.loop_cap[p] < count) {   // grow-only; the old buffer may still be read by a slower rank: keep it until the end
    if (d.loop_stage[p]) d.loop_garbage.push_back(d.loop_stage[p]);
    d.loop_stage[p] = nullptr;
    const size_t cap = std::max<size_t>(count + count / 4, 1024);
    if (hipMalloc(reinterpret_cast<void **>(&d.loop_stage[p]), cap * sizeof(double)) != hipSuccess) return 2;
    d.loop_cap[p] = cap;
  }
  if (hipMemcpyAsync(d.loop_stage[p], buf, count * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
  g.stage[p][d.rank] = d.loop_stage[p];      // (published under the barrier's mutex ordering)
  if (int rc = loop_meet(d, s, p)) return rc;
  LoopPtrs src;
  for (int q = 0; q < g.world; ++q) src.p[q] = g.stage[p][q];
  hipLaunchKernelGGL(k_loop_sum, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, g.world, src, (int64_t)count, buf);
  return loop_end(d, s, p);
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
  if (d.loop) {   // loopback: pack into this collective's message buffers, meet, add straight from the neighbours' buffers
    LoopGroup &g = *d.loop;
    int p = 0;
    if (loop_begin(d, s, p)) return 2;
    for (Peer *q : d.peers) {
      double *msg = p ? q->send2.p : q->send.p;
      if (q->n > 0)
        hipLaunchKernelGGL(k_pack_rows<VT>, dim3((unsigned)((q->n * 6 + 255) / 256)), dim3(256), 0, s, q->n, q->loc.p, y, msg);
      g.send[p][d.rank][q->rank] = msg;
    }
    if (loop_meet(d, s, p)) return 2;
    for (Peer *q : d.peers)
      if (q->n > 0)
        hipLaunchKernelGGL(k_add_rows<VT>, dim3((unsigned)((q->n * 6 + 255) / 256)), dim3(256), 0, s, q->n, q->loc.p,
                           g.send[p][q->rank][d.rank], y);
    return loop_end(d, s, p) ? 2 : 0;
  }
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

// in-place sum over ranks of `count` device doubles: RCCL all-reduce, or the loopback transport
inline int dist_allreduce(Dist &d, double *buf, size_t count, hipStream_t s) {
  if (count == 0) return 0;
  if (d.loop) return loop_allreduce(d, buf, count, s);
  return ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, d.comm, s) == ncclSuccess ? 0 : 1;
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
    if (scal && nscal > 0 && dist_allreduce(d, scal, (size_t)nscal, s)) return 2;
    return 0;
  }
  const int64_t nrow = (int64_t)d.n_shared_global * 6;
  if (!scal) nscal = 0;
  const int64_t n = nrow + nscal;
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_pack_message<VT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, nrow, d.slot2loc.p, y, scal,
                     nscal, d.pack.p);
  if (dist_allreduce(d, d.pack.p, (size_t)n, s)) return 2;
  const int64_t m = (int64_t)d.n_shared * 6 + nscal;
  if (m > 0)
    hipLaunchKernelGGL(k_unpack_message<VT>, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d.n_shared,
                       d.local_idx.p, d.global_idx.p, d.pack.p, y, nrow, scal, nscal);
  return 0;
}

inline int dist_sum_scalars(Dist &d, double *dev, int count, hipStream_t s) {
  if (!d.active) return 0;
  return dist_allreduce(d, dev, (size_t)count, s);
}

inline int dist_init(Dist &d, int rank, int world, const void *uid, const int32_t *loc, const int32_t *glob,
                     int32_t n_shared, int32_t n_shared_global, int64_t N, hipStream_t s) {
  if (dist_is_loopback_id(uid)) {
    if (world > kLoopMaxWorld) return 6;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return 1;
    uint64_t key = 0;
    std::memcpy(&key, static_cast<const char *>(uid) + 8, sizeof(key));
    std::lock_guard<std::mutex> lk(loop_registry_mutex());
    LoopGroup *&g = loop_registry()[key];
    if (!g) {
      g = new LoopGroup();
      g->world = world;
      g->device = dev;
      bool ok = true;
      for (int p = 0; p < 2 && ok; ++p)
        for (int q = 0; q < world && ok; ++q)
          ok = hipEventCreateWithFlags(&g->ready[p][q], hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&g->done[p][q], hipEventDisableTiming) == hipSuccess;
      if (!ok) {   // no half-built group stays registered (ranks attaching later would wait on null events)
        for (int p = 0; p < 2; ++p)
          for (int q = 0; q < world; ++q) {
            if (g->ready[p][q]) (void)hipEventDestroy(g->ready[p][q]);
            if (g->done[p][q]) (void)hipEventDestroy(g->done[p][q]);
          }
        delete g;
        loop_registry().erase(key);
        return 1;
      }
    }
    if (g->world != world || g->device != dev || g->attached >= world) return 6;   // one device, world ranks, once each
    ++g->attached;
    d.loop = g;
    d.loop_key = key;
    d.loop_ops = 0;
  } else {
    ncclUniqueId id;
    std::memcpy(&id, uid, sizeof(id));
    if (ncclCommInitRank(&d.comm, world, id, rank) != ncclSuccess) return 1;
    // what the communicator says about itself - the evidence that RCCL saw `world` ranks (pl_stats_t.comm_world / comm_rank)
    if (ncclCommCount(d.comm, &d.comm_count) != ncclSuccess || ncclCommUserRank(d.comm, &d.comm_user_rank) != ncclSuccess)
      return 1;
    if (d.comm_count != world || d.comm_user_rank != rank) return 6;
  }
  if (d.loop) {
    d.comm_count = d.loop->world;
    d.comm_user_rank = rank;
  }
  d.rank = rank;
  d.world = world;
  d.n_shared = n_shared;
  d.n_shared_global = n_shared_global;
  // a rank that fails below never reaches the group's first collective: tell the peers now instead of after the timeout
  struct BreakOnError {
    Dist &d;
    bool armed = true;
    ~BreakOnError() {
      if (armed && d.loop) {
        std::lock_guard<std::mutex> lk(d.loop->m);
        d.loop->broken = true;
        d.loop->cv.notify_all();
      }
    }
  } guard{d};
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
  guard.armed = false;
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
    if (d.loop && q->send2.alloc(std::max<size_t>(1, loc.size() * 6)) != hipSuccess) return 3;
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
  if (d.loop) {
    // the caller has drained this rank's stream; other ranks may still be reading our buffers in kernels they have
    // queued: wait for the device before freeing anything they could touch
    (void)hipDeviceSynchronize();
    for (double *b : d.loop_garbage) (void)hipFree(b);
    d.loop_garbage.clear();
    for (int p = 0; p < 2; ++p) {
      if (d.loop_stage[p]) (void)hipFree(d.loop_stage[p]);
      d.loop_stage[p] = nullptr;
      d.loop_cap[p] = 0;
    }
    std::lock_guard<std::mutex> lk(loop_registry_mutex());
    LoopGroup *g = d.loop;
    {
      std::lock_guard<std::mutex> lg(g->m);
      g->broken = true;             // a group that lost a rank cannot run another collective
      g->cv.notify_all();
    }
    if (--g->attached == 0) {
      for (int p = 0; p < 2; ++p)
        for (int q = 0; q < g->world; ++q) {
          if (g->ready[p][q]) (void)hipEventDestroy(g->ready[p][q]);
          if (g->done[p][q]) (void)hipEventDestroy(g->done[p][q]);
        }
      loop_registry().erase(d.loop_key);
      delete g;
    }
    d.loop = nullptr;
  }
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
