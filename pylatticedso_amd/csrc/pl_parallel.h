// Host-side parallel loops of libpylattice_hip (plain std::thread; no OpenMP runtime to clash with the ones PyTorch and
// numpy bring along).
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

namespace pl {

// Threads per host loop: the CPUs this process may run on (affinity mask), shared between the ranks of a node when a
// launcher says how many there are (LOCAL_WORLD_SIZE, as torch.distributed.run sets it), at most 64; PL_HOST_THREADS
// overrides.
inline unsigned host_workers() {
  static const unsigned cached = []() -> unsigned {
    if (const char *e = std::getenv("PL_HOST_THREADS")) {
      const int v = std::atoi(e);
      if (v > 0) return (unsigned)std::min(v, 256);
    }
    unsigned n = std::thread::hardware_concurrency();
    if (!n) n = 4;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const int c = CPU_COUNT(&set);
      if (c > 0) n = std::min(n, (unsigned)c);
    }
    {   // cgroup CPU quota (a GPU box hands a one-GPU job 16 of its 256 hardware threads while the affinity mask still
        // shows them all: 64 threads on 16 CPUs' worth of time is slower than 16)
      unsigned quota = 0;
      if (FILE *fh = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long q = 0, per = 0;
        if (std::fscanf(fh, "%lld %lld", &q, &per) == 2 && q > 0 && per > 0) quota = (unsigned)((q + per - 1) / per);
        std::fclose(fh);
      } else if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long long q = 0, per = 0;
        if (std::fscanf(fq, "%lld", &q) == 1 && q > 0) {
          if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(fp, "%lld", &per) == 1 && per > 0) quota = (unsigned)((q + per - 1) / per);
            std::fclose(fp);
          }
        }
        std::fclose(fq);
      }
      if (quota > 0) n = std::min(n, std::max(1u, quota));
    }
    if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) {
      const int v = std::atoi(e);
      if (v > 1) n = std::max(1u, n / (unsigned)v);
    }
    return std::max(1u, std::min(n, 64u));
  }();
  return cached;
}

// body(begin, end, worker) on [0, n) cut into one contiguous piece per worker; sequential below `grain` items per worker
template <typename F>
inline void parallel_for(int64_t n, F &&body, int64_t grain = 1024) {
  const unsigned W = (unsigned)std::min<int64_t>(host_workers(), std::max<int64_t>(1, n / grain));
  if (W <= 1) {
    body((int64_t)0, n, 0u);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(W);
  for (unsigned w = 0; w < W; ++w) {
    const int64_t b = n * w / W, e = n * (w + 1) / W;
    th.emplace_back([&body, b, e, w]() { body(b, e, w); });
  }
  for (auto &t : th) t.join();
}

}  // namespace pl
