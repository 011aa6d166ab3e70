// Host-side parallel loops of libpylattice_hip (plain std::thread; no OpenMP runtime to clash with the ones PyTorch and
// numpy bring along).
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

namespace pl {

inline unsigned host_workers() {
  const unsigned n = std::thread::hardware_concurrency();
  return std::max(1u, std::min(n ? n : 4u, 64u));
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
